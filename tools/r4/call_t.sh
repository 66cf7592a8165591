#!/bin/bash
# full -m gpu suite (no -x) + default bench
set -u
O=gpurun_out/${1:-r4t}; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider ) > $O/all_tests.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $O/all_tests.log | tail -15
( timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline ) > $O/bench_1.log 2>&1
grep '^{' $O/bench_1.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], r['achieved'], r['frac'], r['step'])"
