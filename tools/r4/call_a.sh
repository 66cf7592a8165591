#!/bin/bash
# round-4 GPU call A: whole -m gpu suite without -x at HEAD, the 2-rank shared-GPU bench lines, the default bench line
set -u
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider ) > $O/all_tests.log 2>&1
tail -60 $O/all_tests.log
for ex in phased one_message; do
  ( time timeout 600 python bench.py --gpus 2 --steps 10 --warmup 5 --no-cpu-baseline --exchange $ex ) > $O/bench_2rank_$ex.log 2>&1
  tail -3 $O/bench_2rank_$ex.log | cut -c1-1500
done
( time timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline ) > $O/bench_1.log 2>&1
tail -2 $O/bench_1.log | cut -c1-3000
