#!/bin/bash
# last GPU call of the round: full -m gpu suite (no -x) + smoke + the default bench with the CPU-baseline leg (reads profiles/r4_* for traffic / trace legs)
set -u
O=gpurun_out/${1:-r4final}; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider ) > $O/all_tests.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $O/all_tests.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(d["ms_per_step"], d["value"], r["achieved"], r["frac"], "traffic", r["traffic"], "alg", r["algorithmic_bytes_per_launch"], "trace", r["from_kernel_trace"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
