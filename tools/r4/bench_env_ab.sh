#!/bin/bash
# same-box A/B of one environment knob on the default bench: bench_env_ab.sh <outdir> <VAR> <rounds> <value>...   ("-" = unset)
O=gpurun_out/${1:-abe}; VAR=$2; N=$3; shift 3; mkdir -p $O
for i in $(seq 1 $N); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    ( timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline ) > $O/bench_${v}_$i.log 2>&1
    grep '^{' $O/bench_${v}_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', $i, d['ms_per_step'])" | tee -a $O/ab.txt
  done
done
