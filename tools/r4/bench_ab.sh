#!/bin/bash
# same-box A/B of two builds of the library on the default bench: bench_ab.sh <outdir> <variant-tag> [rounds]   (variant = speecht5_amd/libspeecht5_hip_<tag>.so)
O=gpurun_out/${1:-ab}; V=$2; N=${3:-2}; mkdir -p $O
for i in $(seq 1 $N); do
  for v in base $V; do
    if [ $v = base ]; then unset ST5_HIP_LIB; else export ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_$v.so; fi
    ( timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline ) > $O/bench_${v}_$i.log 2>&1
    grep '^{' $O/bench_${v}_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', $i, d['ms_per_step'])" | tee -a $O/ab.txt
  done
done
