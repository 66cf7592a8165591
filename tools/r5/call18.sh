#!/bin/bash
# F_RES specialised NT epilogue: tests + same-box A/B against the previous library (ST5_HIP_LIB)
mkdir -p gpurun_out/r5r
timeout 600 python -m pytest tests/test_bf16_path_gpu.py -x -q -m gpu -k "relay" > gpurun_out/r5r/tests.log 2>&1; tail -2 gpurun_out/r5r/tests.log
for i in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_prev.so; else unset ST5_HIP_LIB; fi
    timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r5r/${lib}_$i.json 2> gpurun_out/r5r/${lib}_$i.err
    python -c "import json;d=json.load(open('gpurun_out/r5r/${lib}_$i.json'));print('$lib $i', d['ms_per_step'], d['roofline']['frac'])"
  done
done
