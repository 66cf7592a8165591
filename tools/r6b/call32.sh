#!/bin/bash
# full GPU suite + smoke at the tree with the conv layer 0 changes, then a bench pair (GELU tables off / on) on the same box
mkdir -p gpurun_out/r6b
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6b/gpu_suite_conv0.log 2>&1; grep -E "passed|failed" gpurun_out/r6b/gpu_suite_conv0.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for t in 0 1 0 1; do
  ST5_CONV0_GELU_TABLE=$t ST5_CONV0_FOLD=$t timeout 400 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > gpurun_out/r6b/bench_conv0_$t.json 2>/dev/null < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/bench_conv0_$t.json'));print('conv0 new paths $t:', d['ms_per_step'], 'ms', {k:v.get('device_ms') for k,v in d.get('hbm_bound_kernels',{}).items() if isinstance(v,dict)})"
done | tee gpurun_out/r6b/bench_conv0_ab.txt
