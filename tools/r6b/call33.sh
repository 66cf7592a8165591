#!/bin/bash
# moments carried from forward to backward (st5_conv0_gn_gelu_{fwd,bwd}_m): conv0 tests, device times, then the full suite
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k conv0 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
timeout 300 python - <<PY 2>&1 | grep "conv0 device" | tee gpurun_out/r6b/conv0_mom_device_time.txt
import torch, bench
for _ in range(2):
    print("conv0 device time (B=8, back to back, ms):", bench.conv0_device_time(torch.device("cuda:0"), 8, reps=200))
PY
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6b/gpu_suite_conv0.log 2>&1; grep -E "passed|failed" gpurun_out/r6b/gpu_suite_conv0.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
