#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r6_gpu_suite.log; tail -3 gpurun_out/r6_gpu_suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
