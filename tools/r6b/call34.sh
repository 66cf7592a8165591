#!/bin/bash
# closing validation at the tree with the conv layer 0 experiments backed out: the side-by-side repro loop, the full suite, smoke, finals
mkdir -p gpurun_out/r6b
for i in $(seq 1 12); do timeout 400 python -m pytest tests/test_fullsize_gpu.py tests/test_graph_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tail -1; done | tee gpurun_out/r6b/repro_reverted.txt
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6_gpu_suite_full.log 2>&1; tail -40 gpurun_out/r6_gpu_suite_full.log > gpurun_out/r6_gpu_suite.log; grep -E "passed|failed" gpurun_out/r6_gpu_suite.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/finals.sh r6 2>&1 | grep -E "rc=|ms_per_step" | cut -c1-300
