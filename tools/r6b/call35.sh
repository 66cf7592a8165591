#!/bin/bash
# closing validation at the final tree: full GPU suite, smoke, then the round's finals (bench line last, with this run's summaries in place)
mkdir -p gpurun_out/r6b
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6_gpu_suite_full.log 2>&1; tail -40 gpurun_out/r6_gpu_suite_full.log > gpurun_out/r6_gpu_suite.log; grep -E "passed|failed" gpurun_out/r6_gpu_suite.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/finals.sh r6 2>&1 | grep -E "rc=|ms_per_step" | cut -c1-300
