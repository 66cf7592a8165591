#!/bin/bash
# GPU call 7 of round 5: conv layer 0's forward on the matrix cores + the transcendental-free GELU: parity tests, conv0 / epilogue timings.
O=gpurun_out/r5g; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_bf16_path_gpu.py tests/test_fullsize_gpu.py tests/test_cfg2_shape_gpu.py tests/test_extractor_ln.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
run() { echo -n "$1: "; env $2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $3 2>$O/err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['roofline']['hbm_bound_kernels']; print(d['ms_per_step'], 'ms | NT frac', d['roofline']['frac'], '| conv0 fwd', h['conv0_gn_gelu_fwd']['ms'], 'ms', h['conv0_gn_gelu_fwd']['frac_of_8TBps'], '| bwd', h['conv0_gn_gelu_bwd']['ms'], 'ms', h['conv0_gn_gelu_bwd']['frac_of_8TBps'], '| frontend', d['roofline']['frontend']['ms'])"; }
{
run "matrix-core conv0 apply" "A=1" ""
run "VALU conv0 apply" "ST5_CONV0_MFMA=0" ""
run "matrix-core conv0 apply" "A=1" ""
run "VALU conv0 apply" "ST5_CONV0_MFMA=0" ""
} > $O/conv0.log 2>&1
cat $O/conv0.log
timeout 300 python tools/gemm_cases.py nt > $O/gemm_nt.log 2>&1; grep "^NT" $O/gemm_nt.log | cut -c1-230
