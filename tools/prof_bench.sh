R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -- python $R/bench.py --steps 5 --warmup 2 > /tmp/bench_prof.log 2>&1
mkdir -p $R/gpurun_out/prof_r1g
cp $(find /tmp/prof_out -name "*kernel_stats.csv") $R/gpurun_out/prof_r1g/bench_kernel_stats.csv
tail -1 /tmp/bench_prof.log | cut -c1-200
