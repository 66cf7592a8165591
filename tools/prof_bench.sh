# rocprofv3 kernel trace + stats of bench.py -> gpurun_out/prof_$1/bench_kernel_stats.csv (copy the summary to profiles/)
TAG=${1:-r2}
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -- python $R/bench.py --steps ${2:-8} --warmup 3 --no-cpu-baseline > /tmp/bench_prof.log 2>&1
mkdir -p $R/gpurun_out/prof_$TAG
cp $(find /tmp/prof_out -name "*kernel_stats.csv") $R/gpurun_out/prof_$TAG/bench_kernel_stats.csv
cp $(find /tmp/prof_out -name "*kernel_trace.csv") /tmp/kt.csv 2>/dev/null
tail -1 /tmp/bench_prof.log | cut -c1-200
