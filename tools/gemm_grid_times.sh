# Stand-alone durations of the step's GEMM / attention kernels by launch grid, from a kernel trace of the replayed step with the
# micro-batches IN TURN (ST5_OVERLAP_FWD=0, no weight-gradient / attention side streams: nothing co-resident) -> gpurun_out/gemm_grid_$1.txt.  grid.x = tiles, grid.y = k-splits.
TAG=${1:-r4}
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gg_out
ST5_OVERLAP_FWD=0 ST5_WGRAD_STREAM=0 ST5_ATTN_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/gg_out -- python $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline > /tmp/gg.log 2>&1
tail -1 /tmp/gg.log | cut -c1-120
python - <<PY > $R/gpurun_out/gemm_grid_$TAG.txt
import csv, glob, collections
f = glob.glob("/tmp/gg_out/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
seg = rows[adam[-3] + 1:adam[-2] + 1]   # one timed (replayed) step
agg = collections.OrderedDict()
def short(n):
    for k in ("(anonymous namespace)::", "void ", "_ZN12_GLOBAL__N_1"):
        n = n.replace(k, "")
    return n.split("(")[0][:44]
for r in seg:
    n = r["Kernel_Name"]
    if not any(k in n for k in ("gemm", "fa2::", "splitk", "ln_", "transpose")):
        continue
    wg = int(r["Workgroup_Size_X"])
    key = (short(n), int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    c, t, mn, mx = agg.get(key, (0, 0, 1 << 60, 0))
    agg[key] = (c + 1, t + d, min(mn, d), max(mx, d))
tot = sum(v[1] for v in agg.values())
print(f"step kernels {len(seg)}, listed kernels total {tot/1e6:.2f} ms")
for k, (c, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:44s} blocks {k[1]:5d} x{k[2]:3d} x{k[3]:3d}  n {c:3d}  total {t/1e6:7.3f} ms  avg {t/c/1e3:7.1f} us  (min {mn/1e3:.1f} max {mx/1e3:.1f})")
PY
head -64 $R/gpurun_out/gemm_grid_$TAG.txt
