# Kernel-trace timeline of the LAST step of bench.py (graph replay by default): span, union-busy, idle, gap histogram, and the
# full kernel sequence (start offset, duration, gap before, queue, name) -> gpurun_out/timeline_$1.txt
TAG=${1:-r2}
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_out -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline ${2:-} > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log | cut -c1-160
python - <<PY > $R/gpurun_out/timeline_$TAG.txt
import csv, glob, collections
f = glob.glob("/tmp/tl_out/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
# the bench profiles one extra eager step after the timed region: take the step before it (a timed one)
a, b = adam[-3], adam[-2]
seg = rows[a + 1:b + 1]
t0 = seg[0][0]
span = seg[-1][1] - t0
ev = []
for s, e, _, _ in seg:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = over = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
summed = sum(e - s for s, e, _, _ in seg)
print(f"step: span {span/1e6:.2f} ms  union-busy {busy/1e6:.2f}  summed {summed/1e6:.2f}  >=2 resident {over/1e6:.2f}  idle {(span-busy)/1e6:.2f}  kernels {len(seg)}")
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    for k in ("at::native::", "_ZN12_GLOBAL__N_1"):
        n = n.replace(k, "")
    return n[:70]
end = t0
hist = collections.Counter(); tot = collections.Counter()
for s, e, n, q in seg:
    gap = s - end
    if gap > 0:
        k = min(int(gap / 1e3) // 2 * 2, 40); hist[k] += 1; tot[k] += gap
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  gap {gap/1e3:7.1f}  q{q}  {short(n)}")
    end = max(end, e)
print("gap histogram (us bucket: count, total ms):", {k: (hist[k], round(tot[k] / 1e6, 2)) for k in sorted(hist)})
PY
head -1 $R/gpurun_out/timeline_$TAG.txt; tail -1 $R/gpurun_out/timeline_$TAG.txt
