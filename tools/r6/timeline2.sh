# Per-queue view of one replayed update (kernel trace): for every hardware queue its first start / last end / busy time / gaps,
# and the kernel sequence with queue ids -> gpurun_out/r6/timeline_$1.txt
TAG=${1:-a}
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_out -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline ${2:-} > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log | cut -c1-160
mkdir -p $R/gpurun_out/r6
python - <<PY > $R/gpurun_out/r6/timeline_$TAG.txt
import csv, glob, collections
f = glob.glob("/tmp/tl_out/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2] or "adam_step" in r[2]]
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
byq = collections.defaultdict(list)
for s, e, n, q in seg:
    byq[q].append((s, e, n))
for q, lst in sorted(byq.items()):
    bs = sum(e - s for s, e, _ in lst)
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    small = [g for g in gaps if 0 <= g < 20000]
    big = [g for g in gaps if g >= 20000]
    print(f"queue {q}: {len(lst)} kernels, first start {(lst[0][0]-t0)/1e6:.2f} ms, last end {(lst[-1][1]-t0)/1e6:.2f} ms, busy {bs/1e6:.2f} ms, "
          f"gaps < 20 us: {len(small)} totalling {sum(small)/1e6:.2f} ms (mean {sum(small)/max(len(small),1)/1e3:.1f} us), gaps >= 20 us: {len(big)} totalling {sum(big)/1e6:.2f} ms")
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    for k in ("at::native::", "_ZN12_GLOBAL__N_1"):
        n = n.replace(k, "")
    return n[:70]
for s, e, n, q in seg:
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  q{q}  {short(n)}")
PY
head -8 $R/gpurun_out/r6/timeline_$TAG.txt
