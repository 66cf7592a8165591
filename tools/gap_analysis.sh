R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gap_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_out -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/gap.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("/tmp/gap_out/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
# last 2 steps: find adam kernels as step boundaries
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
print("adam launches", len(adam), "total kernels", len(rows))
a, b = adam[-2], adam[-1]
seg = rows[a + 1:b + 1]
span = seg[-1][1] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
print(f"last step: span {span/1e6:.2f} ms, kernel busy {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms, kernels {len(seg)}")
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(seg[:-1], seg[1:]):
    g = s1 - e0
    if g > 0: gaps.append((g, n0, n1))
gaps.sort(reverse=True)
print("largest gaps (us): before <- after")
for g, n0, n1 in gaps[:25]:
    print(f"{g/1e3:8.1f}  {n0[:50]:50s} -> {n1[:50]}")
import collections
hist = collections.Counter()
for g, _, _ in gaps:
    hist[min(int(g / 1e3) // 5 * 5, 100)] += 1
tot = collections.Counter()
for g, _, _ in gaps:
    tot[min(int(g / 1e3) // 5 * 5, 100)] += g
print("gap histogram (us bucket: count, total ms):", {k: (hist[k], round(tot[k] / 1e6, 2)) for k in sorted(hist)})
PY
