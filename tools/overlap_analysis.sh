# Kernel-trace of bench.py: per-step span, union-of-kernels busy time, summed kernel time, time with >=2 kernels
# resident (weight-gradient stream beside the main stream), idle time.
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ov_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/ov_out -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/ov.log 2>&1
tail -1 /tmp/ov.log | cut -c1-200
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/ov_out/**/*kernel_trace.csv", recursive=True)[0]
rows = []
hdr = None
for r in csv.DictReader(open(f)):
    hdr = hdr or list(r.keys())
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Queue_Id", "?")))
print(hdr)
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
a, b = adam[-2], adam[-1]
seg = rows[a + 1:b + 1]
span = seg[-1][1] - seg[0][0]
summed = sum(e - s for s, e, _, _ in seg)
ev = []
for s, e, _, _ in seg:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = over = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
print(f"last step: span {span/1e6:.2f} ms  union-busy {busy/1e6:.2f}  summed kernel time {summed/1e6:.2f}  >=2 resident {over/1e6:.2f}  idle {(span-busy)/1e6:.2f}  kernels {len(seg)}")
byq = collections.Counter()
for s, e, n, q in seg: byq[q] += e - s
print("kernel time by queue (ms):", {k: round(v / 1e6, 2) for k, v in byq.items()})
PY
