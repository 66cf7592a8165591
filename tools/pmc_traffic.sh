# HBM traffic of the bench step per kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
# `python bench.py --steps 3 --warmup 2 --no-cpu-baseline`, summarised into profiles/r1_pmc_traffic.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/pmc_w.log 2>&1
mkdir -p $R/gpurun_out/pmc
python - <<PY
import csv, glob, json, collections
out = collections.defaultdict(dict)
for d, name in (("/tmp/pmc_f", "FETCH_SIZE"), ("/tmp/pmc_w", "WRITE_SIZE")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counter file in", d); continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(fs[0])):
        if row["Counter_Name"] != name: continue
        k = row["Kernel_Name"]
        acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
    for k, (n, v) in acc.items():
        out[k][name + "_launches"] = n
        out[k][name + "_sum"] = v
res = {}
for k, v in out.items():
    if "gemm" in k or "flash" in k or "conv0" in k or "ln_" in k:
        res[k[:120]] = v
json.dump(res, open("$R/gpurun_out/pmc/pmc_traffic_raw.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_sum", 0))[:12]:
    print(k[:80], v)
PY
