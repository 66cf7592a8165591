# HBM traffic of the bench step per kernel: two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; no trace domains beside
# them) over `python bench.py --steps 3 --warmup 3 --no-cpu-baseline`, summarised into gpurun_out/pmc/$1_pmc_traffic.json
# (copy it to profiles/: bench.py reads profiles/r3_pmc_traffic.json and only trusts it when its gemm.hip hash matches).
TAG=${1:-r2}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline > /tmp/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline > /tmp/pmc_w.log 2>&1
mkdir -p $R/gpurun_out/pmc
python - <<PY
import csv, glob, json, collections, hashlib, sys; sys.path.insert(0, "$R")
acc = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for d, name in (("/tmp/pmc_f", "FETCH_SIZE"), ("/tmp/pmc_w", "WRITE_SIZE")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counter file in", d); continue
    for row in csv.DictReader(open(fs[0])):
        if row["Counter_Name"] != name: continue
        k = row["Kernel_Name"]
        if name == "FETCH_SIZE": acc[k]["launches"] += 1
        acc[k][name] += float(row["Counter_Value"])
kern = {}
for k, v in acc.items():
    if not any(t in k for t in ("gemm", "fa2::", "conv0", "ln_", "adam", "conv1d_narrow")) or not v["launches"]:
        continue
    n = v["launches"]
    f_kb, w_kb = v["FETCH_SIZE"] / n, v["WRITE_SIZE"] / n
    kern[k[:140]] = {"launches": n, "fetch_kb_per_launch": round(f_kb, 1), "write_kb_per_launch": round(w_kb, 1),
                     "hbm_corrected_bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024)}
out = {"command": "python bench.py --steps 3 --warmup 3 --no-cpu-baseline (two separate rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE)",
       "units": "FETCH_SIZE / WRITE_SIZE are KB per dispatch as reported by rocprofv3; gfx950 correction (MI355X_MICROARCH.md, HBM section): "
                "FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads -> doubled in hbm_corrected_bytes_per_launch",
       "gemm_hip_sha1": __import__("bench").kernel_source_hash(),
       "kernels": kern}
json.dump(out, open("$R/gpurun_out/pmc/${TAG}_pmc_traffic.json", "w"), indent=1)
for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["hbm_corrected_bytes_per_launch"] * kv[1]["launches"])[:10]:
    print(k[:70], v)
PY
