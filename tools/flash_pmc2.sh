#!/bin/bash
# Fused-attention kernels (fa2::*) under rocprofv3: kernel durations and SQ counters (MFMA busy, VALU, LDS, waits), with and without the
# relative-position bias, at the text micro-batch's shape (B 16, H 12, T 512, hd 64, dropout 0.1)  -> gpurun_out/flpmc/TAG_flash_pmc.json
# Counter passes are separate from the trace pass (gpurun refuses them combined).  Nothing here reads stdin.
TAG=${1:-r3}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for rel in 1 0; do
  rm -rf /tmp/fl_kt$rel /tmp/fl_a$rel /tmp/fl_b$rel
  REL=$rel timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fl_kt$rel -- python $R/tools/flash_pmc.py > /dev/null 2>&1 < /dev/null
  REL=$rel timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d /tmp/fl_a$rel -- python $R/tools/flash_pmc.py > /dev/null 2>&1 < /dev/null
  REL=$rel timeout 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_WAIT_INST_LDS --output-format csv -d /tmp/fl_b$rel -- python $R/tools/flash_pmc.py > /dev/null 2>&1 < /dev/null
done
mkdir -p $R/gpurun_out/flpmc
python - <<PY
import csv, glob, collections, json
out = {"shape": "B 16, H 12, T = S = 512, hd 64, dropout 0.1, bf16; 3 launches of each kernel per pass; counters summed over all SEs / XCDs, per launch",
       "flops_per_launch": {"fwd": 4 * 512 * 512 * 64 * 192, "bwd_dq": 6 * 512 * 512 * 64 * 192, "bwd_dkv": 8 * 512 * 512 * 64 * 192}}
for rel in (1, 0):
    res = collections.defaultdict(dict)
    fs = glob.glob(f"/tmp/fl_kt{rel}/**/*kernel_stats.csv", recursive=True)
    if fs:
        for row in csv.DictReader(open(fs[0])):
            if "fa2::" in row["Name"]:
                res[row["Name"][:48]]["avg_us"] = round(float(row["AverageNs"]) / 1e3, 1)
    for d in (f"/tmp/fl_a{rel}", f"/tmp/fl_b{rel}"):
        fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not fs:
            res["_missing"][d] = True
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(fs[0])):
            k = row["Kernel_Name"][:48]
            if "fa2::" not in k: continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_MFMA"): n[k] += 1
        for k, v in acc.items():
            for c, x in v.items():
                res[k][c] = round(x / max(n[k], 1))
    out["bias" if rel else "no_bias"] = res
json.dump(out, open("$R/gpurun_out/flpmc/${TAG}_flash_pmc.json", "w"), indent=1)
for part in ("bias", "no_bias"):
    for k, v in out[part].items():
        print(part, k, v)
PY
