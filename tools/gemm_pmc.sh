R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/g_kt /tmp/g_pmc /tmp/g_pmc2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/g_kt -- python $R/tools/gemm_pmc.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d /tmp/g_pmc -- python $R/tools/gemm_pmc.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/g_pmc2 -- python $R/tools/gemm_pmc.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/g_kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"])
for d in ("/tmp/g_pmc", "/tmp/g_pmc2"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(fs[0])):
        acc[row["Kernel_Name"][:50]][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in acc.items():
        if "gemm" in k or "splitk" in k: print(k, {a: round(b / 5) for a, b in v.items()})
PY
