set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fl_kt -- python $R/tools/flash_pmc.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d /tmp/fl_pmc -- python $R/tools/flash_pmc.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/fl_pmc2 -- python $R/tools/flash_pmc.py > /dev/null 2>&1 || true
mkdir -p $R/gpurun_out/flpmc
cp $(find /tmp/fl_kt -name "*kernel_stats.csv") $R/gpurun_out/flpmc/kernel_stats.csv
python - <<PY
import csv, glob, collections
for d in ("/tmp/fl_pmc", "/tmp/fl_pmc2"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print("no counter file in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in acc.items():
        if "flash" in k: print(k, {a: round(b / 3) for a, b in v.items()})
PY
