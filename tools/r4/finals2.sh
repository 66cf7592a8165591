#!/bin/bash
# round-4 end measurements beside tools/finals.sh: Large bf16 / fp8 lines, 2-rank shared-GPU lines
O=gpurun_out/r4fin; mkdir -p $O
for dt in bf16 fp8; do timeout 300 python bench.py --arch large --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/large_$dt.json; done
for ex in phased one_message; do timeout 300 python bench.py --gpus 2 --steps 10 --warmup 5 --no-cpu-baseline --exchange $ex 2>/dev/null | grep '^{' > $O/bench_2rank_$ex.json; done
python - <<'PY'
import json
for f in ("large_bf16", "large_fp8", "bench_2rank_phased", "bench_2rank_one_message"):
    try:
        d = json.load(open(f"gpurun_out/r4fin/{f}.json")); print(f, d["ms_per_step"], d["value"], d["roofline"]["achieved"], d["roofline"]["frac"])
    except Exception as e: print(f, "missing", e)
PY
