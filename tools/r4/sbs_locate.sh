#!/bin/bash
# which parameters differ when a replayed side-by-side update goes wrong?  clipping off (a wrong gradient only moves its own
# parameter), per-parameter hashes dumped per run and compared with the in-turn reference
O=gpurun_out/${1:-r4f}; mkdir -p $O; N=${2:-4}; REPS=${3:-10}
DIAG_CLIP=0 DIAG_DUMP=$O/ref.json timeout 120 python tools/one_run.py graph in_turn $N 0.05 2>/dev/null | grep RESULT
for i in $(seq $REPS); do
  DIAG_CLIP=0 DIAG_DUMP=$O/run$i.json timeout 120 python tools/one_run.py graph side_by_side $N 0.05 2>/dev/null | grep RESULT
done
python - $O $REPS <<'PY'
import json, sys
o, n = sys.argv[1], int(sys.argv[2])
ref = json.load(open(f"{o}/ref.json"))
for i in range(1, n + 1):
    try: r = json.load(open(f"{o}/run{i}.json"))
    except Exception as e: print(i, "missing", e); continue
    bad = [k for k in ref if r.get(k) != ref[k]]
    print(f"run {i}: {len(bad)} of {len(ref)} parameters differ", bad[:12])
PY
