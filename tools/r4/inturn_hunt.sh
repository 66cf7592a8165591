#!/bin/bash
# the same hunt for the DEFAULT mode: is the replayed in-turn update reproducible across processes, update by update?
O=gpurun_out/${1:-r4j}; mkdir -p $O; N=${2:-300}; CLIP=${3:-0}; TRIES=${4:-3}; MODE=${5:-in_turn}
timeout 300 python tools/r4/sbs_hunt.py record $MODE $N $CLIP $O/ref.json 2>&1 | grep -E "HUNT|Error|error"
for i in $(seq $TRIES); do
  timeout 300 python tools/r4/sbs_hunt.py check $MODE $N $CLIP $O/ref.json $O/bad$i.json 2>&1 | grep -E "HUNT|Error|error"
  if [ -f $O/bad$i.json ]; then
    k=$(python -c "import json;print(json.load(open('$O/bad$i.json'))['k'])")
    timeout 300 python tools/r4/sbs_hunt.py dump $MODE $k $CLIP $O/good$i.json 2>&1 | grep -E "HUNT"
    python - $O/good$i.json $O/bad$i.json <<'PY'
import json, sys
g, b = json.load(open(sys.argv[1]))["digests"], json.load(open(sys.argv[2]))["digests"]
bad = [k for k in g if g[k] != b[k]]
print(f"  {len(bad)} of {len(g)} tensors differ:", bad[:12])
PY
  fi
done
