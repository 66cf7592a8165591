#!/bin/bash
O=gpurun_out/${1:-r4i}; mkdir -p $O; N=${2:-250}; CLIP=${3:-0}; TRIES=${4:-4}
timeout 300 python tools/r4/sbs_hunt.py record in_turn $N $CLIP $O/ref.json 2>&1 | grep -E "HUNT|Error|error" 
timeout 300 python tools/r4/sbs_hunt.py check in_turn $N $CLIP $O/ref.json $O/self.json 2>&1 | grep -E "HUNT|Error|error"
for i in $(seq $TRIES); do
  timeout 300 python tools/r4/sbs_hunt.py check side_by_side $N $CLIP $O/ref.json $O/bad$i.json 2>&1 | grep -E "HUNT|Error|error"
  if [ -f $O/bad$i.json ]; then
    k=$(python -c "import json;print(json.load(open('$O/bad$i.json'))['k'])")
    timeout 300 python tools/r4/sbs_hunt.py dump in_turn $k $CLIP $O/good$i.json 2>&1 | grep -E "HUNT"
    python - $O/good$i.json $O/bad$i.json <<'PY'
import json, sys
g, b = json.load(open(sys.argv[1]))["digests"], json.load(open(sys.argv[2]))["digests"]
bad = [k for k in g if g[k] != b[k]]
print(f"  {len(bad)} of {len(g)} tensors differ:", bad[:40])
PY
  fi
done
