#!/bin/bash
# GPU call 13: the several-rank bench path end to end on the 1-GPU box (2 ranks share the device, gloo): both launch forms, stdout must be ONE JSON line
O=gpurun_out/r5p; mkdir -p $O
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 4 --no-cpu-baseline > $O/respawn.out 2> $O/respawn.err; echo "respawn rc=$? lines=$(wc -l < $O/respawn.out)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 5 --warmup 4 --no-cpu-baseline > $O/torchrun.out 2> $O/torchrun.err; echo "torchrun rc=$? lines=$(wc -l < $O/torchrun.out)"
for f in respawn torchrun; do python - <<PY
import json
d=json.loads(open('$O/$f.out').read().strip().splitlines()[-1])
print('$f', d['n_gpus'], d['ms_per_step'], d['value'], d['config'].get('exchange'), d['config'].get('note'))
PY
done
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
