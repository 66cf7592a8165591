#!/bin/bash
O=gpurun_out/r5s; mkdir -p $O
H="python tools/r5/replay_hunt.py"; G='HUNT ran|DIFF|Error|error|assert|differ|identical'
timeout 300 $H run $O/ref.json --n 1000 2>&1 | grep -E "$G"
timeout 300 $H run $O/sbs1.json --n 1000 --mode side_by_side 2>&1 | grep -E "$G"; $H diff $O/ref.json $O/sbs1.json
timeout 300 $H run $O/sbs2.json --n 1000 --mode side_by_side --sync 1 2>&1 | grep -E "$G"; $H diff $O/ref.json $O/sbs2.json
