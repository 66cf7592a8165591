#!/bin/bash
# one GPU call, several knobs, each against the default on the same box (one round each; the default is re-measured between knobs)
bash tools/r4/bench_env_ab.sh abk ST5_NT_TILE 1 - 1
bash tools/r4/bench_env_ab.sh abk ST5_DEEP_RING 1 - 0,2 512,4 256,3
bash tools/r4/bench_env_ab.sh abk ST5_LN_MAX_BLOCKS 1 - 512 1024
bash tools/r4/bench_env_ab.sh abk ST5_NT_SLOTS 1 - 5
