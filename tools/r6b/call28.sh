#!/bin/bash
bash tools/finals.sh r6 2>&1 | grep -E "rc=|ms_per_step" | cut -c1-200
