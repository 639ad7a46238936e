#!/bin/bash
# per-kernel table of one bench run (stderr of --kernels), aggregated by kernel family
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-split --no-side --kernels "$@" 2>&1 >/dev/null | awk '{k=$2; ms[k]+=$3; n[k]++} END {for (k in ms) printf "%-32s %3d launches %8.3f ms\n", k, n[k], ms[k]}' | sort
