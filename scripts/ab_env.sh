#!/bin/bash
# A/B of an environment switch on one box: scripts/ab_env.sh <dtype> VAR v0 v1 ... -> frames/s (3 contexts) and one-context frame kernels per value
dt=$1; var=$2; shift 2
for v in "$@"; do
  r=$(env $var=$v python bench.py --dtype $dt --no-side --no-split --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))")
  k=$(env $var=$v python bench.py --dtype $dt --contexts-per-gpu 1 --no-side --no-split --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['frame_kernel_ms'],3), round(d['value'],1))")
  echo "$var=$v: $r frames/s (3 contexts); one context: frame kernels ms, frames/s = $k"
done
