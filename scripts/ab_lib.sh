#!/bin/bash
# A/B of library builds on one box: scripts/ab_lib.sh <dtype> libA.so libB.so ... -> frames/s (3 contexts) and one-context frame kernels per build
dt=$1; shift
for lib in "$@"; do
  r=$(INFUR_LIB_PATH=$lib python bench.py --dtype $dt --no-side --no-split --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))")
  k=$(INFUR_LIB_PATH=$lib python bench.py --dtype $dt --contexts-per-gpu 1 --no-side --no-split --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['frame_kernel_ms'],3), round(d['value'],1))")
  echo "$lib: $r frames/s (3 contexts); one context: frame kernels ms, frames/s = $k"
done
