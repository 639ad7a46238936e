#!/bin/bash
# per-layer times with every layer forced to one tile configuration (where it is a candidate)
#   bash scripts/cfg_sweep.sh f32s 0 5 11 12 15  -> gpurun_out/sweep_<dtype>_<cfg>.txt
DT=$1; shift
for c in "$@"; do
  INFUR_CONV_CFG=$c timeout 200 python bench.py --dtype $DT --kernels --no-cpu-baseline > gpurun_out/sweep_${DT}_$c.txt 2>&1
  echo "cfg $c: $(tail -1 gpurun_out/sweep_${DT}_$c.txt | cut -c85-110)"
done
