#!/bin/bash
# PCIe-inclusive frames/s of the native front end (tools/infur_pipeline.cpp) on a raw bgr24 clip held in the page cache:
# BASELINE configs[2] (1080p stream, scale 0.5) and scale 1.0, one and two lanes.  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python - <<'PY'
import numpy as np
from infur_amd import weights as W
open("/dev/shm/fcn50.infurw", "wb").write(W.synth_blob())
with open("/dev/shm/clip1080.bgr", "wb") as f:
    fr = [W.synth_frame(1080, 1920, index=i) for i in range(8)]
    for i in range(160):
        f.write(fr[i % 8].tobytes())
PY
for DT in f32 f32s; do
  for SC in 0.5 1.0; do
    for LANES in 1 2; do
      echo -n "dtype $DT scale $SC lanes $LANES: "
      ./infur_amd/infur_pipeline --width 1920 --height 1080 --scale $SC --model /dev/shm/fcn50.infurw --dtype $DT --lanes $LANES --depth 4 \
        --input /dev/shm/clip1080.bgr --output none 2>&1 | tail -1
    done
  done
done
rm -f /dev/shm/clip1080.bgr /dev/shm/fcn50.infurw
