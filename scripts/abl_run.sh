#!/bin/bash
# experiment helper (GPU box): bench each experiments/abl/lib<k>.so variant in place of the library
#   bash scripts/abl_run.sh "<bench args>" k1 k2 ...
ARGS=$1; shift
cp infur_amd/libinfur_hip.so /tmp/orig.so
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/orig.so infur_amd/libinfur_hip.so; else cp experiments/abl/lib$v.so infur_amd/libinfur_hip.so; fi
  echo -n "variant $v: "
  timeout 200 python bench.py $ARGS --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'fps; conv ms', round(r['all_convs']['ms'],2), 'c3', round(r['conv3x3']['ms'],2), 'c1', round(r['conv1x1']['ms'],2))"
done
cp /tmp/orig.so infur_amd/libinfur_hip.so
