#!/bin/bash
# A/B on ONE GPU box: bench every experiments/ab/<name>.so in place of the library, ROUNDS alternations
#   bash scripts/ab_run.sh "<bench args>" name1 name2 ...     (box-to-box variation is +-3 %, same-box +-0.5 %)
ARGS=$1; shift
cp infur_amd/libinfur_hip.so /tmp/orig.so
for r in 1 2; do
  for v in "$@"; do
    cp experiments/ab/$v.so infur_amd/libinfur_hip.so
    echo -n "$v: "
    timeout 200 python bench.py $ARGS --no-split --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],2))"
  done
done
cp /tmp/orig.so infur_amd/libinfur_hip.so
