#!/bin/bash
# A/B of the f16hl K loop: INFUR_HL_PIPE=0 (plain) vs 1 (software-pipelined across the barrier), same box, per-layer HIP-event times
mkdir -p gpurun_out/pipe
for p in 0 1; do
  INFUR_HL_PIPE=$p python bench.py --dtype f16hl --contexts-per-gpu 1 --kernels --no-side --no-split --no-cpu-baseline > gpurun_out/pipe/out_$p.txt 2> gpurun_out/pipe/layers_$p.txt
  INFUR_HL_PIPE=$p python bench.py --dtype f16hl --no-side --no-split --no-cpu-baseline --no-profile 2>/dev/null | tail -1 > gpurun_out/pipe/rate_$p.txt
done
python - <<'PY'
import re
def load(p):
    d=[]
    for ln in open(p):
        m=re.match(r"(\S+)\s+(\S+)\s+([\d.]+) ms",ln)
        if m: d.append((m.group(1),m.group(2),float(m.group(3))))
    return d
a,b=load('gpurun_out/pipe/layers_0.txt'),load('gpurun_out/pipe/layers_1.txt')
ta=tb=0
for (n,k,x),(n2,k2,y) in zip(a,b):
    ta+=x;tb+=y
    if abs(x-y)>0.003: print(f"{n:40s} {k:22s} {x*1e3:7.1f} -> {y*1e3:7.1f} us  {k2 if k2!=k else ''}")
print("frame kernels ms", ta, tb)
import json
for p in (0,1):
    print(p, json.loads(open(f'gpurun_out/pipe/rate_{p}.txt').read())['value'])
PY
