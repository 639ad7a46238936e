#!/bin/bash
# Everything profiles/ needs for one round, on the GPU box:   bash scripts/profile_round.sh r05
# -> gpurun_out/profiles_<tag>/ (summaries, kernel stats, traffic files, bench lines under rocprof) to be copied into profiles/.
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-r05}
bash scripts/profile_all.sh
python scripts/profile_summary.py gpurun_out/prof ${TAG} f32 > /dev/null
for m in f32s f32x f16hl f16 i8; do python scripts/profile_summary.py gpurun_out/prof_$m ${TAG}_$m $m > /dev/null; done
python scripts/profile_summary.py gpurun_out/prof_f16_4k ${TAG}_f16_4k_r101 f16_4k > /dev/null
python scripts/profile_summary.py gpurun_out/prof_scale05 ${TAG}_scale05 scale05 > /dev/null
mkdir -p gpurun_out/profiles_${TAG}
cp profiles/${TAG}_* profiles/traffic_*.json gpurun_out/profiles_${TAG}/ 2>/dev/null
for d in gpurun_out/prof gpurun_out/prof_*; do [ -f $d/smi.csv ] && cp $d/smi.csv gpurun_out/profiles_${TAG}/$(basename $d | sed "s/^prof/${TAG}/")_smi.csv; done
rm -rf gpurun_out/prof gpurun_out/prof_*   # (the raw traces stay on the box: 64 MiB merge limit)
ls gpurun_out/profiles_${TAG} | wc -l
