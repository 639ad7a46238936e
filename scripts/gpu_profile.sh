#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun).
#   bash scripts/gpu_profile.sh [dtype] [tag] [extra bench.py args...]  -> gpurun_out/prof[_dtype][_tag]/{trace,pmc_*,smi.csv}
#   e.g. bash scripts/gpu_profile.sh f16 4k --depth 101 --width 3840 --height 2160   (BASELINE configs[4])
# --pmc passes are separate runs with --kernel-trace only (gpurun refuses other combinations).
R=${GRAFT_REPO_ROOT:-/root/repo}
DT=${1:-f32}
TAG=${2:-}
shift; shift
EXTRA="$@"
OUT=$R/gpurun_out/prof$([ $DT = f32 ] || echo _$DT)$([ -z "$TAG" ] || echo _$TAG)
rm -rf $OUT; mkdir -p $OUT
# power / clock trace next to the counters (rocm-smi once per 200 ms while the trace run is alive): the evidence behind
# any "the package clocks down under f16 MFMA load" statement in DESIGN.md
( while true; do rocm-smi --showpower --showclocks --csv 2>/dev/null | tail -n +2 | sed "s/^/$(date +%s.%N),/"; sleep 0.2; done ) > $OUT/smi.csv &
SMI=$!
cd /tmp && export TMPDIR=/tmp
# frames in flight of the "default" trace: what bench.py itself uses for this arithmetic (3; F16_CONTEXTS = 2 for the f16 mode's side object)
CTX=3; [ $DT = f16 ] && CTX=2
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --dtype $DT --contexts-per-gpu $CTX --no-split --no-side --no-cpu-baseline $EXTRA > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
kill $SMI 2>/dev/null
# The default command keeps several frames in flight (--contexts-per-gpu, default 3): kernels of the streams time-share the chip, so the
# trace above gives per-kernel wall time UNDER SHARING.  The solo trace (one context, strictly one kernel at a time) is
# the one whose average durations must agree with the HIP events of bench.py's roofline frame, which also runs alone.
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_solo -- python $R/bench.py --dtype $DT --contexts-per-gpu 1 --no-split --no-side --no-cpu-baseline $EXTRA > $OUT/bench_trace_solo.json 2> $OUT/trace_solo.err
echo "solo trace rc=$?"
SMALL="--dtype $DT --contexts-per-gpu 1 --no-split --no-side --no-cpu-baseline --no-profile --steps 1 --warmup 1 --frames-per-step 2 $EXTRA"
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_$P -- python $R/bench.py $SMALL > /dev/null 2> $OUT/pmc_$P.err; echo "pmc $P rc=$?"
done
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_GRBM -- python $R/bench.py $SMALL > /dev/null 2> $OUT/pmc_GRBM.err; echo "pmc GRBM rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_SQ -- python $R/bench.py $SMALL > /dev/null 2> $OUT/pmc_SQ.err; echo "pmc SQ rc=$?"
du -sh $OUT
