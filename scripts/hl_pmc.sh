#!/bin/bash
# PMC passes over one-context f16hl frames (run on the GPU box):   bash scripts/hl_pmc.sh [tag]
# One rocprofv3 run per counter group (--pmc never together with the trace domains gpurun refuses); the per-kernel means go to
# gpurun_out/hl_pmc_<tag>.log.  Counters: matrix-pipe busy, LDS conflicts, what the waves wait for, the texture-address path.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
TAG=${1:-r05}
OUT=gpurun_out/hl_pmc_$TAG
mkdir -p "$OUT"
CMD="python scripts/hl_check.py --time-only"
summarise='
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if not any(s in k for s in ("conv_hl", "wino_", "stem_pool")):
        continue
    print(k, " ".join(f"{c}={sum(v) / len(v):.4g}(n={len(v)})" for c, v in sorted(agg[k].items())))
'
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    d="$OUT/$(echo $grp | tr ' ' '_' | cut -c1-40)"
    timeout 600 rocprofv3 --pmc $grp -d "$d" -o pmc --output-format csv -- $CMD > "$d.stdout" 2>&1
    echo "== $grp" >> "$OUT.log"
    python -c "$summarise" "$d" >> "$OUT.log" 2>&1
done
cat "$OUT.log"
