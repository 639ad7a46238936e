#!/bin/bash
# where the LDS bank conflicts of stem_pool_kernel come from: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per launch under the
# timing ablations (INFUR_STEM_ABL: 1 no MFMA loop, 2 no prologue, 3 no stage + pool)   -> gpurun_out/stem_conf/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/stem_conf
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for A in 0 1 2 3; do
  INFUR_STEM_ABL=$A timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/a$A -- \
    python $R/bench.py --no-cpu-baseline --no-profile --no-side --no-split --steps 1 --warmup 1 --frames-per-step 1 --contexts-per-gpu 1 > /dev/null 2> $OUT/a$A.err
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/a$A/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stem_pool" in r["Kernel_Name"]:
            k = r["Counter_Name"]
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
print("ABL=$A", {k: round(v[1] / max(v[0], 1)) for k, v in acc.items()}, "launches", max([v[0] for v in acc.values()] + [0]))
PY
done
