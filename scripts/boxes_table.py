#!/usr/bin/env python3
"""profiles/rNN_bench_boxes.md from the JSON lines of `python bench.py` runs on different boxes:
    python scripts/boxes_table.py label=path.json [label=path.json ...]"""
import json
import sys


def line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def v(o, *keys, nd=1):
    for k in keys:
        o = (o or {}).get(k)
    return "-" if o is None else (f"{o:.{nd}f}" if isinstance(o, float) else str(o))


print("| box | f32 headline (frac of f32 MFMA peak, dominant kernel) | f32s | f32x | **f16hl** (executed frac of its 2-unit peak; PCIe-incl. copying / zero-copy) "
      "| f16 | int8 | configs[2] stream | configs[4] 4K R101 f16 (frac) | configs[3] group f32 / f16hl |")
print("|---|---|---|---|---|---|---|---|---|---|")
for arg in sys.argv[1:]:
    label, path = arg.split("=", 1)
    d = line(path)
    side = lambda k: d.get(k) or d.get("config", {}).get(k) or {}
    hl = side("f16hl_mode_1080p")
    print(f"| {label} | {v(d, 'value')} ({v(d, 'roofline', 'frac', nd=3)}) | {v(side('f32_split_mode'), 'value')} | {v(side('f32_split_fp8_mode'), 'value')} | "
          f"**{v(hl, 'value')}** ({v(hl, 'roofline', 'executed_frac', nd=3)}; {v(hl, 'pcie_inclusive_frames_per_s', nd=0)} / {v(hl, 'pcie_inclusive_zero_copy_frames_per_s', nd=0)}) | "
          f"{v(side('f16_mode_1080p'), 'value')} | {v(side('int8_quantised_model'), 'value')} | {v(side('configs2_stream_scale05'), 'value')} | "
          f"{v(side('configs4_r101_f16_4k'), 'value')} ({v(side('configs4_r101_f16_4k'), 'roofline', 'frac', nd=3)}) | "
          f"{v(side('configs3_batch64_group'), 'value')} / {v(side('configs3_batch64_group_f16hl'), 'value')} |")
