#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/...) into the tracked files under profiles/.

    python scripts/profile_summary.py gpurun_out/prof r01 [dtype]

Inputs (any subset):
  <dir>/trace/**/_kernel_stats.csv        from  rocprofv3 --kernel-trace --stats -- python bench.py
  <dir>/pmc_FETCH_SIZE/**/counter_collection.csv, <dir>/pmc_WRITE_SIZE/**  (separate --pmc passes)
  <dir>/pmc_SQ/** , <dir>/pmc_GRBM/**                                        (utilisation passes)
Outputs: profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.md, profiles/traffic_latest.json
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def find(d, pat):
    g = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return max(g, key=os.path.getmtime) if g else None  # newest (gpurun merges, never deletes)


_DEMANGLED = {}


def short(name):
    if name.startswith("_Z"):  # rocprofv3 leaves some template instantiations mangled
        if name not in _DEMANGLED:
            try:
                import subprocess

                # binutils' demangler does not know DF16_ (_Float16): go through Dh (half) and rename
                out = subprocess.run(["c++filt", name.replace("DF16_", "Dh")], capture_output=True, text=True, timeout=10).stdout.strip()
                _DEMANGLED[name] = out.replace("half", "_Float16") if out and not out.startswith("_Z") else name
            except (OSError, subprocess.SubprocessError):
                _DEMANGLED[name] = name
        name = _DEMANGLED[name]
    n = name.replace("(anonymous namespace)::", "").replace("void infur::", "").replace("infur::", "")
    return n.split("(")[0]


def pmc(d):
    f = find(d, "*counter_collection.csv")
    if not f:
        return {}
    by = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        by[short(r["Kernel_Name"])][k][r["Counter_Name"]] = float(r["Counter_Value"])
        by[short(r["Kernel_Name"])][k]["_dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return by


def main():
    d, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    md = [f"# rocprofv3 summary {tag}", ""]
    ks = find(os.path.join(d, "trace"), "*kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(out, f"{tag}_kernel_stats.csv"))
        nctx = 3
        try:
            nctx = int(json.loads(open(os.path.join(d, "bench_trace.json")).read().strip().splitlines()[-1])["config"].get("contexts_per_gpu", 3))
        except Exception:  # noqa: BLE001
            pass
        md += [f"## kernel trace (`rocprofv3 --kernel-trace --stats -- python bench.py --contexts-per-gpu {nctx} --no-cpu-baseline`: {nctx} frames in flight)", "",
               "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
        for r in csv.DictReader(open(ks)):
            md.append(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | "
                      f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
        md.append("")
        bj = os.path.join(d, "bench_trace.json")
        if os.path.exists(bj):
            shutil.copy(bj, os.path.join(out, f"{tag}_bench_under_rocprof.json"))
    ks2 = find(os.path.join(d, "trace_solo"), "*kernel_stats.csv")
    if ks2:
        shutil.copy(ks2, os.path.join(out, f"{tag}_solo_kernel_stats.csv"))
        md += ["## kernel trace, one context (`… bench.py --contexts-per-gpu 1`): one kernel at a time -- these average durations are the ones",
               "that must agree with the HIP-event durations of bench.py's roofline frame (which runs alone); the table above is the",
               "command the bench line is measured with (several frames in flight, see its caption), where kernels of the streams time-share the chip", "",
               "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
        for r in csv.DictReader(open(ks2)):
            md.append(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | "
                      f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
        md.append("")
        bj = os.path.join(d, "bench_trace_solo.json")
        if os.path.exists(bj):
            shutil.copy(bj, os.path.join(out, f"{tag}_solo_bench_under_rocprof.json"))
    traffic = {}
    fe, wr = pmc(os.path.join(d, "pmc_FETCH_SIZE")), pmc(os.path.join(d, "pmc_WRITE_SIZE"))
    if fe or wr:
        md += ["## HBM traffic per launch (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes)", "",
               "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their",
               "bytes (MI355X_MICROARCH.md, HBM section), so read bytes = 2 x FETCH_SIZE x 1024.", "",
               "| kernel | launches | avg read MB (2xFETCH) | avg write MB | avg us |", "|---|---|---|---|---|"]
        for k in sorted(set(fe) | set(wr)):
            f = [v.get("FETCH_SIZE", 0.0) for v in fe.get(k, {}).values()]
            w = [v.get("WRITE_SIZE", 0.0) for v in wr.get(k, {}).values()]
            du = [v["_dur"] for v in fe.get(k, {}).values()]
            rd = 2.0 * 1024.0 * sum(f) / max(len(f), 1)
            wb = 1024.0 * sum(w) / max(len(w), 1)
            traffic[k] = {"launches": max(len(f), len(w)), "read_bytes_per_launch": rd, "write_bytes_per_launch": wb}
            md.append(f"| `{k}` | {max(len(f), len(w))} | {rd / 1e6:.1f} | {wb / 1e6:.1f} | {sum(du) / max(len(du), 1) / 1e3:.1f} |")
        md.append("")
        dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
        with open(os.path.join(out, "traffic_latest.json" if dtype == "f32" else f"traffic_{dtype}.json"), "w") as fjs:
            json.dump({"tag": tag, "note": "bytes per launch, averaged over all launches of the kernel in one frame pass; "
                       "read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB", "kernels": traffic}, fjs, indent=1)
    sq, gr = pmc(os.path.join(d, "pmc_SQ")), pmc(os.path.join(d, "pmc_GRBM"))
    if gr:
        md += ["## MFMA utilisation (`--pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES`)", "",
               "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); GRBM_GUI_ACTIVE is summed over the 8 XCDs.",
               "clock = GRBM_GUI_ACTIVE / 8 / duration.  A clock ABOVE 2.4 GHz (the part's maximum) is an artefact of short kernels: GRBM_GUI_ACTIVE",
               "keeps counting through launch / drain time that the kernel's own duration does not contain; read such rows' MfmaUtil as a lower bound.",
               "", "| kernel | launches | MfmaUtil % (time-weighted) | clock GHz |", "|---|---|---|---|"]
        for k, disp in sorted(gr.items()):
            num = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in disp.values())
            den = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in disp.values()) / 8.0 * 1024.0
            clk = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in disp.values()) / 8.0 / max(sum(v["_dur"] for v in disp.values()), 1)
            if den > 0 and num > 0:
                md.append(f"| `{k}` | {len(disp)} | {100.0 * num / den:.1f} | {clk:.2f}{' (artefact)' if clk > 2.45 else ''} |")
        md.append("")
    if sq:
        md += ["## SQ wave states (`--pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT`)", "",
               "| kernel | wait_inst % | wait_any % | active % | LDS bank conflict cycles |", "|---|---|---|---|---|"]
        for k, disp in sorted(sq.items()):
            wc = sum(v.get("SQ_WAVE_CYCLES", 0.0) for v in disp.values())
            if wc <= 0:
                continue
            g = lambda n: 100.0 * sum(v.get(n, 0.0) for v in disp.values()) / wc  # noqa: E731
            md.append(f"| `{k}` | {g('SQ_WAIT_INST_ANY'):.1f} | {g('SQ_WAIT_ANY'):.1f} | {g('SQ_ACTIVE_INST_ANY'):.1f} | "
                      f"{sum(v.get('SQ_LDS_BANK_CONFLICT', 0.0) for v in disp.values()):.0f} |")
        md.append("")
    with open(os.path.join(out, f"{tag}_summary.md"), "w") as f:
        f.write("\n".join(md))
    print("\n".join(md))


if __name__ == "__main__":
    main()
