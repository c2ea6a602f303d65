#!/usr/bin/env python3
"""Digest tools/profile_bench.sh's rocprofv3 CSVs of the ragged search (SSYM_PROFILE_PY=tools/ragged_profile_cmd.py) into
profiles/<name>.{md,json}: the filter runs as one launch per class of source lengths (dtw_filter_sp_kernel<1|2|3 tiles>), so
the figures are kept per class and summed per search.
usage: python tools/summarize_ragged_profile.py gpurun_out/<tag> profiles/<name>"""
import collections, csv, glob, json, os, re, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import synth

src_dir, dst = sys.argv[1], sys.argv[2]
cmd = open(os.path.join(src_dir, "command.txt")).read().strip()
n, lo, hi = 4096, 5, 40
ls = lo + synth.Stream(0x5EED0A28).integers(n, hi - lo + 1)          # the lengths make_ragged draws first
st = synth.Stream(0x5EED0A28); _ = st.integers(n, hi - lo + 1); lt = lo + st.integers(n, hi - lo + 1)
sum_t = float(lt.sum())
true_cells = {1: float(ls[ls <= 16].sum()) * sum_t, 2: float(ls[(ls > 16) & (ls <= 32)].sum()) * sum_t,
              3: float(ls[ls > 32].sum()) * sum_t}

def cls(name):
    # <NT, SQ, OCC, KU, G, MP>: the multi-pair instantiation (MP = true, three one-tile pairs per wave) is the one-tile class
    m = re.search(r"dtw_filter_sp_kernelILi(\d)ELb[01]ELi\dELi\dELi\d+ELb([01])", name)
    if not m:
        return None
    return 1 if m.group(2) == "1" else int(m.group(1))

stats = []
for f in glob.glob(os.path.join(src_dir, "trace", "**", "*kernel_stats.csv"), recursive=True):
    stats += list(csv.DictReader(open(f)))
pm = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds", "pmc_misc"):
    for f in glob.glob(os.path.join(src_dir, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = cls(r["Kernel_Name"])
            if c:
                pm[c][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": src_dir, "command": cmd, "classes": {}, "kernels": []}
lines = ["# rocprofv3 summary (%s)" % os.path.basename(dst), "",
         "Command: `%s` on 1 MI355X (tools/profile_bench.sh: one --kernel-trace --stats pass, separate --pmc passes): "
         "bench.py's `secondary.ragged.dtw` workload, 4096 x 4096 segments of 5...40 frames x 13 dims (synth.make_ragged, seed "
         "0x5EED0A28), 12 searches." % cmd, "", "## kernel trace (--kernel-trace --stats)", "",
         "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
for r in sorted(stats, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    k = {"name": r["Name"], "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
         "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])}
    out["kernels"].append(k)
    lines.append("| `%s` | %d | %.3f | %.4f | %.2f |" % (k["name"][:88], k["calls"], k["total_ms"], k["avg_ms"], k["pct"]))
lines += ["", "## the filter, per class of source lengths (one launch each per search; PMC means per launch)", "",
          "| tiles (source frames) | avg ms | true cells | T true cells/s | VALU busy | clock GHz | VALU instr / true cell | "
          "VALU-busy / elapsed SIMD cycles per true wave-cell | wait_any / wave cycles | HBM MB |", "|---|---|---|---|---|---|---|---|---|---|"]
tot_ms = tot_cells = tot_hbm = 0.0
for c in (1, 2, 3):
    ms = [float(r["AverageNs"]) / 1e6 for r in stats if cls(r["Name"]) == c]
    if not ms:
        continue
    ms = ms[0]
    v = {k: sum(x) / len(x) for k, x in pm[c].items()}
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    busy = 4.0 * v["SQ_ACTIVE_INST_VALU"] / (cyc * 1024.0)
    ghz = cyc / (ms * 1e6)
    wc = true_cells[c] / 64.0
    hbm = (v.get("FETCH_SIZE", 0) * 2 + v.get("WRITE_SIZE", 0)) * 1024 / 1e6
    out["classes"][str(c)] = {"avg_ms": ms, "true_cells": true_cells[c], "true_cells_per_s": true_cells[c] / ms * 1e3,
                              "valu_busy_fraction": busy, "clock_ghz": ghz, "valu_instr_per_true_cell": v["SQ_INSTS_VALU"] / wc,
                              "valu_busy_cycles_per_true_wave_cell": 4.0 * v["SQ_ACTIVE_INST_VALU"] / wc,
                              "simd_cycles_per_true_wave_cell": cyc * 1024.0 / wc,
                              "wait_any_over_wave_cycles": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], "hbm_mb": hbm,
                              "pmc": v}
    lines.append("| %d (%s) | %.4f | %.3g | %.2f | %.3f | %.2f | %.2f | %.1f / %.1f | %.3f | %.1f |" % (
        c, {1: "5...16", 2: "17...32", 3: "33...40"}[c], ms, true_cells[c], true_cells[c] / ms / 1e9, busy, ghz,
        v["SQ_INSTS_VALU"] / wc, 4.0 * v["SQ_ACTIVE_INST_VALU"] / wc, cyc * 1024.0 / wc,
        v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], hbm))
    tot_ms += ms; tot_cells += true_cells[c]; tot_hbm += hbm
out["filter_ms_per_search"] = tot_ms
out["true_cells_per_s"] = tot_cells / tot_ms * 1e3
out["valu_frac_true_cells"] = out["true_cells_per_s"] / 9.8304e12
out["hbm_traffic_bytes_per_search"] = tot_hbm * 1e6
lines += ["", "Per search: the three launches take **%.3f ms** for %.3g true cells (sum of source frames x sum of target frames) = "
          "**%.2f T true cells/s = %.3f of the 16-cycle cell model at 2.4 GHz** (9.83e12 cells/s); fabric traffic %.1f MB per search "
          "(FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024; compulsory: 67 MB of cost matrix written, ~10 MB of records read)." % (
              tot_ms, tot_cells, tot_cells / tot_ms / 1e9, out["valu_frac_true_cells"], tot_hbm)]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
open(dst + ".md", "w").write("\n".join(lines) + "\n")
json.dump(out, open(dst + ".json", "w"), indent=1)
print("\n".join(lines))
