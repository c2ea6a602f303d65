#!/usr/bin/env python3
"""Digest the rocprofv3 CSVs written by tools/profile_bench.sh into profiles/<name>.{md,json}.

usage: python tools/summarize_profile.py gpurun_out/<tag> profiles/<name>
HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB, collected
in separate --pmc passes; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes for wide
(16 B/lane) reads, so the read side is doubled.
"""
import collections
import csv
import glob
import json
import os
import sys


def kernel_stats(d):
    rows = []
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def pmc(d, sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    src, dst = sys.argv[1], sys.argv[2]
    stats = kernel_stats(src)
    out = {"source": src, "kernels": [], "pmc": {}}
    lines = ["# rocprofv3 summary (%s)" % os.path.basename(dst), "",
             "Command: `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary` "
             "(4096x4096 segments, 128 frames x 13 dims, 1 MI355X).", "",
             "## kernel trace (--kernel-trace --stats)", "",
             "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
    for r in sorted(stats, key=lambda r: -float(r["TotalDurationNs"])):
        k = {"name": r["Name"], "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
             "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])}
        out["kernels"].append(k)
        lines.append("| `%s` | %d | %.3f | %.4f | %.2f |" % (k["name"][:90], k["calls"], k["total_ms"], k["avg_ms"], k["pct"]))
    lines += ["", "## PMC (separate --pmc passes), per launch of the dominant kernel", ""]
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_misc"):
        for kern, ctrs in pmc(src, sub).items():
            if "dtw_filter_kernel" not in kern:
                continue
            for c, vals in ctrs.items():
                out["pmc"][c] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
    p = out["pmc"]
    for c, v in p.items():
        lines.append("- `%s` = %.6g (mean over %d launches)" % (c, v["mean_per_launch"], v["launches"]))
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        rd = p["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2      # KiB, x2 gfx950 correction
        wr = p["WRITE_SIZE"]["mean_per_launch"] * 1024
        out["hbm_traffic_bytes_per_launch"] = rd + wr
        out["hbm_read_bytes_per_launch"] = rd
        out["hbm_write_bytes_per_launch"] = wr
        lines += ["", "HBM traffic per launch of `dtw_filter_kernel`: read %.1f MB (FETCH_SIZE KiB x 1024 x 2, "
                  "gfx950 correction) + write %.1f MB (WRITE_SIZE KiB x 1024) = **%.1f MB**." % (rd / 1e6, wr / 1e6, (rd + wr) / 1e6)]
    if "SQ_ACTIVE_INST_VALU" in p and "SQ_WAVE_CYCLES" in p:
        valu = p["SQ_ACTIVE_INST_VALU"]["mean_per_launch"]
        wave = p["SQ_WAVE_CYCLES"]["mean_per_launch"]
        insts = p["SQ_INSTS_VALU"]["mean_per_launch"]
        out["valu_active_over_wave_cycles"] = valu / wave
        out["cycles_per_valu_inst"] = 4.0 * valu / insts
        lines += ["", "VALU active / wave cycles = %.3f; cycles per VALU instruction = %.2f (quad-cycle counters x 4)."
                  % (valu / wave, 4.0 * valu / insts)]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in p and "SQ_BUSY_CYCLES" in p:
            lines.append("MFMA busy cycles per launch = %.4g; SQ busy cycles = %.4g." %
                         (p["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"], p["SQ_BUSY_CYCLES"]["mean_per_launch"]))
    if "GRBM_GUI_ACTIVE" in p and "SQ_ACTIVE_INST_VALU" in p:
        # GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles; 1024 SIMDs; SQ counters are in quad-cycles
        k = [k for k in out["kernels"] if "dtw_filter_kernel" in k["name"]]
        cyc = p["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0
        busy = 4.0 * p["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / (cyc * 1024.0)
        out["valu_busy_fraction"] = busy
        line = "Per-SIMD VALU busy fraction of the dominant kernel = %.3f (4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))" % busy
        if k:
            ghz = cyc / (k[0]["avg_ms"] * 1e6)
            out["clock_ghz_under_load"] = ghz
            cells = 4096.0 * 4096.0 * 128.0 * 128.0 / 64.0
            line += "; clock under load %.2f GHz; %.2f VALU-busy cycles and %.2f elapsed SIMD cycles per wave-cell (floor: 16)" % (
                ghz, 4.0 * p["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / cells, cyc * 1024.0 / cells)
        lines += ["", line + "."]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in p:
            # MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 MFMA), summed over the SIMDs
            mf = p["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / (cyc * 1024.0)
            out["mfma_busy_fraction"] = mf
            lines += ["", "MFMA utilisation of the dominant kernel = %.3f of the matrix pipes' cycles (SQ_VALU_MFMA_BUSY_CYCLES / "
                      "(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); three 32x32x16 f16 MFMAs per 32x32 tile of cells): the matrix pipe runs beside the VALU, "
                      "which sets the pace." % mf]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    open(dst + ".md", "w").write("\n".join(lines) + "\n")
    json.dump(out, open(dst + ".json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
