#!/usr/bin/env python3
"""Digest the rocprofv3 CSVs written by tools/profile_bench.sh into profiles/<name>.{md,json}.

usage: python tools/summarize_profile.py gpurun_out/<tag> profiles/<name> [kernel-substring] [wave-units per launch] [unit name]
  kernel-substring   the dominant kernel the PMC figures are reported for (default dtw_filter_kernel)
  wave-units         algorithmic work of one launch in 64-lane units (DP cells / 64, MACs / 64 ...): cycles per unit are
                     reported against it (default: the headline's 4096 x 4096 x 128 x 128 / 64 wave-cells)
HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB, collected
in separate --pmc passes; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes for wide
(16 B/lane) reads, so the read side is doubled.
"""
import collections
import csv
import glob
import json
import os
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
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
    kname = sys.argv[3] if len(sys.argv) > 3 else "dtw_filter_kernel"
    units = float(sys.argv[4]) if len(sys.argv) > 4 else 4096.0 * 4096.0 * 128.0 * 128.0 / 64.0
    uname = sys.argv[5] if len(sys.argv) > 5 else "wave-cell"
    cmdf = os.path.join(src, "command.txt")
    cmd = open(cmdf).read().strip() if os.path.exists(cmdf) else "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
    stats = kernel_stats(src)
    out = {"source": src, "command": cmd, "dominant_kernel": kname, "kernels": [], "pmc": {}}
    lines = ["# rocprofv3 summary (%s)" % os.path.basename(dst), "",
             "Command: `%s` on 1 MI355X (tools/profile_bench.sh: one --kernel-trace --stats pass, separate --pmc passes)." % cmd, "",
             "## kernel trace (--kernel-trace --stats)", "",
             "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
    for r in sorted(stats, key=lambda r: -float(r["TotalDurationNs"])):
        k = {"name": r["Name"], "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
             "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])}
        out["kernels"].append(k)
        if len(out["kernels"]) <= 16:
            lines.append("| `%s` | %d | %.3f | %.4f | %.2f |" % (k["name"][:90], k["calls"], k["total_ms"], k["avg_ms"], k["pct"]))
    dom = [k for k in out["kernels"] if kname in k["name"]]
    if dom:
        out["dominant_kernel_avg_ms"] = sum(k["total_ms"] for k in dom) / max(1, sum(k["calls"] for k in dom))
    lines += ["", "## PMC (separate --pmc passes), per launch of `%s`" % kname, ""]
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds", "pmc_misc"):
        for kern, ctrs in pmc(src, sub).items():
            if kname not in kern:
                continue
            for c, vals in ctrs.items():
                out["pmc"][c] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
    p = out["pmc"]
    v = lambda c: p[c]["mean_per_launch"]
    for c, x in p.items():
        lines.append("- `%s` = %.6g (mean over %d launches)" % (c, x["mean_per_launch"], x["launches"]))
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        rd = v("FETCH_SIZE") * 1024 * 2      # KiB, x2 gfx950 correction
        wr = v("WRITE_SIZE") * 1024
        out["hbm_traffic_bytes_per_launch"] = rd + wr
        out["hbm_read_bytes_per_launch"] = rd
        out["hbm_write_bytes_per_launch"] = wr
        lines += ["", "HBM traffic per launch of `%s`: read %.1f MB (FETCH_SIZE KiB x 1024 x 2, "
                  "gfx950 correction) + write %.1f MB (WRITE_SIZE KiB x 1024) = **%.1f MB**." % (kname, rd / 1e6, wr / 1e6, (rd + wr) / 1e6)]
    if "SQ_ACTIVE_INST_VALU" in p and "SQ_WAVE_CYCLES" in p:
        valu, wave, insts = v("SQ_ACTIVE_INST_VALU"), v("SQ_WAVE_CYCLES"), v("SQ_INSTS_VALU")
        out["valu_active_over_wave_cycles"] = valu / wave
        out["cycles_per_valu_inst"] = 4.0 * valu / insts
        lines += ["", "Per wave: VALU active / wave cycles = %.3f; cycles per VALU instruction = %.2f (quad-cycle counters x 4); "
                  "VALU instructions per %s = %.2f." % (valu / wave, 4.0 * valu / insts, uname, insts / units)]
        parts = []
        for c, label in (("SQ_WAIT_ANY", "parked at s_waitcnt / barrier"), ("SQ_WAIT_INST_ANY", "issue-stalled"),
                         ("SQ_WAIT_INST_LDS", "of which LDS issue stalls"), ("SQ_ACTIVE_INST_LDS", "LDS instructions active")):
            if c in p:
                out[c.lower() + "_over_wave_cycles"] = v(c) / wave
                parts.append("%s %.3f (%s)" % (c, v(c) / wave, label))
        if parts:
            lines += ["", "Share of the waves' cycles: " + "; ".join(parts) + "."]
    if "SQ_LDS_BANK_CONFLICT" in p and "SQ_LDS_IDX_ACTIVE" in p:
        out["lds_bank_conflict_fraction"] = v("SQ_LDS_BANK_CONFLICT") / max(v("SQ_LDS_IDX_ACTIVE"), 1.0)
        lines += ["", "LDS: bank-conflict cycles / LDS-array cycles = %.4f; %.3g LDS instructions per launch." % (
            out["lds_bank_conflict_fraction"], v("SQ_INSTS_LDS") if "SQ_INSTS_LDS" in p else float("nan"))]
    if "GRBM_GUI_ACTIVE" in p and "SQ_ACTIVE_INST_VALU" in p:
        # GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles; 1024 SIMDs; SQ counters are in quad-cycles
        cyc = v("GRBM_GUI_ACTIVE") / 8.0
        busy = 4.0 * v("SQ_ACTIVE_INST_VALU") / (cyc * 1024.0)
        out["valu_busy_fraction"] = busy
        line = "Per-SIMD VALU busy fraction of `%s` = %.3f (4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))" % (kname, busy)
        if dom:
            ghz = cyc / (out["dominant_kernel_avg_ms"] * 1e6)
            out["clock_ghz_under_load"] = ghz
            out["valu_busy_cycles_per_unit"] = 4.0 * v("SQ_ACTIVE_INST_VALU") / units
            out["simd_cycles_per_unit"] = cyc * 1024.0 / units
            line += "; clock under load %.2f GHz; %.2f VALU-busy cycles and %.2f elapsed SIMD cycles per %s" % (
                ghz, out["valu_busy_cycles_per_unit"], out["simd_cycles_per_unit"], uname)
        lines += ["", line + "."]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in p:
            # MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the SIMDs
            mf = v("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024.0)
            out["mfma_busy_fraction"] = mf
            lines += ["", "MFMA utilisation of `%s` = %.3f of the matrix pipes' cycles (SQ_VALU_MFMA_BUSY_CYCLES / "
                      "(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)); %.4g MFMA instructions per launch." % (
                          kname, mf, v("SQ_INSTS_MFMA") if "SQ_INSTS_MFMA" in p else float("nan"))]
    if "TCC_HIT_sum" in p and "TCC_MISS_sum" in p:
        out["l2_hit_rate"] = v("TCC_HIT_sum") / max(v("TCC_HIT_sum") + v("TCC_MISS_sum"), 1.0)
        lines += ["", "L2 hit rate = %.3f." % out["l2_hit_rate"]]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    open(dst + ".md", "w").write("\n".join(lines) + "\n")
    json.dump(out, open(dst + ".json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
