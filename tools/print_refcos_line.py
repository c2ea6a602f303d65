#!/usr/bin/env python3
"""tools/print_refcos_line.py <bench line.json> -- the refcos figures of a bench.py line (both filters) on one screen."""
import json, sys
l = json.load(open(sys.argv[1]))
r = l["secondary"]["refcos"]
print("headline: %.3g pairs/s, %.2f ms per step, roofline.frac %.3f" % (l["value"], l["ms_per_step"], l["roofline"]["frac"]))
print("refcos (%s): %.3g pairs/s per call, %.3f ms per call; device %s; main kernel %.3f ms = %.3f of its peak; %d pairs keyed exactly" % (
    r["filter"], r["value"], r["ms_per_step"], r["phase_ms"], r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["pairs_rescored_exactly"]))
f = r["f64_filter"]
print("refcos (f64 filter): %.3g pairs/s per call, %.3f ms per call; main kernel %.3f ms = %.3f of the f64 matrix peak" % (
    f["value"], f["ms_per_step"], f["roofline"]["kernel_ms"], f["roofline"]["frac"]))
print("match_one %.1f us/query, chain %.2f us/step" % (l["secondary"]["match_one"]["value"], l["secondary"]["chain"]["value"]))
