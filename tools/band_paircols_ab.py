#!/usr/bin/env python3
"""dtw_band_kernel, two columns per source read (SSYM_BAND_PAIRCOLS): same filter values bit for bit, and the time.

    python tools/band_paircols_ab.py build          (here, no GPU: the library with the variant compiled in)
    python tools/band_paircols_ab.py [n] [reps]     (on the GPU box)

The variant measured nothing (LAB.md R4.3) and is not in the product library: `build` makes
soundsym_amd/csrc/build_pc/libsoundsym_amd_pc.so with -DSSYM_BAND_PAIRCOLS_BUILD, which the run loads through SSYM_LIB.

First the filter's whole pair matrix with the knob off and on over uniform and ragged lengths, both radii the variant
is built for (24: four tiles of diagonals, 32: five), both record layouts (13 values: K = 32, 40 values: K = 48) and both
distances; then configs[4]'s shape (n x n segments of 256 frames, r = 32), alternating the two variants.
"""
import os, subprocess, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "soundsym_amd", "csrc")
LIB = os.path.join(CSRC, "build_pc", "libsoundsym_amd_pc.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.join(CSRC, "build_pc"), exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
    for o in os.listdir(os.path.join(CSRC, "build")):           # only dtw_filter.hip sees the define
        if o.endswith(".o") and o != "dtw_filter.o":
            subprocess.check_call(["cp", "-p", os.path.join(CSRC, "build", o), os.path.join(CSRC, "build_pc", o)])
    subprocess.check_call(["make", "-C", CSRC, "EXTRA=-DSSYM_BAND_PAIRCOLS_BUILD", "BUILD=build_pc", "OUT=" + LIB])
    print("built", LIB)
    sys.exit(0)
if not os.path.exists(LIB):
    sys.exit("run `python tools/band_paircols_ab.py build` first")
os.environ["SSYM_LIB"] = LIB
import numpy as np
sys.path.insert(0, ROOT)
from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def knob(on):
    os.environ["SSYM_BAND_PAIRCOLS"] = "1" if on else "0"


cases = []
for band in (24, 32):
    for dim in (13, 40):
        for squared in (False, True):
            cases.append(("grid", band, dim, squared))
            cases.append(("ragged", band, dim, squared))
bad = 0
for kind, band, dim, squared in cases:
    seed = 0x5EED0B00 + band * 4 + dim
    if kind == "grid":
        f = 97 if band == 24 else 130
        g = synth.make_grid(70, 300, f, dim, seed)
        src, tgt = list(g.sources), list(g.targets)
    else:
        src, tgt = synth.make_ragged(70, 300, 1, 140, dim, seed)
    sf, so = pack_segments(src, dim, np.float32)
    tf, to = pack_segments(tgt, dim, np.float32)
    mats = []
    for on in (False, True):
        knob(on)
        e = Engine(metric="dtw", dtype="f32", band=band, squared=squared)
        d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
        mats.append(e.pair_matrix(d, q, exact=False))
        e.close()
    same = np.array_equal(mats[0], mats[1])       # (+inf where the end cell lies outside the band, on both sides)
    fin = np.isfinite(mats[0])
    print(f"{kind:6s} r={band} dim={dim} squared={int(squared)}: finite {int(fin.sum())}/{fin.size}, "
          f"identical {same}", flush=True)
    if not same:
        diff = mats[0] != mats[1]
        ij = np.argwhere(diff)[:5]
        print("   first differences (source, target, off, on):",
              [(int(i), int(j), float(mats[0][i, j]), float(mats[1][i, j])) for i, j in ij], int(diff.sum()))
        bad += 1
if bad:
    sys.exit(1)

for dim in (40, 13):
    g = synth.make_grid(n, n, 256, dim, 0x5EED0B77)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    res = {False: [], True: []}
    for rep in range(reps):
        for on in (False, True):
            knob(on)
            e = Engine(metric="dtw", dtype="f32", band=32)
            d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
            e.match(d, q)
            best = 1e9
            for _ in range(4):
                e.match(d, q)
                best = min(best, e.timings()["main_ms"])
            res[on].append(best)
            e.close()
        print(f"{dim} values, rep {rep}: off {res[False][-1]:.3f} ms, on {res[True][-1]:.3f} ms", flush=True)
    off, on = min(res[False]), min(res[True])
    cells = 15584.0 * n * n
    print(f"configs[4] shape, {dim} values, {n} x {n}: off {off:.3f} ms ({cells / (off * 1e-3) / 9.83e12:.3f} of the cell "
          f"model), on {on:.3f} ms ({cells / (on * 1e-3) / 9.83e12:.3f}), on/off {on / off:.4f}", flush=True)
