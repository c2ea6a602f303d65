"""Regenerates the .npz fixtures in this directory from the CPU oracle (oracle/).

The reference cannot be built or run here (Rust, no toolchain), so these
vectors pin the PRODUCT against the oracle restatement, and the oracle itself against the
hand-derived answers in refcos_kat.json.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from oracle.oracle import pack_segments  # noqa: E402
from soundsym_amd import synth  # noqa: E402


def main():
    o = oracle.load()
    # refcos: ragged 12-dim segments (the reference's NCOEFFS), f64
    src, tgt = synth.make_ragged(24, 16, 3, 21, 12, 0x5EED0100)
    src = [s.astype(np.float64) * 0.01 for s in src]
    tgt = [t.astype(np.float64) * 0.01 for t in tgt]
    tgt[3] = src[7].copy()             # an exact copy of a dictionary entry
    src[11] = src[5].copy()            # duplicates inside the dictionary (first wins)
    sf, so = pack_segments(src, 12)
    tf, to = pack_segments(tgt, 12)
    idx, val = o.refcos_match_all(sf, so, tf, to, 12)
    dist = np.linspace(0.0, 0.02, len(tgt))
    idx_d, val_d = o.refcos_match_all(sf, so, tf, to, 12, dist)
    sims = np.array([[o.cosine_sim(s, t) for t in tgt] for s in src])
    np.savez(os.path.join(HERE, "refcos_ragged.npz"), src=sf, src_off=so, tgt=tf, tgt_off=to,
             idx=idx, val=val, dist=dist, idx_d=idx_d, val_d=val_d, sims=sims)

    # dtw: planted grid 32 x 32 x 16f x 13d, f32 data
    g = synth.make_grid(32, 32, 16, 13, 0x5EED0101)
    sf, so = g.flat("sources", np.float64)
    tf, to = g.flat("targets", np.float64)
    idx, cost, mat = o.dtw_match_all(sf, so, tf, to, 13, want_matrix=True)
    idx_b, cost_b = o.dtw_match_all(sf, so, tf, to, 13, band=3)
    idx_s, cost_s = o.dtw_match_all(sf, so, tf, to, 13, squared=True)
    np.savez(os.path.join(HERE, "dtw_grid_32x32x16x13.npz"), sources=g.sources, targets=g.targets,
             planted=g.planted, idx=idx, cost=cost, matrix=mat, idx_band3=idx_b, cost_band3=cost_b,
             idx_sq=idx_s, cost_sq=cost_s)

    # dtw: ragged 13-dim
    src, tgt = synth.make_ragged(20, 12, 1, 40, 13, 0x5EED0102)
    tgt[2] = src[9][:-2].copy()
    sf, so = pack_segments(src, 13)
    tf, to = pack_segments(tgt, 13)
    idx, cost, mat = o.dtw_match_all(sf, so, tf, to, 13, want_matrix=True)
    np.savez(os.path.join(HERE, "dtw_ragged.npz"), src=sf.astype(np.float32), src_off=so,
             tgt=tf.astype(np.float32), tgt_off=to, idx=idx, cost=cost, matrix=mat)
    # rows F1 / F3 / F4: top-k, MFCC front-end, greedy chain -- one small case each
    g = np.load(os.path.join(HERE, "refcos_ragged.npz"))
    sims = o.refcos_matrix(g["src"], g["src_off"], g["tgt"], g["tgt_off"], 12)
    top_idx, top_key = o.topk(sims, 4, distance=g["dist"])
    u = synth.splitmix64(0x5EED0104, 6000).astype(np.float64) / 2.0 ** 64      # uniform [0, 1)
    t = np.arange(6000) / 44100.0
    wave = 0.4 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 3100.0 * t) + 0.05 * (u - 0.5)
    mfcc = o.mfcc(wave, 44100.0)
    start = g["tgt"][int(g["tgt_off"][1]) * 12:int(g["tgt_off"][2]) * 12]
    chain_dist = np.linspace(0.9, 1.1, 9)
    chain_idx, chain_val = o.chain(g["src"], g["src_off"], 12, start, chain_dist)
    np.savez(os.path.join(HERE, "rows_f.npz"), top_idx=top_idx, top_key=top_key, wave=wave, mfcc=mfcc,
             chain_start=start, chain_dist=chain_dist, chain_idx=chain_idx, chain_val=chain_val)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
