"""tools/prune_speechlike.py -- how much early abandoning saves on data shaped like the reference's use:
two synthetic "utterances" strung together from one inventory of 48 sustained spectra ("phonemes":
a harmonic source through three random formant peaks) with random durations, pitch and gain, analysed
by the GPU MFCC front-end, cut at the unit boundaries.  The nearest neighbour of a target segment is then
another realisation of the same unit -- close, but not a copy."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments

SR, HOP = 44100.0, 256


def utterance(st, n_units, inventory, lo, hi):
    durs = (lo + st.integers(n_units, hi - lo + 1)) * HOP
    units = st.integers(n_units, len(inventory))
    out, bounds, pos = [], [0], 0
    for u, d in zip(units, durs):
        f0 = 90.0 + 120.0 * (st.integers(1, 1000)[0] / 1000.0)
        t = np.arange(int(d)) / SR
        sig = np.zeros(int(d))
        for h in range(1, 40):
            fr = f0 * h
            if fr > 8000:
                break
            amp = sum(a * np.exp(-0.5 * ((fr - c) / w) ** 2) for c, w, a in inventory[u]) + 0.01
            sig += amp * np.sin(2 * np.pi * fr * t + 0.37 * h)
        sig *= 0.2 + 0.3 * (st.integers(1, 1000)[0] / 1000.0)
        sig += 0.003 * st.normal(int(d))
        out.append(sig)
        pos += int(d)
        bounds.append(pos)
    return np.concatenate(out), np.asarray(bounds), units


def segments(e, wave, bounds):
    m = e.mfcc(wave, SR)                       # [frames][12]
    segs = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        segs.append(m[a // HOP: b // HOP])
    return segs


def main(n=4096, m=4096, lo=8, hi=40):
    st = synth.Stream(0x5EED7001)
    inv = [[(300 + 2500 * st.integers(1, 1000)[0] / 1000.0 * (k + 1) / 2, 80 + 200 * st.integers(1, 1000)[0] / 1000.0,
             0.3 + st.integers(1, 1000)[0] / 1000.0) for k in range(3)] for _ in range(48)]
    e = Engine(metric="dtw", dtype="f64", device=0)
    wa, ba, ua = utterance(st, n, inv, lo, hi)
    wb, bb, ub = utterance(st, m, inv, lo, hi)
    src, tgt = segments(e, wa, ba), segments(e, wb, bb)
    sf, so = pack_segments(src, 12, np.float64)
    tf, to = pack_segments(tgt, 12, np.float64)
    d, q = e.dictionary(sf, so, 12), e.queries(tf, to, 12)
    res = {}
    for prune in (False, False, True, True):         # (the first call of a context also builds the filter records)
        t0 = time.perf_counter()
        idx, cost = e.match(d, q, prune=prune)
        res[prune] = (idx, cost, (time.perf_counter() - t0) * 1e3, e.timings())
    same = np.array_equal(res[False][0], res[True][0]) and np.array_equal(res[False][1], res[True][1])
    tmf, tmp = res[False][3], res[True][3]
    hit = float(np.mean(ua[res[True][0]] == ub))
    print(f"{n}x{m} segments of {lo}..{hi} frames x 12 MFCCs: full {tmf['total_ms']:.2f} ms (filter {tmf['main_ms']:.2f})  "
          f"pruned {tmp['total_ms']:.2f} ms (thresholds {tmp['prune_ms']:.2f}, filter {tmp['main_ms']:.2f}, select {tmp['select_ms']:.2f}, "
          f"refine {tmp['refine_ms']:.2f})  identical={same}  "
          f"same-unit matches {hit:.2f}", flush=True)
    assert same
    e.close()


if __name__ == "__main__":
    main(4096, 4096, 8, 40)
    main(2048, 2048, 40, 120)
