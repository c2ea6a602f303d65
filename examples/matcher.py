#!/usr/bin/env python3
"""examples/matcher.py -- the flow of the reference's examples/matcher.rs on the GPU.

    python examples/matcher.py -d DICTIONARY_DIR -p PHONEME_DIR -o OUT.wav [--metric refcos|dtw]

Like the reference example (examples/matcher.rs:18-58): a dictionary from every *.wav of one
directory (SoundDictionary::from_path), then every *.wav of a second directory is replaced by its
nearest dictionary sound -- silence when the phoneme's max_power is below 0.03 (:41-45), otherwise
the match's samples zero-padded / truncated to the phoneme's length and scaled by
4^max_power into 16-bit PCM (:47-51) -- and the concatenation is written as a 16-bit WAV.
What differs: the reference calls match_sound once per phoneme; here all phonemes go through ONE
batched match (SoundDictionary.match_indices), and the MFCCs are this repository's own definition
(ssym_mfcc; the reference's arithmetic is in the un-vendored vox_box crate).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from soundsym_amd import Engine, Sound, SoundDictionary  # noqa: E402
from soundsym_amd.io import write_wav16  # noqa: E402


def main(argv=None) -> np.ndarray:
    ap = argparse.ArgumentParser()
    ap.add_argument("-d", "--dictionary", required=True)
    ap.add_argument("-p", "--phonemes", required=True)
    ap.add_argument("-o", "--output", required=True)
    ap.add_argument("--metric", choices=["refcos", "dtw"], default="refcos")
    args = ap.parse_args(argv)
    eng = Engine(metric=args.metric, dtype="f64")
    dictionary = SoundDictionary.from_path(args.dictionary, engine=eng)
    names = sorted(n for n in os.listdir(args.phonemes) if os.path.splitext(n)[1] == ".wav")
    phonemes = [Sound.from_path(os.path.join(args.phonemes, n), engine=eng) for n in names]
    loud = [p for p in phonemes if p.max_power() >= 0.03 and p.num_frames() > 0]
    idx, _ = dictionary.match_indices(loud, None) if loud else (np.zeros(0, dtype=np.uint32), None)
    match_of = {id(p): dictionary.sounds[int(i)] for p, i in zip(loud, idx)}
    out = []
    for p in phonemes:
        n = p.samples().size
        m = match_of.get(id(p))
        if m is None:                                       # examples/matcher.rs:41-45
            out.append(np.zeros(n, dtype=np.int16))
            continue
        smp = np.zeros(n)
        k = min(n, m.samples().size)
        smp[:k] = m.samples()[:k]                           # chain(repeat(0)).take(len), :48
        v = smp * 32767.0 * 4.0 ** p.max_power()            # :49
        out.append(np.trunc(np.clip(v, -32768.0, 32767.0)).astype(np.int16))      # `as i16` saturates
    pcm = np.concatenate(out) if out else np.zeros(0, dtype=np.int16)
    rate = phonemes[0].sample_rate() if phonemes else 44100.0
    write_wav16(args.output, pcm, rate)
    eng.close()
    return pcm


if __name__ == "__main__":
    main()
