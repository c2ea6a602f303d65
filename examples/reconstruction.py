#!/usr/bin/env python3
"""examples/reconstruction.py -- the flow of the reference's examples/reconstruction.rs on the GPU.

    python examples/reconstruction.py -s SOURCE.wav -t TARGET.wav -o OUT.wav [--labels LABELS.txt]
                                      [--metric refcos|dtw] [--segment-frames 16]

Like the reference example (examples/reconstruction.rs:26-86) it cuts the source sound into a
dictionary of segments, cuts the target into segments, replaces every target segment by its
nearest dictionary segment (SoundSequence::clone_from_dictionary) and writes the concatenation
as a 32-bit WAV.  What differs, and why: segmentation is by fixed-length chunks (or by an Audacity
label file for the target) instead of the GMM / voting-experts partitioner (outside the hot path,
SURVEY.md section 2), and the MFCCs are this repository's own definition (ssym_mfcc; the
reference's arithmetic is in the un-vendored vox_box crate).  Feature analysis, matching, length
fit, concatenation and the 32-bit conversion all run on the GPU through the C ABI.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from soundsym_amd import Engine, Sound, SoundDictionary, SoundSequence  # noqa: E402
from soundsym_amd.api import HOP  # noqa: E402
from soundsym_amd.api import frame_features  # noqa: E402
from soundsym_amd.io import audacity_labels_to_timestamps, read_wav, write_wav32  # noqa: E402


def chunk_lengths(n_samples: int, seg: int):
    """Segment lengths in samples, multiples of HOP like the partitioner's (src/lib.rs:137)."""
    full = [seg] * (n_samples // seg)
    rest = (n_samples - sum(full)) // HOP * HOP
    return full + ([rest] if rest else [])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-s", required=True, help="source file")
    ap.add_argument("-t", required=True, help="target file")
    ap.add_argument("-o", required=True, help="output path")
    ap.add_argument("--labels", help="Audacity label file segmenting the TARGET (tests/vowel.txt style)")
    ap.add_argument("--metric", default="refcos", choices=["refcos", "dtw"])
    ap.add_argument("--segment-frames", type=int, default=16)
    args = ap.parse_args(argv)

    engine = Engine(metric=args.metric, dtype="f64")
    seg = args.segment_frames * HOP
    src_samples, rate = read_wav(args.s)
    source = Sound(src_samples, rate, frame_features(src_samples, rate))
    dictionary = SoundDictionary.from_segments(source, chunk_lengths(src_samples.size, seg), engine=engine)
    dictionary.sounds = [s for s in dictionary.sounds if s.num_frames() > 0]

    tgt_samples, trate = read_wav(args.t)
    if args.labels:
        # SoundSequence::from_timestamps (src/sound.rs:419-428): samples [round(start*sr), round(end*sr)]
        targets = []
        for start, end, label in audacity_labels_to_timestamps(args.labels):
            a, b = int(round(start * trate)), int(round(end * trate))
            piece = tgt_samples[a:b + 1]
            if piece.size >= HOP:
                targets.append(Sound(piece, trate, frame_features(piece, trate), label))
    else:
        target = Sound(tgt_samples, trate, frame_features(tgt_samples, trate))
        td = SoundDictionary.from_segments(target, chunk_lengths(tgt_samples.size, seg), engine=engine)
        targets = [s for s in td.sounds if s.num_frames() > 0]
    sequence = SoundSequence.new(targets)

    samples, pcm = sequence.reconstruct_from_dictionary(dictionary, want_pcm32=True)
    write_wav32(args.o, sample_rate=trate, pcm=pcm)
    print(f"{len(dictionary.sounds)} dictionary segments, {len(targets)} target segments, "
          f"{samples.size} samples -> {args.o}")
    return samples


if __name__ == "__main__":
    main()
