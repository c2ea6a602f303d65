"""Host-side mirror of the reference's matching interface: Sound / SoundDictionary / SoundSequence.

Same names, argument meaning and error behaviour as the reference for the hot path
(upstream src/sound.rs), so callers and tests read like the reference's own:

    Sound.from_samples / from_path / write_file             src/sound.rs:92, 114, 129
    SoundDictionary.new / from_path / from_segments / add_segments   src/sound.rs:296, 304, 323, 330
    SoundDictionary.match_sound / at_distance               src/sound.rs:346, 351
    SoundSequence.new / from_timestamps / morph_to / clone_from_dictionary   src/sound.rs:392, 419, 440, 451
    SoundSequence.from_distances / to_sound                 src/sound.rs:405, 475

Every comparison runs on the GPU through the C ABI (`engine.Engine`); this module only keeps the
containers, does the length fit of src/sound.rs:456-465 on the matched samples, and translates
errors.  A Sound is built from samples plus ready-made features (the `Some(mfccs)` form of
Sound::from_samples, src/sound.rs:92-94) or analysed on the GPU (`None`: `ssym_mfcc`, a
self-consistent MFCC -- the reference's arithmetic is in un-vendored crates, parity unpinned).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from ._native import NO_MATCH, EmptyDictionaryError
from .engine import Engine, pack_segments

NCOEFFS = 12   # src/lib.rs:22
HOP = 256      # src/lib.rs:24
BIN = 1024     # src/lib.rs:25

_default_engine: Optional[Engine] = None


def default_engine() -> Engine:
    """refcos / f64 on GPU 0: the reference's own metric and dtype."""
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(metric="refcos", dtype="f64", device=0)
    return _default_engine


def frame_features(samples, sample_rate: float, ncoeffs: int = NCOEFFS, engine: Optional[Engine] = None,
                   pad_tail: bool = True) -> np.ndarray:
    """analyze_mfccs (src/sound.rs:215-242) for a whole sound on the GPU (`ssym_mfcc`; this package's own MFCC
    definition -- the reference's lives in un-vendored crates, PARITY UNPINNED): [n_frames * ncoeffs] f64,
    frame-major.  pad_tail=True gives len(samples) // 256 frames (the tail windows read zeros past the end), which is
    what the segment arithmetic of SoundDictionary::add_segments (`seg / HOP * NCOEFFS` values per segment,
    src/sound.rs:335) expects of a parent sound; pad_tail=False keeps full windows only."""
    e = engine or default_engine()
    return e.mfcc(samples, sample_rate, ncoeffs, pad_tail=pad_tail).reshape(-1)


def _round_half_away(x: float) -> int:
    """f64::round (src/sound.rs:422-423): half away from zero -- Python's round() is half to even."""
    import math
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


def _round_as_usize(x: float) -> int:
    """`x.round() as usize` (src/sound.rs:422-423): f64::round, then Rust's SATURATING float-to-integer cast -- a
    negative value or NaN becomes 0, +inf (and anything beyond) usize::MAX."""
    import math
    if x != x or x <= 0:
        return 0
    if math.isinf(x) or x >= 2.0 ** 64:
        return 2 ** 64 - 1
    return _round_half_away(x)


class _SoundList(list):
    """`pub sounds: Vec<Arc<Sound>>` as a list that counts its own mutations: the dictionary's GPU copies are rebuilt
    when `version` has moved, which costs O(1) per query instead of an identity scan of the whole list."""

    def __init__(self, *a):
        super().__init__(*a)
        self.version = 0


def _bumping(name):
    base = getattr(list, name)

    def method(self, *a, **k):
        self.version += 1
        return base(self, *a, **k)
    method.__name__ = name
    return method


for _m in ("append", "extend", "insert", "pop", "remove", "clear", "reverse", "sort", "__setitem__", "__delitem__",
           "__iadd__", "__imul__"):
    setattr(_SoundList, _m, _bumping(_m))


class Sound:
    """Samples + flat frame-major features of one sound (src/sound.rs:73-82)."""

    def __init__(self, samples, sample_rate: float, mfccs, name: Optional[str] = None,
                 ncoeffs: int = NCOEFFS):
        self.name = name
        self._samples = np.ascontiguousarray(samples, dtype=np.float64).reshape(-1)
        self._sample_rate = float(sample_rate)
        self.ncoeffs = int(ncoeffs)
        self._mfccs = None
        if mfccs is not None:
            m = np.ascontiguousarray(mfccs, dtype=np.float64).reshape(-1)
            if m.size % self.ncoeffs:
                raise ValueError("mfccs must hold whole frames of `ncoeffs` values")
            self._mfccs = m

    @staticmethod
    def from_samples(samples, sample_rate: float, mfccs=None, name: Optional[str] = None,
                     ncoeffs: int = NCOEFFS, engine: Optional[Engine] = None) -> "Sound":
        """Sound::from_samples (src/sound.rs:92-107).  mfccs=None runs the MFCC analysis as the
        reference does (analyze_mfccs, :215-242) -- here on the GPU (`ssym_mfcc`; its arithmetic is
        this package's own definition, parity unpinned)."""
        if mfccs is None:
            e = engine or default_engine()
            mfccs = e.mfcc(samples, sample_rate, ncoeffs).reshape(-1)
        return Sound(samples, sample_rate, mfccs, name, ncoeffs)

    @staticmethod
    def from_path(path, engine: Optional[Engine] = None) -> "Sound":
        """Sound::from_path (src/sound.rs:114-127): read a WAV (integer PCM scaled by
        i32::MAX >> (32 - bits), :116-119), name = file stem, features analysed (mfccs = None)."""
        import os
        from . import io as sio
        samples, rate = sio.read_wav(str(path))
        stem = os.path.splitext(os.path.basename(str(path)))[0]
        return Sound.from_samples(samples, float(rate), None, stem, engine=engine)

    def write_file(self, path) -> None:
        """Sound::write_file (src/sound.rs:129-143): mono 32-bit integer WAV,
        sample -> (i32::MAX as f64 * sample) as i32."""
        from . import io as sio
        sio.write_wav32(str(path), self._samples, int(self._sample_rate))

    def max_power(self) -> float:             # src/sound.rs:197, analyze_max_power :244-256 (host side)
        from . import io as sio
        return sio.max_power(self._samples)

    def mean_mfccs(self) -> np.ndarray:       # src/sound.rs:205, analyze_mean_mfccs :271-286
        m = self.mfccs().reshape(-1, self.ncoeffs)
        acc = np.zeros(self.ncoeffs)
        for row in m:                         # frames in order, like the reference's fold
            acc = acc + row
        return acc / m.shape[0] if m.shape[0] else acc * np.nan

    def samples(self) -> np.ndarray:          # src/sound.rs:181
        return self._samples

    def sample_rate(self) -> float:           # src/sound.rs:185
        return self._sample_rate

    def mfccs(self) -> np.ndarray:            # src/sound.rs:191
        if self._mfccs is None:
            raise ValueError("this Sound carries no features (build it with Sound.from_samples(.., None) to analyse)")
        return self._mfccs

    def has_mfccs(self) -> bool:
        return self._mfccs is not None

    def num_frames(self) -> int:              # src/sound.rs:210
        return self.mfccs().size // self.ncoeffs


class SoundDictionary:
    """Cache of Sounds searched by similarity (src/sound.rs:290-371)."""

    def __init__(self, engine: Optional[Engine] = None):
        self._sounds = _SoundList()           # `pub sounds: Vec<Arc<Sound>>` (the `sounds` property below)
        self._engine = engine
        self._resident = None
        self._resident_key = None
        self._samples_res = None
        self._samples_key = None

    # constructors -----------------------------------------------------------------------------
    @staticmethod
    def new(engine: Optional[Engine] = None) -> "SoundDictionary":     # src/sound.rs:296
        return SoundDictionary(engine)

    @staticmethod
    def from_path(path, engine: Optional[Engine] = None) -> "SoundDictionary":
        """SoundDictionary::from_path (src/sound.rs:304-321): every *.wav of a directory, in directory
        order (sorted here, so that indices do not depend on the file system), features analysed."""
        import os
        d = SoundDictionary(engine)
        for name in sorted(os.listdir(str(path))):
            if os.path.splitext(name)[1] != ".wav":
                continue
            d.sounds.append(Sound.from_path(os.path.join(str(path), name), engine=engine))
        return d

    @staticmethod
    def from_segments(sound: Sound, segments: Sequence[int],
                      engine: Optional[Engine] = None) -> "SoundDictionary":   # src/sound.rs:323
        d = SoundDictionary(engine)
        d.add_segments(sound, segments)
        return d

    def add_segments(self, sound: Sound, segments: Sequence[int]) -> None:
        """src/sound.rs:330-343: consecutive `seg` samples and `seg / HOP * ncoeffs` feature
        values per segment (integer division), taken in order from the parent sound; a segment
        running past the end gets what is left, like `take` on an exhausted iterator."""
        samples, mfccs = sound.samples(), sound.mfccs()
        spos = mpos = 0
        for seg in segments:
            seg = int(seg)
            samp = samples[spos:spos + seg]
            spos = min(spos + seg, samples.size)
            nm = seg // HOP * sound.ncoeffs
            mf = mfccs[mpos:mpos + nm]
            mpos = min(mpos + nm, mfccs.size)
            self.sounds.append(Sound(samp.copy(), sound.sample_rate(), mf.copy(), None, sound.ncoeffs))

    # device residency ---------------------------------------------------------------------------
    @property
    def engine(self) -> Engine:
        if self._engine is None:
            self._engine = default_engine()
        return self._engine

    def _dim(self) -> int:
        return self.sounds[0].ncoeffs

    @property
    def sounds(self) -> List[Sound]:
        return self._sounds

    @sounds.setter
    def sounds(self, value) -> None:
        self._sounds = _SoundList(value)      # a fresh list object: its identity is part of the key below

    def _content_key(self):
        """What `sounds` holds now, in O(1).  `sounds` is the public, mutable list (`pub sounds`): entries may have
        been replaced, reordered, or popped and pushed since the last pack, so the length alone does not say.  The
        list counts its own mutations (_SoundList.version) and assigning a new list makes a new object, so (the list
        object, its version) changes whenever the content can have; a Sound's arrays are fixed at construction.  The
        key holds the list, so its identity cannot be reused while the key is kept."""
        return (self._sounds, self._sounds.version)

    @staticmethod
    def _same(key, now) -> bool:
        return key is not None and key[0] is now[0] and key[1] == now[1]

    def invalidate(self) -> None:
        """Drop the GPU copies (they are rebuilt on the next query)."""
        if self._resident is not None:
            self._resident.close()
        self._resident = self._resident_key = None
        if self._samples_res is not None:
            self._samples_res.close()
        self._samples_res = self._samples_key = None

    def resident(self):
        """Pack the dictionary's features once per content change, not per query."""
        key = self._content_key()
        if self._resident is None or not self._same(self._resident_key, key):
            if self._resident is not None:
                self._resident.close()
            dim = self._dim() if self.sounds else NCOEFFS
            flat, off = pack_segments([s.mfccs() for s in self.sounds], dim, self.engine.np_dtype)
            self._resident = self.engine.dictionary(flat, off, dim)
            self._resident_key = key
        return self._resident

    def resident_samples(self):
        """The sounds' samples on the GPU, for the reconstruction tail (ssym_samples_create)."""
        key = self._content_key()
        if self._samples_res is None or not self._same(self._samples_key, key):
            if self._samples_res is not None:
                self._samples_res.close()
            smp = [s.samples() for s in self.sounds]
            off = np.concatenate([[0], np.cumsum([x.size for x in smp])]).astype(np.uint64)
            flat = np.concatenate(smp) if smp else np.zeros(0)
            self._samples_res = self.engine.samples(flat, off)
            self._samples_key = key
        return self._samples_res

    # queries ------------------------------------------------------------------------------------
    def match_sound(self, other: Sound) -> Sound:                       # src/sound.rs:346
        if not self.sounds:
            raise EmptyDictionaryError(-2, "empty dictionary")          # reference: panic, :369
        default = 1.0 if self.engine.metric == "refcos" else 0.0
        return self.at_distance(default, other)

    def at_distance(self, distance: float, other: Sound) -> Sound:      # src/sound.rs:351
        if not self.sounds:
            # the reference indexes an empty Vec and panics (src/sound.rs:369)
            raise EmptyDictionaryError(-2, "empty dictionary")
        idx, _ = self.engine.match_one(self.resident(), other.mfccs(), distance)
        return self.sounds[idx]              # Some(self.sounds[min_idx].clone())

    def match_indices(self, targets: Sequence[Sound], distances=None):
        """Batched form of the loops at src/sound.rs:442-446 and :453-454."""
        if not self.sounds:
            raise EmptyDictionaryError(-2, "empty dictionary")
        flat, off = pack_segments([t.mfccs() for t in targets], self._dim(), self.engine.np_dtype)
        return self.engine.match_batch(self.resident(), flat, off, distances)


    def candidates(self, targets: Sequence[Sound], k: int, distances=None) -> List[List[Sound]]:
        """The k best dictionary sounds per target, best first: what k
        successive at_distance calls (src/sound.rs:351) would return if each winner were removed."""
        if not self.sounds:
            raise EmptyDictionaryError(-2, "empty dictionary")
        flat, off = pack_segments([t.mfccs() for t in targets], self._dim(), self.engine.np_dtype)
        q = self.engine.queries(flat, off, self._dim())
        idx, _ = self.engine.match_topk(self.resident(), q, k, distances)
        q.close()
        return [[self.sounds[int(i)] for i in row if int(i) != NO_MATCH] for row in idx]


def length_fit(matched: np.ndarray, n_target: int) -> np.ndarray:
    """src/sound.rs:456-465: zero-pad the matched samples up to the target's sample count, or
    truncate them down to it."""
    out = np.zeros(n_target, dtype=np.float64)
    n = min(matched.size, n_target)
    out[:n] = matched[:n]
    return out


class SoundSequence:
    """Sequence of sounds (src/sound.rs:375-484); `distances` between neighbours is not kept."""

    def __init__(self, sounds: Sequence[Sound]):
        self._sounds = list(sounds)

    @staticmethod
    def new(sounds: Sequence[Sound]) -> "SoundSequence":                # src/sound.rs:392
        return SoundSequence(sounds)

    def sounds(self) -> List[Sound]:                                    # src/sound.rs:432
        return self._sounds

    @staticmethod
    def from_timestamps(sound: Sound, timestamps, engine: Optional[Engine] = None) -> "SoundSequence":
        """SoundSequence::from_timestamps (src/sound.rs:419-430): one Sound per (start s, end s, label),
        samples [round(start * rate), round(end * rate)] INCLUSIVE (:422-424), features analysed.  The cast is Rust's
        saturating one: a negative or NaN time reads as sample 0 and the call proceeds; only an end beyond the sound
        (or a start beyond the end) fails, where the reference's slice panics."""
        out = []
        smp, rate = sound.samples(), sound.sample_rate()
        for start, end, label in timestamps:
            a, b = _round_as_usize(start * rate), _round_as_usize(end * rate)        # `.round() as usize`, :422-423
            if b + 1 > smp.size or a > b + 1:
                # the reference slices `samples[start_sample..end_sample + 1]` (:424) and panics out of range
                raise IndexError(f"timestamp ({start}, {end}) -> samples [{a}, {b}] outside the sound's {smp.size} samples")
            out.append(Sound.from_samples(smp[a:b + 1].copy(), rate, None, label, sound.ncoeffs, engine=engine))
        return SoundSequence(out)

    @staticmethod
    def from_distances(distances: Sequence[float], start: Sound,
                       dict_: SoundDictionary) -> "SoundSequence":      # src/sound.rs:405-417
        """Greedy chain: each step's query is the previous result.  The steps run on the device
        back to back (ssym_chain): one call, one wait."""
        if not len(distances):
            return SoundSequence([start])
        if not dict_.sounds:
            raise EmptyDictionaryError(-2, "empty dictionary")          # reference: panic, :369
        idx, _ = dict_.engine.chain(dict_.resident(), start.mfccs(), distances)
        return SoundSequence([start] + [dict_.sounds[int(i)] for i in idx])

    def morph_to(self, distances: Sequence[float], dict_: SoundDictionary) -> "SoundSequence":
        """src/sound.rs:440-449: zip(sounds, distances) -> at_distance, here as ONE batch."""
        n = min(len(self._sounds), len(distances))
        if n == 0:
            return SoundSequence([])
        idx, _ = dict_.match_indices(self._sounds[:n], np.asarray(distances[:n], dtype=np.float64))
        return SoundSequence([dict_.sounds[int(i)] for i in idx])

    def clone_from_dictionary(self, dict_: SoundDictionary) -> "SoundSequence":
        """src/sound.rs:451-472: match every sound, then fit the match to the target's length."""
        if not self._sounds:
            return SoundSequence([])
        idx, _ = dict_.match_indices(self._sounds, None)
        out = []
        for sound, i in zip(self._sounds, idx):
            s = dict_.sounds[int(i)]
            diff = sound.samples().size - s.samples().size
            if diff == 0:
                out.append(s)                                            # :463-464 shares the Arc
            else:
                # :457-462 builds a new Sound from the fitted samples and re-analyses it; the
                # re-analysis (MFCC) is outside this package, the samples are exact
                out.append(Sound(length_fit(s.samples(), sound.samples().size), sound.sample_rate(),
                                 None, None, s.ncoeffs))
        return SoundSequence(out)

    def reconstruct_from_dictionary(self, dict_: "SoundDictionary", want_pcm32: bool = False):
        """clone_from_dictionary(dict).to_sound().samples() in one go (src/sound.rs:451-480): match
        on the GPU, then gather / length-fit / concatenate on the GPU (ssym_reconstruct), optionally
        with write_file's 32-bit conversion (:139)."""
        if not self._sounds:
            return (np.zeros(0), np.zeros(0, dtype=np.int32)) if want_pcm32 else np.zeros(0)
        idx, _ = dict_.match_indices(self._sounds, None)
        lens = np.array([s.samples().size for s in self._sounds], dtype=np.uint64)
        out_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        return dict_.engine.reconstruct(dict_.resident_samples(), idx, out_off, want_pcm32)

    def to_sound(self) -> Sound:                                        # src/sound.rs:475-483
        parts = [s.samples() for s in self._sounds]
        samples = np.concatenate(parts) if parts else np.zeros(0)
        rate = self._sounds[0].sample_rate() if self._sounds else 44100.0
        ncoeffs = self._sounds[0].ncoeffs if self._sounds else NCOEFFS
        return Sound(samples, rate, None, None, ncoeffs)
