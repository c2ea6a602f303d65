#!/usr/bin/env python3
"""Where dtw_filter_sp_kernel's waves spend their time (library built with -DSSYM_SP_PROF, loaded through SSYM_LIB).
usage: SSYM_LIB=build_ab/libssym_prof.so python tools/sp_prof.py n "slo-shi:tlo-thi" ..."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth, _native
from soundsym_amd.engine import pack_segments

n = int(sys.argv[1])
lib = _native.lib()
lib.ssym_debug_sp_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 12)()
e = Engine(metric="dtw", dtype="f32")
for spec in sys.argv[2:]:
    s, t = spec.split(":")
    slo, shi = map(int, s.split("-"))
    tlo, thi = map(int, t.split("-"))
    st = synth.Stream(0x5EED0B00 + shi * 1000 + thi)
    sig = synth.sigma(13)
    ls = slo + st.integers(n, shi - slo + 1)
    lt = tlo + st.integers(n, thi - tlo + 1)
    src = [(st.normal(int(f) * 13).reshape(int(f), 13) * sig).astype(np.float32) for f in ls]
    tgt = [(st.normal(int(f) * 13).reshape(int(f), 13) * sig).astype(np.float32) for f in lt]
    sf, so = pack_segments(src, 13, np.float32)
    tf, to = pack_segments(tgt, 13, np.float32)
    d, q = e.dictionary(sf, so, 13), e.queries(tf, to, 13)
    for _ in range(6):
        e.match(d, q)
    assert lib.ssym_debug_sp_prof(buf) == 0
    e.match(d, q)
    ms = e.timings()["main_ms"]
    assert lib.ssym_debug_sp_prof(buf) == 0
    waves, life, cols, starts, grabs, tasks, ncols, first, head, between, tail = [float(x) for x in buf][:11]
    print(f"{spec}: filter {ms:.3f} ms; {waves:.0f} waves (all launches of the call), {tasks:.0f} tasks, {ncols / max(tasks, 1):.1f} columns per task; "
          f"of a wave's life: column loops {cols / life:.3f}, task starts {starts / life:.3f}, grabs {grabs / life:.3f}, "
          f"first task's operands {first / life:.3f}, start to first task {head / life:.3f}, between tasks {between / life:.3f}, "
          f"after the last task {tail / life:.3f}; "
          f"ticks per column {cols / max(ncols, 1):.1f}, per task start {starts / max(tasks, 1):.1f}, mean life {life / waves:.0f} ticks", flush=True)
