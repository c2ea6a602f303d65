"""CPU oracle for the soundsym matching path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Nothing under ``soundsym_amd/`` does.  See ``oracle/ssym_oracle.c`` for what is
restated, from which reference lines, and for the "parity unpinned" statement.
"""
from .oracle import (  # noqa: F401
    Oracle,
    build,
    load,
    np_cosine_sim,
    np_at_distance,
    np_dtw,
)
