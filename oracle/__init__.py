"""CPU oracle package (TEST INFRASTRUCTURE ONLY, parity unpinned -- see cutseq_oracle.c).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``cutseq_amd``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from cutseq_amd import abi

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> Path:
    so = _HERE / "libcutseq_oracle.so"
    src = _HERE / "cutseq_oracle.c"
    hdr = _HERE.parent / "include" / "cutseq_hip.h"
    stale = (not so.exists()) or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime)
    if force or stale:
        subprocess.run(["make", "-C", str(_HERE), "-B", "libcutseq_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(str(so))
        L.cs_oracle_locate2.restype = C.c_int
        L.cs_oracle_locate2.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int * 6)]
        L.cs_oracle_quality_trim_index.restype = C.c_int
        L.cs_oracle_quality_trim_index.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.cs_oracle_trim_mt.restype = C.c_int
        L.cs_oracle_trim_mt.argtypes = [C.c_void_p, C.c_int, C.POINTER(abi.cs_params), C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.POINTER(abi.cs_stats), C.c_int]
        _LIB = L
    return _LIB


def locate(ref: str, query: str, max_error_rate: float, flags: int, min_overlap: int = 1,
           select_rule: int = abi.CS_SELECT_LEFTMOST, indel_tie: int = abi.CS_TIE_INSERTION):
    """Aligner.locate through the C restatement -> tuple or None."""
    m = len(ref)
    thr = (C.c_uint8 * (m + 1))(*[int(L * max_error_rate) for L in range(m + 1)])
    out = (C.c_int * 6)()
    hit = lib().cs_oracle_locate2(ref.encode(), m, query.encode(), len(query), thr, int(max_error_rate * m),
                                  flags, min(min_overlap, m), select_rule, indel_tie, C.byref(out))
    return tuple(out) if hit else None


def quality_trim_index(qualities: str, cutoff_back: int, base: int = 33) -> int:
    return lib().cs_oracle_quality_trim_index(qualities.encode(), len(qualities), cutoff_back, base)


def trim_mate(ops, n_ops: int, params: abi.cs_params, seq: np.ndarray, qual: np.ndarray, lens: np.ndarray,
              want_cap2: bool = False, threads: int = 1, out: np.ndarray | None = None):
    """Run one mate's batch through the C oracle.

    seq/qual: uint8 [n, stride] C-contiguous, lens: uint16 [n].  ``out``: a RESULT_DTYPE array [n] to fill (callers
    that run thousands of plans over one batch reuse it: fresh pages cost more than the alignments of short reads).
    Returns (results[RESULT_DTYPE], cap2 | None, cs_stats).
    """
    assert seq.dtype == np.uint8 and qual.dtype == np.uint8 and lens.dtype == np.uint16
    assert seq.flags.c_contiguous and qual.flags.c_contiguous and seq.shape == qual.shape
    n, stride = seq.shape
    if out is None:
        out = np.zeros(n, dtype=abi.RESULT_DTYPE)
    assert out.dtype == abi.RESULT_DTYPE and out.shape == (n,) and out.flags.c_contiguous
    cap2 = np.zeros(n, dtype=abi.CAP2_DTYPE) if want_cap2 else None
    st = abi.cs_stats()
    rc = lib().cs_oracle_trim_mt(C.cast(ops, C.c_void_p), n_ops, C.byref(params), seq.ctypes.data,
                                 qual.ctypes.data, lens.ctypes.data, n, stride, out.ctypes.data,
                                 cap2.ctypes.data if cap2 is not None else None, C.byref(st), threads)
    if rc != 0:
        raise RuntimeError(f"cs_oracle_trim_mt failed: {rc}")
    return out, cap2, st


def host_threads() -> int:
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)
