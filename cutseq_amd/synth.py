"""Seeded synthetic read generator: TAKARAV3-shaped libraries.

Thin ctypes wrapper over ``csh_synth_pairs`` in ``csrc/cutseq_host.c`` (plain C, host
only).  Molecule (top strand):  P5 | inline5 umi5 mask5 | insert | mask3 umi3 inline3 | P7
    R1 = head + insert + tail + p7.fw ...   (read-through when the insert is short)
    R2 = rc(tail) + rc(insert) + rc(head) + p5.rc ...
Mixture (mirrors the reference fixture test/input_R{1,2}.fq.gz): ~35 % of inserts shorter than
the read so the 3' adapter shows up (positions skewed towards the read end), ~9 % end with only
a 3..19 nt adapter prefix, 0.1 % carry a 5' adapter artefact, 2 % carry a poly-T/A stretch,
0.7 % N, 1 % substitutions, single-base indels inside 3 % of the adapters, qualities from the
four bins the fixture uses (# - 9 I) with a degrading tail on 20 % of the reads.

Each pair has its own counter-based random stream keyed by (seed, global pair index):
chunking, threading and rank splits never change the bytes.
"""
from __future__ import annotations

import ctypes as C
import threading
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Iterator, Tuple

import numpy as np

from .common import BarcodeConfig

HOST_LIB_PATH = Path(__file__).with_name("libcutseq_host.so")

DEFAULT_SEED = 0xC0FFEE


class _SynthParams(C.Structure):
    _fields_ = [
        ("read_len", C.c_uint32), ("stride", C.c_uint32), ("seed", C.c_uint64), ("first_index", C.c_uint64),
        ("p5_fw", C.c_char_p), ("p7_fw", C.c_char_p), ("p5_rc", C.c_char_p), ("p7_rc", C.c_char_p),
        ("inline5", C.c_char_p), ("inline3", C.c_char_p),
        ("umi5", C.c_int32), ("umi3", C.c_int32), ("mask5", C.c_int32), ("mask3", C.c_int32),
        ("strand", C.c_int32), ("single_end", C.c_int32),
        ("adapter_fraction", C.c_double), ("partial_fraction", C.c_double), ("poly_fraction", C.c_double),
        ("art5_fraction", C.c_double), ("sub_rate", C.c_double), ("indel_frac", C.c_double), ("n_rate", C.c_double),
    ]


_host = None


_host_lock = threading.Lock()


def host_lib() -> C.CDLL:
    """One library object per process (other modules bind their own prototypes on it), created under a lock:
    threads that start together must not end up with two objects, one of them only half configured."""
    global _host
    if _host is None:
        with _host_lock:
            if _host is None:
                if not HOST_LIB_PATH.exists():
                    from . import build
                    build.build_host()
                L = C.CDLL(str(HOST_LIB_PATH))
                L.csh_synth_pairs.restype = C.c_int
                L.csh_synth_pairs.argtypes = [C.POINTER(_SynthParams), C.c_uint32] + [C.c_void_p] * 6 + [C.c_int]
                _host = L
    return _host


@dataclass
class SynthBatch:
    """SoA batch as the C ABI wants it (uint8 [n, stride] x2, uint16 [n])."""

    seq1: np.ndarray
    qual1: np.ndarray
    len1: np.ndarray
    seq2: np.ndarray | None
    qual2: np.ndarray | None
    len2: np.ndarray | None

    @property
    def n(self) -> int:
        return self.seq1.shape[0]

    @property
    def stride(self) -> int:
        return self.seq1.shape[1]


def usable_cpus(cap: int = 32) -> int:
    """affinity mask capped by the cgroup CPU quota (the GPU boxes grant a share of the host)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(cap, n))


def _stride_for(read_len: int) -> int:
    return (read_len + 3) // 4 * 4


def _params(bc: BarcodeConfig, read_len: int, stride: int, seed: int, first_index: int, single_end: bool,
            adapter_fraction: float, partial_fraction: float, poly_fraction: float, art5_fraction: float,
            sub_rate: float, indel_frac: float, n_rate: float) -> _SynthParams:
    p = _SynthParams()
    p.read_len, p.stride, p.seed = read_len, stride, seed & 0xFFFFFFFFFFFFFFFF
    p.first_index = first_index
    p.p5_fw, p.p7_fw = bc.p5.fw.upper().encode(), bc.p7.fw.upper().encode()
    p.p5_rc, p.p7_rc = bc.p5.rc.upper().encode(), bc.p7.rc.upper().encode()
    p.inline5, p.inline3 = bc.inline5.fw.upper().encode(), bc.inline3.fw.upper().encode()
    p.umi5, p.umi3, p.mask5, p.mask3 = bc.umi5.len, bc.umi3.len, bc.mask5.len, bc.mask3.len
    p.strand = {"+": 1, "-": -1, None: 0}[bc.strand]
    p.single_end = 1 if single_end else 0
    p.adapter_fraction, p.partial_fraction, p.poly_fraction = adapter_fraction, partial_fraction, poly_fraction
    p.art5_fraction, p.sub_rate, p.indel_frac, p.n_rate = art5_fraction, sub_rate, indel_frac, n_rate
    return p


def _scheme_config(scheme) -> BarcodeConfig:
    if scheme is None:
        from .common import BUILDIN_ADAPTERS
        scheme = BUILDIN_ADAPTERS["TAKARAV3"]
    return scheme if isinstance(scheme, BarcodeConfig) else BarcodeConfig(scheme)


SYNTH_LIB_PATH = Path(__file__).with_name("libcutseq_synth.so")
_synth_dev = None


def synth_lib() -> C.CDLL:
    """The generator's DEVICE form (csrc/synth_device.hip, include/cutseq_synth.h): same bytes as the host form,
    written straight into device arrays.  Bench / test infrastructure; the trimming library does not link it."""
    global _synth_dev
    if _synth_dev is None:
        with _host_lock:
            if _synth_dev is None:
                if not SYNTH_LIB_PATH.exists():
                    raise RuntimeError(f"{SYNTH_LIB_PATH.name} is not built: python -m cutseq_amd.build")
                L = C.CDLL(str(SYNTH_LIB_PATH))
                L.csd_synth_pairs.restype = C.c_int
                L.csd_synth_pairs.argtypes = [C.POINTER(_SynthParams), C.c_uint64] + [C.c_void_p] * 7
                L.csd_last_error.restype = C.c_char_p
                L.csd_abi_version.restype = C.c_int
                _synth_dev = L
    return _synth_dev


def generate_pairs_device(n: int, ptrs, read_len: int = 150, scheme: str | BarcodeConfig | None = None,
                          seed: int = DEFAULT_SEED, single_end: bool = False,
                          adapter_fraction: float = 0.35, partial_fraction: float = 0.09,
                          poly_fraction: float = 0.02, art5_fraction: float = 0.001,
                          sub_rate: float = 0.01, indel_frac: float = 0.03, n_rate: float = 0.007,
                          first_index: int = 0, stride: int | None = None, stream=None) -> int:
    """:func:`generate_pairs` into DEVICE arrays: ``ptrs`` = (seq1, qual1, len1, seq2, qual2, len2) device addresses
    (the mate-2 entries None for single-end batches) of ``[n][stride]`` byte rows / ``[n]`` uint16 lengths on the current
    device.  Enqueued on ``stream`` (a hipStream_t address; None = the default stream), not waited for.  -> stride"""
    bc = _scheme_config(scheme)
    stride = _stride_for(read_len) if stride is None else stride
    p = _params(bc, read_len, stride, seed, first_index, single_end, adapter_fraction, partial_fraction, poly_fraction,
                art5_fraction, sub_rate, indel_frac, n_rate)
    L = synth_lib()
    rc = L.csd_synth_pairs(C.byref(p), n, *[C.c_void_p(int(x)) if x else None for x in ptrs], stream)
    if rc != 0:
        raise ValueError(f"csd_synth_pairs: {L.csd_last_error().decode()}")
    return stride


def generate_pairs(n: int, read_len: int = 150, scheme: str | BarcodeConfig | None = None,
                   seed: int = DEFAULT_SEED, chunk_index: int = 0, single_end: bool = False,
                   adapter_fraction: float = 0.35, partial_fraction: float = 0.09,
                   poly_fraction: float = 0.02, art5_fraction: float = 0.001,
                   sub_rate: float = 0.01, indel_frac: float = 0.03, n_rate: float = 0.007,
                   first_index: int | None = None, threads: int | None = None, out: SynthBatch | None = None) -> SynthBatch:
    """Generate ``n`` pairs (or single reads) of ``read_len`` bases.

    Pair ``i`` of the call is global pair ``first_index + i`` (default ``chunk_index << 32``)."""
    bc = _scheme_config(scheme)
    stride = _stride_for(read_len)
    if out is None:
        def mk():
            return (np.empty((n, stride), dtype=np.uint8), np.empty((n, stride), dtype=np.uint8),
                    np.empty(n, dtype=np.uint16))
        a = mk()
        b = (None, None, None) if single_end else mk()
        out = SynthBatch(a[0], a[1], a[2], b[0], b[1], b[2])
    assert out.seq1.shape == (n, stride)
    p = _params(bc, read_len, stride, seed, (chunk_index << 32) if first_index is None else first_index, single_end,
                adapter_fraction, partial_fraction, poly_fraction, art5_fraction, sub_rate, indel_frac, n_rate)
    if threads is None:
        threads = usable_cpus()
    rc = host_lib().csh_synth_pairs(
        C.byref(p), n, out.seq1.ctypes.data, out.qual1.ctypes.data, out.len1.ctypes.data,
        None if single_end else out.seq2.ctypes.data, None if single_end else out.qual2.ctypes.data,
        None if single_end else out.len2.ctypes.data, threads)
    if rc != 0:
        raise ValueError("csh_synth_pairs rejected the parameters")
    return out


def generate_single_adapter(n: int, read_len: int = 150, adapter: str = "AGATCGGAAGAGC",
                            seed: int = DEFAULT_SEED, chunk_index: int = 0, **kw) -> SynthBatch:
    """BASELINE.json config 2: single-end reads, one 3' adapter."""
    scheme = f"ACACGACGCTCTTCCGATCT>{adapter}"
    kw.setdefault("poly_fraction", 0.0)
    kw.setdefault("art5_fraction", 0.0)
    return generate_pairs(n, read_len, scheme, seed, chunk_index, single_end=True, **kw)


def iter_chunks(total: int, chunk: int) -> Iterator[Tuple[int, int]]:
    for i, lo in enumerate(range(0, total, chunk)):
        yield i, min(chunk, total - lo)


def headers(n: int, mate: int, first: int = 0):
    """``SIM:<idx> <mate>:N:0:IDX`` (id + comment, like the fixture's Illumina headers)."""
    return [f"SIM:{first + i} {mate}:N:0:IDX" for i in range(n)]
