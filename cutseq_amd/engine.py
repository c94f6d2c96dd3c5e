"""Batch engine: one :class:`TrimEngine` per GPU drives the two HIP kernels (scan, resolve).

Counterpart of ``runner.run(pipeline, Progress(), outfiles)`` (cutseq/run.py:473, 794):
where cutadapt loops over reads calling modifiers, this hands whole batches (SoA rows,
see ``include/cutseq_hip.h``) to ``cs_trim_batch`` / ``cs_trim_device`` and gets back one
8-byte ``cs_result`` per read.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import abi, capi
from .plan import TrimPlan


class TrimEngine:
    def __init__(self, plan: TrimPlan, device: int = 0, slots: int = 2, max_reads: int = 1 << 18,
                 max_stride: int = 152):
        self.L = capi.load()
        self.plan = plan
        self.device = device
        self._plan_h = C.c_void_p()
        self._eng_h = C.c_void_p()
        if any(op.tabulated and op.table is None for _m, _i, op in plan.demux_ops()):
            from . import demux  # tables of the demultiplexing ops: built on this device when still missing
            demux.ensure_tables(plan, device)
        a1, n1, a2, n2 = plan.pack()
        params = plan.params()
        capi.check(self.L.cs_plan_create(C.cast(a1, C.c_void_p), n1,
                                         C.cast(a2, C.c_void_p) if a2 is not None else None, n2,
                                         C.byref(params), C.byref(self._plan_h)))
        for mate, index, op in plan.demux_ops():
            if op.tabulated:
                capi.check(self.L.cs_plan_set_demux(self._plan_h, mate, index, op.table.ctypes.data, op.table.size))
            else:
                from .plan import pack_ops
                ops = pack_ops(op.barcode_ops(), limit=255)
                capi.check(self.L.cs_plan_set_demux_ops(self._plan_h, mate, index, C.cast(ops, C.c_void_p), len(op.barcodes)))
        self.n_slots, self.max_reads, self.max_stride = slots, max_reads, max_stride
        try:
            capi.check(self.L.cs_engine_create(self._plan_h, device, slots, max_reads, max_stride,
                                               C.byref(self._eng_h)))
        except Exception:
            self.L.cs_plan_destroy(self._plan_h)
            self._plan_h = C.c_void_p()
            raise

    # -- lifetime ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "_eng_h", None):
            self.L.cs_engine_destroy(self._eng_h)
            self._eng_h = C.c_void_p()
        if getattr(self, "_plan_h", None):
            self.L.cs_plan_destroy(self._plan_h)
            self._plan_h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- host-buffer path (H2D -> kernel -> D2H on the engine stream) ----------------------
    @staticmethod
    def _reads(seq, qual, lens, out, cap2=None) -> abi.cs_reads:
        r = abi.cs_reads()
        r.seq, r.qual, r.len, r.out = seq.ctypes.data, qual.ctypes.data, lens.ctypes.data, out.ctypes.data
        r.cap2 = cap2.ctypes.data if cap2 is not None else None
        return r

    @staticmethod
    def _check_mate(name, seq, qual, lens, n, stride):
        """The C ABI trusts its arrays (include/cutseq_hip.h, data contract): refuse anything else here."""
        for label, a in ((f"seq{name}", seq), (f"qual{name}", qual)):
            if not isinstance(a, np.ndarray) or a.dtype != np.uint8 or a.shape != (n, stride) or not a.flags.c_contiguous:
                raise ValueError(f"{label}: expected a C-contiguous uint8 array of shape ({n}, {stride})")
        if not isinstance(lens, np.ndarray) or lens.dtype != np.uint16 or lens.shape != (n,) or not lens.flags.c_contiguous:
            raise ValueError(f"len{name}: expected a C-contiguous uint16 array of shape ({n},)")
        if n and int(lens.max()) > stride:
            raise ValueError(f"len{name}: a read is longer than the row stride {stride}")

    def submit(self, slot: int, seq1, qual1, len1, seq2=None, qual2=None, len2=None, out=None, bc=None):
        """Asynchronous: returns the (still being filled) result arrays; call ``wait(slot)``.
        ``out``: optional (res1, cap2 | None, res2 | None) arrays to fill (e.g. pinned memory).
        ``bc``: optional uint8 array [n] for the barcode index (plans with a demultiplexing op; filled from the mate
        whose chain holds it)."""
        if seq1.ndim != 2:
            raise ValueError("seq1: expected a 2-D array [n_reads, stride]")
        n, stride = seq1.shape
        if stride % 4:
            raise ValueError(f"row stride {stride} is not a multiple of 4")
        self._check_mate(1, seq1, qual1, len1, n, stride)
        if (seq2 is not None) != self.plan.paired:
            raise ValueError("plan is %s-end" % ("paired" if self.plan.paired else "single"))
        if seq2 is not None:
            self._check_mate(2, seq2, qual2, len2, n, stride)
        if out is not None:
            out1, cap2, out2 = out
        else:
            out1 = np.empty(n, dtype=abi.RESULT_DTYPE)
            cap2 = np.empty(n, dtype=abi.CAP2_DTYPE) if self.plan.needs_cap2 else None
            out2 = np.empty(n, dtype=abi.RESULT_DTYPE) if seq2 is not None else None
        r1 = self._reads(seq1, qual1, len1, out1, cap2)
        if bc is not None and (bc.dtype != np.uint8 or bc.shape != (n,) or not bc.flags.c_contiguous):
            raise ValueError(f"bc: expected a C-contiguous uint8 array of shape ({n},)")
        bc_mate = self.plan.demux_mate or 1
        if bc is not None and bc_mate == 1:
            r1.bc = bc.ctypes.data
        r2p = None
        if seq2 is not None:
            r2 = self._reads(seq2, qual2, len2, out2)
            if bc is not None and bc_mate == 2:
                r2.bc = bc.ctypes.data
            r2p = C.byref(r2)
        capi.check(self.L.cs_trim_batch(self._eng_h, slot, C.byref(r1), r2p, n, stride))
        return out1, cap2, out2

    def wait(self, slot: int):
        capi.check(self.L.cs_sync(self._eng_h, slot))

    def trim(self, seq1, qual1, len1, seq2=None, qual2=None, len2=None, slot: int = 0):
        """Synchronous convenience wrapper -> (res1, cap2 | None, res2 | None)."""
        res = self.submit(slot, seq1, qual1, len1, seq2, qual2, len2)
        self.wait(slot)
        return res

    # -- device-pointer path (inputs resident in HBM) --------------------------------------
    def trim_device(self, r1: abi.cs_reads, r2: Optional[abi.cs_reads], n_reads: int, stride: int,
                    stream: Optional[int] = None, pipelined: bool = False):
        """Two kernel launches on ``stream``.  ``pipelined``: the resolve kernel goes to the engine's resolve stream
        (the next call's scan kernel overlaps it); results are complete only after ``join(stream)``."""
        fn = self.L.cs_trim_device_pipelined if pipelined else self.L.cs_trim_device
        capi.check(fn(self._eng_h, stream, C.byref(r1), C.byref(r2) if r2 is not None else None, n_reads, stride))

    def join(self, stream: Optional[int] = None):
        """Make ``stream`` wait for every resolve kernel issued so far (pipelined calls)."""
        capi.check(self.L.cs_join(self._eng_h, stream))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        capi.check(self.L.cs_last_kernel_ms(self._eng_h, C.byref(ms)))
        return float(ms.value)

    def last_kernel_split_ms(self) -> Tuple[float, float]:
        """(scan kernel, resolve kernel) of the last ``trim_device`` launch, ms."""
        ms = (C.c_float * 2)()
        capi.check(self.L.cs_last_kernel_split_ms(self._eng_h, C.byref(ms)))
        return float(ms[0]), float(ms[1])

    def kernel_time_totals(self, reset: bool = False) -> Tuple[int, float, float]:
        """(timed calls, scan-kernel ms, resolve-kernel ms) summed over the ``trim_device`` calls since the last reset."""
        calls, ms = C.c_uint32(), (C.c_float * 2)()
        capi.check(self.L.cs_kernel_time_totals(self._eng_h, C.byref(calls), C.byref(ms), 1 if reset else 0))
        return int(calls.value), float(ms[0]), float(ms[1])

    def stats(self, reset: bool = False) -> Tuple[abi.cs_stats, abi.cs_stats]:
        st = (abi.cs_stats * 2)()
        capi.check(self.L.cs_stats_fetch(self._eng_h, C.byref(st), 1 if reset else 0))
        return st[0], st[1]
