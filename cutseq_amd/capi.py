"""ctypes binding of ``libcutseq_hip.so`` (the C ABI in ``include/cutseq_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` /
``python -m cutseq_amd.build``.  There is no fallback: a missing library, a missing GPU
or a non-gfx950 device raises :class:`HipUnavailable`.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from . import abi

LIB_PATH = Path(__file__).with_name("libcutseq_hip.so")

EXPORTS = (
    "cs_abi_version", "cs_last_error", "cs_device_count", "cs_plan_create", "cs_plan_destroy", "cs_plan_set_demux", "cs_plan_set_demux_ops",
    "cs_engine_create", "cs_engine_destroy", "cs_trim_device", "cs_trim_device_pipelined", "cs_join", "cs_trim_batch", "cs_sync",
    "cs_stats_fetch", "cs_last_kernel_ms", "cs_last_kernel_split_ms", "cs_kernel_time_totals", "cs_alloc_pinned", "cs_alloc_pinned_huge", "cs_free_pinned", "cs_alloc_device",
    "cs_free_device", "cs_copy_to_device", "cs_copy_to_host",
    "cs_text_create", "cs_text_destroy", "cs_text_submit", "cs_text_wait", "cs_text_routes", "cs_text_fetch",
)


class HipUnavailable(RuntimeError):
    """The HIP extension (or a usable MI355X) is missing; the product has no CPU path."""


class CsError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"cutseq_hip error {code}: {message}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    import os
    lib_path = Path(os.environ.get("CUTSEQ_HIP_LIB", str(LIB_PATH)))  # A/B builds when tuning
    if not lib_path.exists():
        raise HipUnavailable(
            f"{lib_path} not found: build it with `python -m cutseq_amd.build` "
            "(hipcc --offload-arch=gfx950). cutseq_amd has no CPU trimming path."
        )
    try:
        L = C.CDLL(str(lib_path))
    except OSError as exc:  # pragma: no cover
        raise HipUnavailable(f"cannot load {lib_path}: {exc}") from exc
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    L.cs_abi_version.restype = i32
    L.cs_last_error.restype = C.c_char_p
    L.cs_device_count.restype = i32
    L.cs_plan_create.restype = i32
    L.cs_plan_create.argtypes = [vp, i32, vp, i32, C.POINTER(abi.cs_params), C.POINTER(vp)]
    L.cs_plan_set_demux.restype = i32
    L.cs_plan_set_demux.argtypes = [vp, i32, i32, vp, C.c_size_t]
    L.cs_plan_set_demux_ops.restype = i32
    L.cs_plan_set_demux_ops.argtypes = [vp, i32, i32, vp, i32]
    L.cs_plan_destroy.restype = None
    L.cs_plan_destroy.argtypes = [vp]
    L.cs_engine_create.restype = i32
    L.cs_engine_create.argtypes = [vp, i32, u32, u32, u32, C.POINTER(vp)]
    L.cs_engine_destroy.restype = None
    L.cs_engine_destroy.argtypes = [vp]
    L.cs_trim_device.restype = i32
    L.cs_trim_device.argtypes = [vp, vp, C.POINTER(abi.cs_reads), C.POINTER(abi.cs_reads), u32, u32]
    L.cs_trim_device_pipelined.restype = i32
    L.cs_trim_device_pipelined.argtypes = [vp, vp, C.POINTER(abi.cs_reads), C.POINTER(abi.cs_reads), u32, u32]
    L.cs_join.restype = i32
    L.cs_join.argtypes = [vp, vp]
    L.cs_trim_batch.restype = i32
    L.cs_trim_batch.argtypes = [vp, u32, C.POINTER(abi.cs_reads), C.POINTER(abi.cs_reads), u32, u32]
    L.cs_sync.restype = i32
    L.cs_sync.argtypes = [vp, u32]
    L.cs_stats_fetch.restype = i32
    L.cs_stats_fetch.argtypes = [vp, C.POINTER(abi.cs_stats * 2), i32]
    L.cs_last_kernel_ms.restype = i32
    L.cs_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.cs_last_kernel_split_ms.restype = i32
    L.cs_last_kernel_split_ms.argtypes = [vp, C.POINTER(C.c_float * 2)]
    L.cs_kernel_time_totals.restype = i32
    L.cs_kernel_time_totals.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_float * 2), i32]
    L.cs_alloc_pinned.restype = vp
    L.cs_alloc_pinned.argtypes = [C.c_size_t]
    L.cs_alloc_pinned_huge.restype = vp
    L.cs_alloc_pinned_huge.argtypes = [C.c_size_t]
    L.cs_free_pinned.restype = None
    L.cs_free_pinned.argtypes = [vp]
    L.cs_alloc_device.restype = vp
    L.cs_alloc_device.argtypes = [i32, C.c_size_t]
    L.cs_free_device.restype = None
    L.cs_free_device.argtypes = [i32, vp]
    L.cs_copy_to_device.restype = i32
    L.cs_copy_to_device.argtypes = [i32, vp, vp, C.c_size_t]
    L.cs_copy_to_host.restype = i32
    L.cs_copy_to_host.argtypes = [i32, vp, vp, C.c_size_t]
    u64 = C.c_uint64
    L.cs_text_create.restype = i32
    L.cs_text_create.argtypes = [vp, C.POINTER(abi.cs_text_params), u32, u64, u32, u32, C.POINTER(vp)]
    L.cs_text_destroy.restype = None
    L.cs_text_destroy.argtypes = [vp]
    L.cs_text_submit.restype = i32
    L.cs_text_submit.argtypes = [vp, u32, vp, u64, vp, u64, u32]
    L.cs_text_wait.restype = i32
    L.cs_text_wait.argtypes = [vp, u32, C.POINTER(abi.cs_text_result)]
    L.cs_text_routes.restype = i32
    L.cs_text_routes.argtypes = [vp, u32, vp, vp, vp]
    L.cs_text_fetch.restype = i32
    L.cs_text_fetch.argtypes = [vp, u32, vp, vp]
    if L.cs_abi_version() != abi.CS_ABI_VERSION:
        raise HipUnavailable(f"ABI mismatch: library {L.cs_abi_version()} vs python {abi.CS_ABI_VERSION}")
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        L = load()
        msg = (L.cs_last_error() or b"").decode(errors="replace")
        if rc == abi.CS_ERR_NO_GPU:
            raise HipUnavailable(msg)
        raise CsError(rc, msg)


def device_count() -> int:
    n = load().cs_device_count()
    if n < 0:
        check(n)
    return n
