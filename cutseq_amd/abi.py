"""ctypes mirror of ``include/cutseq_hip.h`` (POD structs and constants only).

No library is loaded here: ``capi.py`` binds the HIP library, ``oracle/`` binds the CPU
checker against the same data contract.
"""
from __future__ import annotations

import ctypes as C

CS_ABI_VERSION = 6
CS_MAX_ADAPTER = 128
CS_MAX_OPS = 24
CS_MAX_STRIDE = 1536
CS_LEN_SKIP = 0xFFFF
CS_MAX_READ = 1 << 24

CS_OP_ADAPTER, CS_OP_CUT, CS_OP_QTRIM, CS_OP_DEMUX = 1, 2, 3, 4
CS_DEMUX_NONE = 0xFF
CS_DEMUX_MAX_PREFIX = 11
CS_DEMUX_MAX_LONG = 24
CS_DEMUX_BY_OPS = 1

CS_REF_START, CS_QUERY_START, CS_REF_END, CS_QUERY_STOP = 1, 2, 4, 8
CS_WHERE_BACK = 14
CS_WHERE_FRONT = 11
CS_WHERE_PREFIX = 8
CS_WHERE_SUFFIX = 2
CS_WHERE_FRONT_NOT_INTERNAL = 9
CS_WHERE_BACK_NOT_INTERNAL = 6
CS_WHERE_ANYWHERE = 15

CS_REMOVE_BEFORE, CS_REMOVE_AFTER = 0, 1
CS_SHORTCUT_NONE, CS_SHORTCUT_FIND = 0, 1
CS_SELECT_LEFTMOST, CS_SELECT_SCORE = 0, 1
CS_CASE_FOLD, CS_CASE_SENSITIVE = 0, 1
CS_TIE_INSERTION, CS_TIE_DELETION = 0, 1

CS_F_ADAPTER5 = 0x01
CS_F_ADAPTER3 = 0x02
CS_F_INLINE = 0x04
CS_F_POLY = 0x08
CS_F_QTRIMMED = 0x10
CS_F_TOO_SHORT = 0x20
CS_F_UNTRIMMED = 0x40
CS_F_AMBIGUOUS = 0x80

CS_OK = 0
CS_ERR_ARG, CS_ERR_HIP, CS_ERR_NO_GPU, CS_ERR_NOMEM, CS_ERR_STATE = -1, -2, -3, -4, -5


class cs_op(C.Structure):
    _fields_ = [
        ("kind", C.c_uint8),
        ("align_flags", C.c_uint8),
        ("reversed", C.c_uint8),
        ("remove", C.c_uint8),
        ("shortcut", C.c_uint8),
        ("match_flag", C.c_uint8),
        ("required", C.c_uint8),
        ("conditional", C.c_uint8),
        ("capture", C.c_uint8),
        ("homopolymer", C.c_uint8),
        ("q_base", C.c_uint8),
        ("stat_slot", C.c_uint8),
        ("m", C.c_uint16),
        ("k", C.c_uint16),
        ("min_overlap", C.c_uint16),
        ("cut_len", C.c_int16),
        ("force_min_len", C.c_uint16),
        ("q_cutoff", C.c_int16),
        ("seq", C.c_uint8 * CS_MAX_ADAPTER),
        ("thr", C.c_uint8 * (CS_MAX_ADAPTER + 1)),
        ("_pad", C.c_uint8 * 3),
    ]


class cs_params(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("min_length", C.c_uint16),
        ("select_rule", C.c_uint8),
        ("use_filter", C.c_uint8),
        ("case_rule", C.c_uint8),
        ("indel_tie", C.c_uint8),
        ("reserved8", C.c_uint8 * 2),
        ("reserved", C.c_uint32 * 5),
    ]


class cs_result(C.Structure):
    _fields_ = [
        ("start", C.c_uint16),
        ("stop", C.c_uint16),
        ("cap_off", C.c_uint16),
        ("cap_len", C.c_uint8),
        ("flags", C.c_uint8),
    ]


class cs_cap2(C.Structure):
    _fields_ = [("off", C.c_uint16), ("len", C.c_uint8), ("_pad", C.c_uint8)]


class cs_stats(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64),
        ("in_bp", C.c_uint64),
        ("out_bp", C.c_uint64),
        ("qualtrim_bp", C.c_uint64),
        ("n_too_short", C.c_uint64),
        ("n_untrimmed", C.c_uint64),
        ("n_exact_dp", C.c_uint64),
        ("n_refiltered", C.c_uint64),
        ("op_matched", C.c_uint64 * CS_MAX_OPS),
    ]

    def as_dict(self) -> dict:
        d = {f: int(getattr(self, f)) for f, _ in self._fields_ if not f.startswith("_") and f != "op_matched"}
        d["op_matched"] = [int(x) for x in self.op_matched]
        return d


class cs_reads(C.Structure):
    _fields_ = [
        ("seq", C.c_void_p),
        ("qual", C.c_void_p),
        ("len", C.c_void_p),
        ("out", C.c_void_p),
        ("cap2", C.c_void_p),
        ("bc", C.c_void_p),
    ]


CS_TEXT_OK, CS_TEXT_ERR_MALFORMED, CS_TEXT_ERR_TOO_LONG, CS_TEXT_ERR_IDS_DIFFER, CS_TEXT_ERR_LINE_COUNT = 0, 1, 2, 3, 4


class cs_text_params(C.Structure):
    _fields_ = [
        ("has_umi", C.c_uint8),
        ("untrimmed_filter", C.c_uint8),
        ("reverse_complement", C.c_uint8),
        ("compress", C.c_uint8),
        ("max_tag", C.c_uint32),
        ("suffix1", C.c_char_p * 2),
        ("suffix2", C.c_char_p * 2),
        ("n_bins", C.c_uint32),
        ("fasta_out", C.c_uint8),
        ("_reserved", C.c_uint8 * 3),
    ]


class cs_text_result(C.Structure):
    _fields_ = [
        ("error", C.c_int32),
        ("error_record", C.c_uint32),
        ("max_len", C.c_uint32),
        ("n_records", C.c_uint32),
        ("route_count", C.c_uint32 * 3),
        ("_pad", C.c_uint32),
        ("route_bytes", (C.c_uint64 * 2) * 3),
        ("out_bytes", C.c_uint64 * 2),
        ("written_bp", C.c_uint64 * 2),
        ("n_lines", C.c_uint32 * 2),
        ("n_long", C.c_uint32 * 2),
        ("text_bytes", (C.c_uint64 * 2) * 3),
    ]


assert C.sizeof(cs_text_params) == 48 and C.sizeof(cs_text_result) == 176
assert C.sizeof(cs_op) == 284, C.sizeof(cs_op)
assert C.sizeof(cs_result) == 8
assert C.sizeof(cs_cap2) == 4
assert C.sizeof(cs_params) == 32
assert C.sizeof(cs_stats) == 8 * (8 + CS_MAX_OPS)

# numpy views of the result records
import numpy as _np  # noqa: E402

RESULT_DTYPE = _np.dtype(
    [("start", "<u2"), ("stop", "<u2"), ("cap_off", "<u2"), ("cap_len", "u1"), ("flags", "u1")]
)
CAP2_DTYPE = _np.dtype([("off", "<u2"), ("len", "u1"), ("_pad", "u1")])
assert RESULT_DTYPE.itemsize == 8 and CAP2_DTYPE.itemsize == 4
