"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header
declares, validates op tables, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import pytest
import torch

from cutseq_amd import abi, capi, plan as planmod
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    from cutseq_amd import build
    build.build()
    return capi.load()


def test_exports_every_declared_symbol(lib):
    header = (ROOT / "include" / "cutseq_hip.h").read_text()
    declared = set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cs_abi_version() == abi.CS_ABI_VERSION


def test_synth_libraries_export_what_their_header_declares():
    """include/cutseq_synth.h: the generator's host form lives in libcutseq_host.so, its device form in
    libcutseq_synth.so (loads without a GPU; no compute call here), and ctypes sees the parameter block as C does."""
    import subprocess, tempfile
    from cutseq_amd import build, synth
    build.build_host()
    build.build_synth()
    header = (ROOT / "include" / "cutseq_synth.h").read_text()
    declared = set(re.findall(r"\b(cs[hd]_[a-z0-9_]+)\s*\(", header))
    assert declared == {"csh_synth_pairs", "csd_synth_pairs", "csd_last_error", "csd_abi_version"}
    host, dev = synth.host_lib(), synth.synth_lib()
    for name in declared:
        assert hasattr(host if name.startswith("csh_") else dev, name), name
    assert dev.csd_abi_version() == 1
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "cutseq_synth.h"\nint main(void){printf("%zu %zu %zu\\n", ' \
          'sizeof(csh_synth_params), offsetof(csh_synth_params, umi5), offsetof(csh_synth_params, n_rate)); return 0;}'
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "p.c").write_text(src)
        subprocess.run(["gcc", "-I", str(ROOT / "include"), "-o", f"{d}/p", f"{d}/p.c"], check=True)
        got = list(map(int, subprocess.run([f"{d}/p"], check=True, capture_output=True, text=True).stdout.split()))
    P = synth._SynthParams
    assert got == [C.sizeof(P), P.umi5.offset, P.n_rate.offset]


def test_struct_layout_matches_header():
    # compile-time truth from the C side: build a tiny probe with the same header
    import subprocess, tempfile
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "cutseq_hip.h"
    int main(void){
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(cs_op), offsetof(cs_op, m), offsetof(cs_op, seq),
             offsetof(cs_op, thr), sizeof(cs_params), sizeof(cs_result), sizeof(cs_cap2), sizeof(cs_stats),
             sizeof(cs_reads));
      return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "p.c").write_text(src)
        subprocess.run(["gcc", "-I", str(ROOT / "include"), "-o", f"{d}/p", f"{d}/p.c"], check=True)
        out = subprocess.run([f"{d}/p"], check=True, capture_output=True, text=True).stdout.split()
    got = list(map(int, out))
    want = [C.sizeof(abi.cs_op), abi.cs_op.m.offset, abi.cs_op.seq.offset, abi.cs_op.thr.offset,
            C.sizeof(abi.cs_params), C.sizeof(abi.cs_result), C.sizeof(abi.cs_cap2), C.sizeof(abi.cs_stats),
            C.sizeof(abi.cs_reads)]
    assert got == want


def _create(lib, tp):
    a1, n1, a2, n2 = tp.pack()
    h = C.c_void_p()
    p = tp.params()
    rc = lib.cs_plan_create(C.cast(a1, C.c_void_p), n1, C.cast(a2, C.c_void_p) if a2 is not None else None, n2,
                            C.byref(p), C.byref(h))
    return rc, h


def test_plan_create_accepts_every_preset(lib):
    for name, scheme in BUILDIN_ADAPTERS.items():
        for paired in (True, False):
            st = planmod.CutadaptConfig()
            st.trim_polyA = True
            st.trim_polyA_wo_direction = True
            bc = BarcodeConfig(scheme)
            tp = (planmod.compile_paired if paired else planmod.compile_single)(bc, st)
            rc, h = _create(lib, tp)
            assert rc == 0, (name, lib.cs_last_error())
            lib.cs_plan_destroy(h)


def test_plan_create_rejects_bad_tables(lib):
    tp = planmod.single_adapter_plan("AGATCGGAAGAGC", 0.1)
    a1, n1, _, _ = tp.pack()
    p = tp.params()
    h = C.c_void_p()
    a1[0].min_overlap = 0
    assert lib.cs_plan_create(C.cast(a1, C.c_void_p), n1, None, 0, C.byref(p), C.byref(h)) == abi.CS_ERR_ARG
    assert b"min_overlap" in lib.cs_last_error()
    a1, n1, _, _ = tp.pack()
    a1[0].thr[5] = 3
    assert lib.cs_plan_create(C.cast(a1, C.c_void_p), n1, None, 0, C.byref(p), C.byref(h)) == abi.CS_ERR_ARG
    a1, n1, _, _ = tp.pack()
    a1[0].kind = 9
    assert lib.cs_plan_create(C.cast(a1, C.c_void_p), n1, None, 0, C.byref(p), C.byref(h)) == abi.CS_ERR_ARG
    p.abi_version = 77
    a1, n1, _, _ = tp.pack()
    assert lib.cs_plan_create(C.cast(a1, C.c_void_p), n1, None, 0, C.byref(p), C.byref(h)) == abi.CS_ERR_ARG


def test_long_barcodes_take_their_own_ops_not_a_table(lib):
    """CS_OP_DEMUX with m + k > CS_DEMUX_MAX_PREFIX: cs_plan_set_demux_ops (host side only here: the checks and the
    candidate table are built without a GPU), and the plan rules around it."""
    import numpy as np
    scheme = "ACACGACGCTCTTCCGATCT(ACGTACGTACGT)NNNNNNNN>AGATCGGAAGAGCACACGTC"
    st = planmod.CutadaptConfig()
    st.demux_barcodes = ["ACGTACGTACGT", "TTGCAACGGTCA", "GGATCCTTAAGC"]
    tp = planmod.compile_paired(BarcodeConfig(scheme), st)
    op = tp.demux
    assert tp.demux_mate == 1 and not op.tabulated and (op.m, op.k) == (12, 2)
    rc, h = _create(lib, tp)
    assert rc == 0, lib.cs_last_error()
    index = next(i for m, i, _ in tp.demux_ops())
    ops = planmod.pack_ops(op.barcode_ops(), limit=255)
    table = np.zeros(10, dtype=np.uint16)
    assert lib.cs_plan_set_demux(h, 1, index, table.ctypes.data, table.size) == abi.CS_ERR_ARG  # no table this long
    assert lib.cs_plan_set_demux_ops(h, 1, index, C.cast(ops, C.c_void_p), 3) == 0, lib.cs_last_error()
    assert lib.cs_plan_set_demux_ops(h, 1, index, C.cast(ops, C.c_void_p), 0) == abi.CS_ERR_ARG
    assert lib.cs_plan_set_demux_ops(h, 1, 0, C.cast(ops, C.c_void_p), 3) == abi.CS_ERR_ARG  # op 0 is an adapter op
    ops[1].seq[3] = ord("N")
    assert lib.cs_plan_set_demux_ops(h, 1, index, C.cast(ops, C.c_void_p), 3) == abi.CS_ERR_ARG
    ops = planmod.pack_ops(op.barcode_ops(), limit=255)
    ops[2].align_flags = abi.CS_WHERE_BACK
    assert lib.cs_plan_set_demux_ops(h, 1, index, C.cast(ops, C.c_void_p), 3) == abi.CS_ERR_ARG
    lib.cs_plan_destroy(h)
    # a 255-plex of 16-mers: the candidate table (16 M list entries) is built in about a second (banded columns,
    # subtrees on up to eight threads; the bound is generous: the container's cores are shared)
    import random, time
    rng = random.Random(1)
    codes = sorted({"".join(rng.choice("ACGT") for _ in range(16)) for _ in range(300)})[:255]
    st.demux_barcodes = codes
    tp = planmod.compile_paired(BarcodeConfig(scheme.replace("ACGTACGTACGT", codes[0])), st)
    rc, h = _create(lib, tp)
    assert rc == 0
    ops = planmod.pack_ops(tp.demux.barcode_ops(), limit=255)
    t0 = time.perf_counter()
    assert lib.cs_plan_set_demux_ops(h, 1, index, C.cast(ops, C.c_void_p), 255) == 0, lib.cs_last_error()
    assert time.perf_counter() - t0 < 30.0
    lib.cs_plan_destroy(h)
    # a 255-plex of 20-mers with four errors: the lists of nine-base prefixes exceed the table format, the library
    # settles for shorter prefixes instead of refusing the plan
    codes = sorted({"".join(rng.choice("ACGT") for _ in range(20)) for _ in range(300)})[:255]
    st.demux_barcodes = codes
    tp = planmod.compile_paired(BarcodeConfig(scheme.replace("ACGTACGTACGT", codes[0])), st)
    assert (tp.demux.m, tp.demux.k) == (20, 4)
    rc, h = _create(lib, tp)
    assert rc == 0
    ops = planmod.pack_ops(tp.demux.barcode_ops(), limit=255)
    assert lib.cs_plan_set_demux_ops(h, 1, index, C.cast(ops, C.c_void_p), 255) == 0, lib.cs_last_error()
    lib.cs_plan_destroy(h)
    # limits and placement
    st.demux_barcodes = ["A" * 21 + "C", "C" * 21 + "A"]  # m + k = 22 + 4
    with pytest.raises(ValueError):
        planmod.compile_paired(BarcodeConfig(scheme.replace("ACGTACGTACGT", "A" * 21 + "C")), st)
    st.demux_barcodes = ["ACGTACGTAC", "TTGCATGCAA"]
    only3 = "ACACGACGCTCTTCCGATCTNNNNNNNN>(ACGTACGTAC)AGATCGGAAGAGCACACGTC"
    tp = planmod.compile_paired(BarcodeConfig(only3), st)
    assert tp.demux_mate == 2 and tp.demux.barcodes == ["GTACGTACGT", "TTGCATGCAA"]  # as R2 reads them
    se = planmod.compile_single(BarcodeConfig(only3), st)  # single-end: the barcode ends the read (SuffixAdapter ops)
    assert se.demux.at_end and not se.demux.tabulated and se.demux.barcodes == ["ACGTACGTAC", "TTGCATGCAA"]
    rc, h = _create(lib, se)
    assert rc == 0
    index = next(i for m, i, _ in se.demux_ops())
    ops = planmod.pack_ops(se.demux.barcode_ops(), limit=255)
    assert ops[0].align_flags == abi.CS_WHERE_SUFFIX
    assert lib.cs_plan_set_demux_ops(h, 1, index, C.cast(ops, C.c_void_p), 2) == 0, lib.cs_last_error()
    ops[0].align_flags = abi.CS_WHERE_PREFIX
    assert lib.cs_plan_set_demux_ops(h, 1, index, C.cast(ops, C.c_void_p), 2) == abi.CS_ERR_ARG
    lib.cs_plan_destroy(h)


@pytest.mark.skipif(torch.cuda.is_available(), reason="GPU present: covered by the gpu suite")
def test_engine_fails_loudly_without_gpu(lib):
    """No GPU in the build container: the product must raise, never fall back to a CPU path."""
    from cutseq_amd.engine import TrimEngine
    tp = planmod.single_adapter_plan("AGATCGGAAGAGC", 0.1)
    with pytest.raises(capi.HipUnavailable):
        TrimEngine(tp, device=0, slots=0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under cutseq_amd/ may reference it."""
    for path in (ROOT / "cutseq_amd").rglob("*.py"):
        text = path.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), path
    for path in (ROOT / "cutseq_amd" / "csrc").iterdir():
        assert "oracle" not in path.read_text(), path
