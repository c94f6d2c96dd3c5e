"""Consumes ``tests/golden/cutadapt_<version>.json.gz`` -- outputs of REAL cutadapt written by
``tools/pin_against_cutadapt.py`` on a machine where cutadapt is installable -- and holds the CPU oracle
(``-m "not gpu"``) and the HIP path (``-m gpu``) to them.  No fixture (the state of this image: cutadapt is in
neither the reference tree nor the wheelhouse) -> every test here skips, and DESIGN.md section 0 keeps saying
"parity unpinned".  The chain cases replay cutseq/run.py:326-426 / 533-731, the vectors run.py:332-417.
"""
import gzip
import json
from pathlib import Path

import numpy as np
import pytest

from cutseq_amd import abi, plan as planmod, synth
from cutseq_amd.common import BarcodeConfig

import guide_vectors
import util

GOLDEN = Path(__file__).resolve().parent / "golden"
FIXTURES = sorted(GOLDEN.glob("cutadapt_*.json.gz"))
needs_fixture = pytest.mark.skipif(not FIXTURES, reason="no tests/golden/cutadapt_*.json.gz: run tools/pin_against_cutadapt.py "
                                                        "where cutadapt is installable")


def load():
    with gzip.open(FIXTURES[-1], "rt") as fh:
        return json.load(fh)


def case_inputs(case):
    """-> (batch, names1, names2 | None) rebuilt from the case's source spec."""
    src = case["source"]
    if "fixture" in src:
        r1 = util.read_fastq_gz(GOLDEN / "fixture10k_R1.fq.gz")
        r2 = util.read_fastq_gz(GOLDEN / "fixture10k_R2.fq.gz")
        return util.batch_from_records(r1, r2), [r[0] for r in r1], [r[0] for r in r2]
    spec = src["synthetic"]
    batch = synth.generate_pairs(spec["n"], 150, case["scheme"], seed=spec["seed"], single_end=not case["paired"],
                                 poly_fraction=0.15, art5_fraction=0.05, indel_frac=0.2)
    n1 = [s.encode() for s in synth.headers(spec["n"], 1)]
    n2 = [s.encode() for s in synth.headers(spec["n"], 2)] if case["paired"] else None
    return batch, n1, n2


def plan_of(case):
    st = planmod.CutadaptConfig()
    for k, v in case["flags"].items():
        setattr(st, k, v)
    fn = planmod.compile_paired if case["paired"] else planmod.compile_single
    return fn(BarcodeConfig(case["scheme"].replace(" ", "").upper()), st)


def check_chain(case, run):
    """``run(tp, batch) -> (res1, cap2, res2)``; the formatted records must equal cutadapt's, record for record."""
    tp = plan_of(case)
    batch, n1, n2 = case_inputs(case)
    res1, cap2, res2 = run(tp, batch)
    got = util.format_batch(tp, batch, n1, n2, res1, cap2, res2)
    want = case["output"]
    assert len(got) == len(want)
    for i, ((rt, a, b), (wrt, wa, wb)) in enumerate(zip(got, want)):
        assert (rt, a, b) == (wrt, wa.encode(), wb.encode() if wb is not None else None), (case["id"], i)
    assert tp.swap_outputs == case["swap"]


def one_adapter_plan(kind, adapter, rate, mo):
    if kind == "anywhere":
        where, remove_before, rightmost = abi.CS_WHERE_ANYWHERE, False, False
    else:
        _, where_name, remove_before, rightmost = guide_vectors.KINDS[kind]
        where = {"BACK": abi.CS_WHERE_BACK, "FRONT": abi.CS_WHERE_FRONT, "PREFIX": abi.CS_WHERE_PREFIX,
                 "SUFFIX": abi.CS_WHERE_SUFFIX, "BACK_NI": abi.CS_WHERE_BACK_NOT_INTERNAL,
                 "FRONT_NI": abi.CS_WHERE_FRONT_NOT_INTERNAL}[where_name]
    op = planmod.AdapterOp(kind, adapter, rate, min(mo, len(adapter)), where,
                           abi.CS_REMOVE_BEFORE if remove_before else abi.CS_REMOVE_AFTER, rightmost=rightmost,
                           match_flag=abi.CS_F_ADAPTER3)
    return planmod.TrimPlan(r1=planmod.MateChain([op]), r2=None, has_umi=False, min_length=0, untrimmed_filter=False)


def check_vectors(case, run):
    groups = {}
    for kind, adapter, rate, mo, read, kept, _ in case["vectors"]:
        groups.setdefault((kind, adapter, rate, mo), []).append((read, kept))
    for (kind, adapter, rate, mo), items in groups.items():
        tp = one_adapter_plan(kind, adapter, rate, mo)
        batch = util.batch_from_reads([(r, "I" * len(r)) for r, _ in items])
        res1, _, _ = run(tp, batch)
        for i, (read, kept) in enumerate(items):
            assert read[int(res1["start"][i]): int(res1["stop"][i])] == kept, (kind, adapter, read)


def oracle(tp, batch):
    (r1, cap2, _), m2 = util.oracle_run(tp, batch, threads=8)
    return r1, cap2, (m2[0] if m2 else None)


def device(tp, batch):
    from cutseq_amd.engine import TrimEngine
    with TrimEngine(tp, device=0, slots=1, max_reads=max(batch.n, 1), max_stride=batch.stride) as eng:
        return eng.trim(batch.seq1, batch.qual1, batch.len1, batch.seq2, batch.qual2, batch.len2)


@needs_fixture
def test_oracle_equals_cutadapt():
    for case in load()["cases"]:
        (check_chain if case["kind"] == "chain" else check_vectors)(case, oracle)


@needs_fixture
@pytest.mark.gpu
def test_hip_path_equals_cutadapt():
    for case in load()["cases"]:
        (check_chain if case["kind"] == "chain" else check_vectors)(case, device)


def test_pin_script_is_importable_and_says_when_cutadapt_is_missing():
    """The one-command pin must at least load here, and must not pretend: no cutadapt -> exit code 2."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pin_against_cutadapt",
                                                  Path(__file__).resolve().parents[1] / "tools" / "pin_against_cutadapt.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.cutadapt_available():
        pytest.skip("cutadapt is importable here: run the script and commit its fixture")
    assert mod.main() == 2
    assert np.dtype(abi.RESULT_DTYPE).itemsize == 8
