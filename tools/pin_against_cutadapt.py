#!/usr/bin/env python3
"""Pin the oracle (and through it the HIP path) against REAL cutadapt -- the day ``import cutadapt`` works.

    python tools/pin_against_cutadapt.py            # writes tests/golden/cutadapt_<version>.json.gz

The arithmetic of the hot path lives in ``cutadapt~=5.0`` (reference pyproject.toml:17), which is in neither
the reference tree nor this image, so every parity claim of this repository is "against our reading of
cutadapt" (DESIGN.md section 0).  This script is the one-command way out: it assembles the modifier chains
from real cutadapt classes the way the reference does (cutseq/run.py:326-426 single-end, 533-731 paired; own
code, table-driven, independent of ``cutseq_amd/plan.py``), runs

  * the reference's own 10 000 pairs (tests/golden/fixture10k_R{1,2}.fq.gz = reference test/input_R{1,2}.fq.gz)
    under ``-A TAKARAV3`` with and without ``--trim-polyA``,
  * every preset x flag combination of tests/test_oracle.py:CHAIN_CASES on seeded synthetic reads,
  * the published-guide vectors (tests/guide_vectors.py) and the adversarial single-adapter generators of
    tests/test_gpu_parity.py for every adapter class the reference uses,

and writes inputs (or their generator spec) + cutadapt's outputs to one fixture.  ``tests/test_cutadapt_pin.py``
consumes the fixture when it is there (CPU: oracle, ``-m gpu``: HIP path) and skips otherwise.  ``bench.py`` calls
:func:`time_cutadapt_chain` for the ``cpu_baseline_cutadapt`` leg when cutadapt is importable.

Without cutadapt the script says so and exits 2; nothing in the product imports it.
"""
from __future__ import annotations

import gzip
import json
import random
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

POLY_LENGTH, POLY_RATE, MAX_ERRORS = 100, 0.15, 0.2  # cutseq/run.py:326, 389-390, 533, 674-675


def cutadapt_available() -> bool:
    try:
        import cutadapt  # noqa: F401
        import dnaio  # noqa: F401
        return True
    except Exception:
        return False


def cutadapt_version() -> str:
    import cutadapt
    return getattr(cutadapt, "__version__", "unknown")


# ------------------------------------------------------------------------------------------------ chain assembly


def _classes():
    from cutadapt import adapters as A, modifiers as M
    from cutadapt.info import ModificationInfo
    from cutadapt.predicates import TooShort

    class ReadThroughCutter(M.SingleEndModifier):
        """The reference's ConditionalCutter (cutseq/run.py:113-161), restated: cut unless the read has no
        adapter match AND is shorter than the forcing length."""

        def __init__(self, length, force_min_len):
            self.length, self.force_min_len = length, force_min_len

        def __call__(self, read, info):
            if not info.matches and len(read.sequence) < self.force_min_len:
                return read
            if self.length > 0:
                info.cut_prefix = read.sequence[: self.length]
                return read[self.length:]
            if self.length < 0:
                info.cut_suffix = read.sequence[self.length:]
                return read[: self.length]
            return read  # (the reference returns None here; a zero-length cut is never compiled)

    class ReverseComplement(M.SingleEndModifier):
        """cutseq/run.py:164-196: sequence reverse-complemented, qualities reversed."""

        def __call__(self, read, info):
            from cutseq_amd.common import reverse_complement
            rc = read[:]
            rc.sequence = reverse_complement(read.sequence)
            rc.qualities = read.qualities[::-1] if read.qualities is not None else None
            return rc

    return A, M, ModificationInfo, TooShort, ReadThroughCutter, ReverseComplement


def _get(settings, name, default):
    return getattr(settings, name, default)


def build_chain(barcode, settings, paired: bool, untrimmed_requested: bool = False):
    """-> dict(mods=[...], refs=(ref adapters mate 1, mate 2), untrimmed_filter, swap, paired).
    ``mods`` entries: ("each", m1, m2 | None) applies m1 / m2 to the mates separately, ("pair", m) is a
    paired modifier (PairedEndRenamer).  Step numbers follow the reference's comments."""
    A, M, _, _, ReadThroughCutter, ReverseComplement = _classes()
    mods = []
    fmin = _get(settings, "force_trim_min_length", 50)
    cond = bool(_get(settings, "conditional_cutter", True))
    anywhere = bool(_get(settings, "force_anywhere", False))

    def cutter(adapter):
        return M.AdapterCutter([adapter], times=1)

    def maybe_conditional(length):
        return ReadThroughCutter(length, fmin) if cond else M.UnconditionalCutter(length)

    def each(m1, m2=None):
        mods.append(("each", m1, m2 if paired else None))

    # step 1: name suffixes
    for a, b in ((".1", ".2"), ("/1", "/2")):
        each(M.SuffixRemover(a), M.SuffixRemover(b))
    # step 2 / 3: 5' artefact, 3' read-through
    each(cutter(A.RightmostFrontAdapter(sequence=barcode.p5.fw, max_errors=MAX_ERRORS, min_overlap=10)),
         cutter(A.RightmostFrontAdapter(sequence=barcode.p7.rc, max_errors=MAX_ERRORS, min_overlap=10)))
    each(cutter(A.BackAdapter(sequence=barcode.p7.fw, max_errors=MAX_ERRORS, min_overlap=3, force_anywhere=anywhere)),
         cutter(A.BackAdapter(sequence=barcode.p5.rc, max_errors=MAX_ERRORS, min_overlap=3, force_anywhere=anywhere)))
    # step 4: inline barcodes
    inline5 = inline3 = None
    if barcode.inline5.len > 0:
        inline5 = A.PrefixAdapter(sequence=barcode.inline5.fw, max_errors=MAX_ERRORS)
        each(cutter(inline5), M.UnconditionalCutter(-barcode.inline5.len))
    if barcode.inline3.len > 0:
        if paired:
            inline3 = A.PrefixAdapter(sequence=barcode.inline3.rc, max_errors=MAX_ERRORS)
            each(M.UnconditionalCutter(-barcode.inline3.len), cutter(inline3))
        else:
            inline3 = A.SuffixAdapter(sequence=barcode.inline3.fw, max_errors=MAX_ERRORS)
            each(cutter(inline3))
    # step 5: UMIs, then the renamer (later cutters overwrite cut_prefix / cut_suffix)
    has_umi = barcode.umi5.len + barcode.umi3.len > 0
    if barcode.umi5.len > 0:
        each(M.UnconditionalCutter(barcode.umi5.len), maybe_conditional(-barcode.umi5.len))
    if barcode.umi3.len > 0:
        if paired:
            each(maybe_conditional(-barcode.umi3.len), M.UnconditionalCutter(barcode.umi3.len))
        else:
            each(M.UnconditionalCutter(-barcode.umi3.len))
    if paired:
        mods.append(("pair", M.PairedEndRenamer("{id}_{r1.cut_prefix}{r2.cut_prefix}" if has_umi else "{id}")))
    else:
        each(M.Renamer("{id}_{cut_prefix}{cut_suffix}" if has_umi else "{id}"))
    # step 6: masks
    if barcode.mask5.len > 0:
        each(M.UnconditionalCutter(barcode.mask5.len), maybe_conditional(-barcode.mask5.len))
    if barcode.mask3.len > 0:
        if paired:
            each(maybe_conditional(-barcode.mask3.len), M.UnconditionalCutter(barcode.mask3.len))
        else:
            each(M.UnconditionalCutter(-barcode.mask3.len))
    # step 7: poly-A / poly-T
    if _get(settings, "trim_polyA", False):
        def tail():
            return cutter(A.NonInternalBackAdapter(sequence="A" * POLY_LENGTH, max_errors=POLY_RATE))

        def head():
            return cutter(A.NonInternalFrontAdapter(sequence="T" * POLY_LENGTH, max_errors=POLY_RATE))

        if _get(settings, "trim_polyA_wo_direction", False):
            each(tail(), head())
            each(head(), tail())
        elif barcode.strand == "+":
            each(tail(), head())
        elif barcode.strand == "-":
            each(head(), tail())
    # step 8
    q = _get(settings, "min_quality", 20)
    each(M.QualityTrimmer(cutoff_front=0, cutoff_back=q), M.QualityTrimmer(cutoff_front=0, cutoff_back=q))
    # step 9
    swap = False
    if _get(settings, "auto_rc", False) and barcode.strand == "-":
        if paired:
            swap = True
        else:
            each(ReverseComplement())
    untrimmed_filter = (barcode.inline5.len + barcode.inline3.len > 0 and _get(settings, "ensure_inline_barcode", False)) \
        or untrimmed_requested
    if paired:
        refs = ([inline5] if inline5 else [], [inline3] if inline3 else [])
    else:
        refs = ([a for a in (inline5, inline3) if a is not None], [])
    return dict(mods=mods, refs=refs, untrimmed_filter=untrimmed_filter, swap=swap, paired=paired,
                min_length=_get(settings, "min_length", 20))


def _untrimmed_any(refs, info) -> bool:
    seen = [m.adapter for m in info.matches]  # IsUntrimmedAny.test, cutseq/run.py:97-110
    return any(a not in seen for a in refs)


def run_chain(chain, rec1, rec2=None):
    """One read (pair) through the chain -> (route, record bytes mate 1, record bytes mate 2 | None);
    routes as the product numbers them: 0 trimmed, 1 short, 2 untrimmed."""
    import dnaio
    _, _, ModificationInfo, TooShort, _, _ = _classes()
    r1 = dnaio.SequenceRecord(*rec1)
    i1 = ModificationInfo(r1)
    r2 = i2 = None
    if chain["paired"]:
        r2 = dnaio.SequenceRecord(*rec2)
        i2 = ModificationInfo(r2)
    for entry in chain["mods"]:
        if entry[0] == "pair":
            r1, r2 = entry[1](r1, r2, i1, i2)
        else:
            r1 = entry[1](r1, i1)
            if r2 is not None and entry[2] is not None:
                r2 = entry[2](r2, i2)
    short = TooShort(chain["min_length"])
    if short.test(r1, i1) or (r2 is not None and short.test(r2, i2)):
        route = 1
    elif chain["untrimmed_filter"] and (_untrimmed_any(chain["refs"][0], i1) or
                                        (r2 is not None and _untrimmed_any(chain["refs"][1], i2))):
        route = 2
    else:
        route = 0

    def text(r):
        return f"@{r.name}\n{r.sequence}\n+\n{r.qualities}\n".encode()

    return route, text(r1), (text(r2) if r2 is not None else None)


def single_adapter(kind: str, sequence: str, rate: float, min_overlap: int):
    """One AdapterCutter of the class ``kind`` names (tests/guide_vectors.py:KINDS)."""
    A, M, _, _, _, _ = _classes()
    cls = {"back": A.BackAdapter, "front": A.FrontAdapter, "prefix": A.PrefixAdapter, "suffix": A.SuffixAdapter,
           "back_ni": A.NonInternalBackAdapter, "front_ni": A.NonInternalFrontAdapter,
           "rightmost_front": A.RightmostFrontAdapter, "anywhere": A.BackAdapter}[kind]
    kw = dict(sequence=sequence, max_errors=rate)
    if kind not in ("prefix", "suffix"):
        kw["min_overlap"] = min_overlap
    if kind == "anywhere":
        kw["force_anywhere"] = True
    return M.AdapterCutter([cls(**kw)], times=1)


def trim_single(cutter, sequence: str) -> str:
    import dnaio
    _, _, ModificationInfo, _, _, _ = _classes()
    r = dnaio.SequenceRecord("r", sequence, "I" * len(sequence))
    return cutter(r, ModificationInfo(r)).sequence


# ------------------------------------------------------------------------------------------------ cases


def _settings(flags: dict):
    from cutseq_amd import plan as planmod
    st = planmod.CutadaptConfig()
    for k, v in flags.items():
        setattr(st, k, v)
    return st


def _records(names, seq, qual, lens):
    return [(names[i].decode(), seq[i, : int(lens[i])].tobytes().decode(), qual[i, : int(lens[i])].tobytes().decode())
            for i in range(len(names))]


def chain_case(case_id: str, scheme: str, flags: dict, paired: bool, source: dict, rec1, rec2):
    from cutseq_amd.common import BarcodeConfig
    chain = build_chain(BarcodeConfig(scheme.replace(" ", "").upper()), _settings(flags), paired)
    out = []
    for i in range(len(rec1)):
        route, a, b = run_chain(chain, rec1[i], rec2[i] if paired else None)
        out.append([route, a.decode(), b.decode() if b is not None else None])
    return {"id": case_id, "kind": "chain", "scheme": scheme, "flags": flags, "paired": paired, "source": source,
            "swap": chain["swap"], "output": out}


def fixture_records(limit=None):
    import util
    g = ROOT / "tests" / "golden"
    r1 = util.read_fastq_gz(g / "fixture10k_R1.fq.gz", limit)
    r2 = util.read_fastq_gz(g / "fixture10k_R2.fq.gz", limit)
    return ([(n.decode(), s.decode(), q.decode()) for n, s, q in r1],
            [(n.decode(), s.decode(), q.decode()) for n, s, q in r2])


def synthetic_records(scheme: str, n: int, seed: int, paired: bool):
    """The generator spec tests/test_cutadapt_pin.py replays: synth.generate_pairs(n, 150, scheme, seed=seed, ...)."""
    from cutseq_amd import synth
    batch = synth.generate_pairs(n, 150, scheme, seed=seed, single_end=not paired, poly_fraction=0.15,
                                 art5_fraction=0.05, indel_frac=0.2)
    n1 = [s.encode() for s in synth.headers(n, 1)]
    rec1 = _records(n1, batch.seq1, batch.qual1, batch.len1)
    rec2 = _records([s.encode() for s in synth.headers(n, 2)], batch.seq2, batch.qual2, batch.len2) if paired else None
    return rec1, rec2


def collect_cases(n_synth: int = 2000):
    from cutseq_amd.common import BUILDIN_ADAPTERS
    import guide_vectors
    from test_oracle import CHAIN_CASES
    cases = []
    rec1, rec2 = fixture_records()
    for flags in ({}, {"trim_polyA": True}):
        cases.append(chain_case(f"fixture10k/TAKARAV3/{sorted(flags)}", BUILDIN_ADAPTERS["TAKARAV3"], flags, True,
                                {"fixture": "fixture10k"}, rec1, rec2))
    for idx, (name, flags, paired) in enumerate(CHAIN_CASES):
        if any(k in flags for k in ("shortcut", "indel_tie", "case_rule")):
            continue  # switches of OUR restatement, not cutadapt options
        scheme = BUILDIN_ADAPTERS.get(name, name)
        r1, r2 = synthetic_records(scheme, n_synth, 100 + idx, paired)
        cases.append(chain_case(f"synthetic/{name[:24]}/{idx}", scheme, flags, paired,
                                {"synthetic": {"n": n_synth, "seed": 100 + idx}}, r1, r2))
    # single-adapter vectors: the guide's, then randomized adversarial reads per adapter class
    vec = []
    for group, kind, adapter, rate, mo, read, kept in guide_vectors.all_vectors():
        vec.append([kind, adapter, rate, mo, read, trim_single(single_adapter(kind, adapter, rate, mo), read), kept])
    cases.append({"id": "guide_vectors", "kind": "single_adapter", "vectors": vec})
    from test_gpu_parity import adversarial_reads
    rng = random.Random(20260101)
    adv = []
    for kind, adapter, rate, mo in (("back", "AGATCGGAAGAGCACACGTC", 0.2, 3), ("rightmost_front", "ACACGACGCTCTTCCGATCT", 0.2, 10),
                                    ("anywhere", "AGATCGGAAGAGCACACGTC", 0.2, 3), ("prefix", "ATCACG", 0.2, 6),
                                    ("suffix", "CGATGT", 0.2, 6), ("front", "ACACGACGCTCTTCCGATCT", 0.2, 3),
                                    ("back", "AGATCGGAAGAGC", 0.1, 3), ("back_ni", "A" * 100, 0.15, 3),
                                    ("front_ni", "T" * 100, 0.15, 3)):
        cutter = single_adapter(kind, adapter, rate, mo)
        ref = adapter if len(set(adapter)) > 1 else adapter[:30]
        for read, _ in adversarial_reads(rng, ref, 3000, "ACGT"):
            adv.append([kind, adapter, rate, mo, read, trim_single(cutter, read), None])
    cases.append({"id": "adversarial", "kind": "single_adapter", "vectors": adv})
    return cases


def time_cutadapt_chain(workload: str, batch, m: int) -> dict:
    """bench.py's ``cpu_baseline_cutadapt`` leg: the real cutadapt chain of the workload, one thread, ``m`` pairs."""
    from cutseq_amd import synth, workloads
    from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig
    if workload not in ("config3", "config4"):
        return {"skipped": f"{workload} is not expressible as a cutseq CLI run"}
    scheme = BUILDIN_ADAPTERS["TAKARAV3"] if workload == "config3" else workloads.CONFIG4_SCHEME
    flags = {"trim_polyA": True} if workload == "config3" else {"ensure_inline_barcode": True}
    chain = build_chain(BarcodeConfig(scheme), _settings(flags), True)
    rec1 = _records([s.encode() for s in synth.headers(m, 1)], batch.seq1[:m], batch.qual1[:m], batch.len1[:m])
    rec2 = _records([s.encode() for s in synth.headers(m, 2)], batch.seq2[:m], batch.qual2[:m], batch.len2[:m])
    t0 = time.perf_counter()
    for i in range(m):
        run_chain(chain, rec1[i], rec2[i])
    dt = time.perf_counter() - t0
    return {"value": round(m / dt / 1e6, 5), "unit": "M read-pairs/s", "cores": 1, "kind": "reference",
            "sample": f"first {m} pairs, cutadapt {cutadapt_version()} modifier chain assembled as cutseq/run.py:533-731 does, "
                      f"in-process, one thread, {dt:.1f} s"}


def main():
    if not cutadapt_available():
        print("cutadapt (and dnaio) are not importable in this environment: nothing to pin against.\n"
              "Run this script where `pip install 'cutadapt~=5.0'` is possible and commit the fixture it writes.",
              file=sys.stderr)
        return 2
    cases = collect_cases()
    out = ROOT / "tests" / "golden" / f"cutadapt_{cutadapt_version()}.json.gz"
    with gzip.open(out, "wt") as fh:
        json.dump({"cutadapt_version": cutadapt_version(), "generator": "tools/pin_against_cutadapt.py", "cases": cases}, fh)
    print(f"wrote {out} ({len(cases)} cases)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
