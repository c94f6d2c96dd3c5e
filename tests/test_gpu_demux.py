"""CS_OP_DEMUX (extension, BASELINE.json config 5): one device pass against B independent
``--ensure-inline-barcode`` runs of the oracle -- the parity definition of SURVEY.md 8 f-4."""
import random

import numpy as np
import pytest

from cutseq_amd import abi, demux, plan as planmod, synth
from cutseq_amd.common import BarcodeConfig
from cutseq_amd.engine import TrimEngine

import util

pytestmark = pytest.mark.gpu


def edit_distance(a: str, b: str) -> int:
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def barcode_set(rng: random.Random, count: int, length: int, min_dist: int):
    codes = []
    while len(codes) < count:
        c = util.random_dna(rng, length)
        if all(edit_distance(c, o) >= min_dist for o in codes):
            codes.append(c)
    return codes


def scheme_with(code: str) -> str:
    return f"ACACGACGCTCTTCCGATCT({code})NNNNNNNN>AGATCGGAAGAGCACACGTC"


def plant_barcodes(rng, batch, codes, length, mate=1):
    """Overwrite the inline barcode at the start of every R1 (R2: ``codes`` as that mate reads them) with one of
    `codes` (sometimes damaged, sometimes foreign)."""
    truth = []
    seqs, lens = (batch.seq1, batch.len1) if mate == 1 else (batch.seq2, batch.len2)
    for i in range(batch.n):
        u = rng.random()
        if u < 0.08:
            code, which = util.random_dna(rng, length), -1
        else:
            which = rng.randrange(len(codes))
            code = codes[which]
            if u < 0.3:
                code = util.mutate(rng, code, 1)
            elif u < 0.35:
                code = util.mutate(rng, code, 2)
        row = seqs[i]
        tail = bytes(row[length:int(lens[i])])
        new = (code.encode() + tail)[: int(lens[i])]
        row[: len(new)] = np.frombuffer(new, dtype=np.uint8)
        truth.append(which)
    return truth


@pytest.mark.parametrize("paired,length,count", [(True, 8, 24), (False, 6, 12), (True, 12, 48), (False, 10, 24), (True, 16, 12), (True, 20, 255)])
def test_demux_equals_independent_runs(paired, length, count):
    """8- and 6-base barcodes: the table over every prefix of m + k bases; 10, 12 and 16 bases (m + k = 12, 14, 19): no
    such table fits, the device runs the candidates' own PrefixAdapter ops (cs_plan_set_demux_ops).  255 barcodes of 20
    bases with four errors: the candidate lists of nine-base prefixes exceed the table format, the library settles for
    eight-base prefixes (same results, a few more candidates per read)."""
    rng = random.Random(length * 100 + count)
    codes = barcode_set(rng, count, length, 4)
    st = planmod.CutadaptConfig()
    st.ensure_inline_barcode = True
    st.trim_polyA = True
    n = 6000
    batch = synth.generate_pairs(n, 150, scheme_with(codes[0]), seed=length, single_end=not paired, art5_fraction=0.01)
    plant_barcodes(rng, batch, codes, length)
    if paired:  # a few very short mates: prefixes shorter than m + k
        batch.len1[:80] = np.arange(80, dtype=np.uint16) % (length + 4)
    compile_ = planmod.compile_paired if paired else planmod.compile_single

    st.demux_barcodes = codes
    tp = compile_(BarcodeConfig(scheme_with(codes[0])), st)
    assert tp.demux is not None and tp.untrimmed_filter
    assert tp.demux.tabulated == (length <= 9)
    bc = np.empty(n, dtype=np.uint8)
    with TrimEngine(tp, device=0, slots=1, max_reads=n, max_stride=batch.stride) as eng:
        res = eng.submit(0, batch.seq1, batch.qual1, batch.len1, batch.seq2, batch.qual2, batch.len2, bc=bc)
        eng.wait(0)
        g1, _, g2 = res
        gst1, _ = eng.stats()
    amb = (g1["flags"] & abi.CS_F_AMBIGUOUS) != 0  # reads that more than one barcode claims at this error rate
    assert amb.mean() < 0.01

    # B independent runs of the reference shape: --ensure-inline-barcode with ONE barcode each (CPU oracle)
    st.demux_barcodes = None
    runs = []
    for code in codes:
        one = compile_(BarcodeConfig(scheme_with(code)), st)
        (o1, _, _), m2 = util.oracle_run(one, batch, threads=8)
        runs.append((o1, m2[0] if m2 else None))
    matched = np.stack([(o1["flags"] & abi.CS_F_INLINE) != 0 for o1, _ in runs])  # [barcode, read]
    assert np.array_equal(matched.sum(axis=0) > 1, amb)  # the flag marks exactly the reads several runs would claim
    want_bc = np.where(matched.any(axis=0), matched.argmax(axis=0), abi.CS_DEMUX_NONE).astype(np.uint8)
    assert np.array_equal(bc[~amb], want_bc[~amb])
    assert np.all(matched[bc[amb].astype(np.int64), np.nonzero(amb)[0]])  # an ambiguous read goes to one of its claimants
    assert 0.5 < float((bc != abi.CS_DEMUX_NONE).mean()) < 0.99
    own = np.where(bc == abi.CS_DEMUX_NONE, 0, bc).astype(np.int64)  # unassigned reads look the same in every run
    pick = lambda arrs: np.stack(arrs)[own, np.arange(n)]
    want1 = pick([o1 for o1, _ in runs])
    want1["flags"] |= np.where(amb, abi.CS_F_AMBIGUOUS, 0).astype(np.uint8)
    assert np.array_equal(g1, want1)
    if paired:
        assert np.array_equal(g2, pick([o2 for _, o2 in runs]))
    assert int(gst1.op_matched[2]) == int((bc != abi.CS_DEMUX_NONE).sum())  # op 2 of mate 1 is the demultiplexer


@pytest.mark.parametrize("length,count", [(8, 16), (12, 24)])
def test_demux_on_the_3prime_barcode_at_the_start_of_r2(length, count):
    """A scheme whose only inline barcode sits at the 3' end: R2 starts with its reverse complement
    (cutseq/run.py:604-608), so the demultiplexing op goes into R2's chain and R1 loses the barcode's length where
    the reference cuts it (run.py:600-603).  Parity: one --ensure-inline-barcode run per barcode."""
    from cutseq_amd.common import reverse_complement

    def scheme3(code):
        return f"ACACGACGCTCTTCCGATCTNNNNNNNN>({code})AGATCGGAAGAGCACACGTC"

    rng = random.Random(length + count)
    codes = barcode_set(rng, count, length, 4)
    st = planmod.CutadaptConfig()
    st.ensure_inline_barcode = True
    n = 5000
    batch = synth.generate_pairs(n, 150, scheme3(codes[0]), seed=length)
    plant_barcodes(rng, batch, [reverse_complement(c) for c in codes], length, mate=2)
    batch.len2[:60] = np.arange(60, dtype=np.uint16) % (length + 4)
    st.demux_barcodes = codes
    tp = planmod.compile_paired(BarcodeConfig(scheme3(codes[0])), st)
    assert tp.demux_mate == 2 and tp.demux.barcodes[0] == reverse_complement(codes[0])
    bc = np.empty(n, dtype=np.uint8)
    with TrimEngine(tp, device=0, slots=1, max_reads=n, max_stride=batch.stride) as eng:
        g1, _, g2 = eng.submit(0, batch.seq1, batch.qual1, batch.len1, batch.seq2, batch.qual2, batch.len2, bc=bc)
        eng.wait(0)
    amb = (g2["flags"] & abi.CS_F_AMBIGUOUS) != 0
    st.demux_barcodes = None
    runs = []
    for code in codes:
        one = planmod.compile_paired(BarcodeConfig(scheme3(code)), st)
        (o1, _, _), m2 = util.oracle_run(one, batch, threads=8)
        runs.append((o1, m2[0]))
    matched = np.stack([(o2["flags"] & abi.CS_F_INLINE) != 0 for _, o2 in runs])
    assert np.array_equal(matched.sum(axis=0) > 1, amb)
    want_bc = np.where(matched.any(axis=0), matched.argmax(axis=0), abi.CS_DEMUX_NONE).astype(np.uint8)
    assert np.array_equal(bc[~amb], want_bc[~amb])
    assert 0.5 < float((bc != abi.CS_DEMUX_NONE).mean()) < 0.99
    own = np.where(bc == abi.CS_DEMUX_NONE, 0, bc).astype(np.int64)
    pick = lambda arrs: np.stack(arrs)[own, np.arange(n)]
    want2 = pick([o2 for _, o2 in runs])
    want2["flags"] |= np.where(amb, abi.CS_F_AMBIGUOUS, 0).astype(np.uint8)
    assert np.array_equal(g2, want2)
    assert np.array_equal(g1, pick([o1 for o1, _ in runs]))


@pytest.mark.parametrize("length,count,text", [(8, 12, False), (12, 16, False), (8, 12, True)])
def test_demux_on_the_3prime_barcode_of_single_end_reads(length, count, text):
    """Single-end reads of a scheme whose only inline barcode sits at the 3' end: once the 3' adapter is gone the
    barcode ENDS the read -- one SuffixAdapter per barcode in the reference's terms (cutseq/run.py:364-370).  The op
    runs the candidates' own suffix ops (the candidate table walks the read backwards).  Parity: one
    --ensure-inline-barcode run per barcode; ``text``: through the text path, one route per barcode."""
    def scheme3(code):
        return f"ACACGACGCTCTTCCGATCTNNNNNNNN>({code})AGATCGGAAGAGCACACGTC"

    rng = random.Random(length * 7 + count)
    codes = barcode_set(rng, count, length, 4)
    st = planmod.CutadaptConfig()
    st.ensure_inline_barcode = True
    per = 400
    parts = [synth.generate_pairs(per, 150, scheme3(code), seed=length + i, single_end=True, adapter_fraction=0.7)
             for i, code in enumerate(codes)]
    parts.append(synth.generate_pairs(per, 150, scheme3(util.random_dna(rng, length)), seed=99, single_end=True))  # foreign
    order = np.random.default_rng(3).permutation(per * len(parts))
    batch = synth.SynthBatch(np.ascontiguousarray(np.vstack([p.seq1 for p in parts])[order]),
                             np.ascontiguousarray(np.vstack([p.qual1 for p in parts])[order]),
                             np.ascontiguousarray(np.concatenate([p.len1 for p in parts])[order]), None, None, None)
    n = batch.n
    st.demux_barcodes = codes
    tp = planmod.compile_single(BarcodeConfig(scheme3(codes[0])), st)
    assert tp.demux.at_end and not tp.demux.tabulated
    st.demux_barcodes = None
    runs = []
    for code in codes:
        one = planmod.compile_single(BarcodeConfig(scheme3(code)), st)
        (o1, _, _), _ = util.oracle_run(one, batch, threads=8)
        runs.append(o1)
    matched = np.stack([(o["flags"] & abi.CS_F_INLINE) != 0 for o in runs])
    assert 0.3 < float(matched.any(axis=0).mean()) < 0.95  # (reads without read-through have no barcode at their end)
    want_bc = np.where(matched.any(axis=0), matched.argmax(axis=0), abi.CS_DEMUX_NONE).astype(np.uint8)
    if not text:
        bc = np.empty(n, dtype=np.uint8)
        with TrimEngine(tp, device=0, slots=1, max_reads=n, max_stride=batch.stride) as eng:
            g1, _, _ = eng.submit(0, batch.seq1, batch.qual1, batch.len1, bc=bc)
            eng.wait(0)
        amb = (g1["flags"] & abi.CS_F_AMBIGUOUS) != 0
        assert np.array_equal(matched.sum(axis=0) > 1, amb)
        assert np.array_equal(bc[~amb], want_bc[~amb])
        own = np.where(bc == abi.CS_DEMUX_NONE, 0, bc).astype(np.int64)
        want1 = np.stack(runs)[own, np.arange(n)]
        want1["flags"] |= np.where(amb, abi.CS_F_AMBIGUOUS, 0).astype(np.uint8)
        assert np.array_equal(g1, want1)
        return
    from cutseq_amd import textpath
    import hostfmt
    names = [f"SE:{i} 1:N:0:X".encode() for i in range(n)]
    text1 = b"".join(b"@" + names[i] + b"\n" + batch.seq1[i, :batch.len1[i]].tobytes() + b"\n+\n" +
                     batch.qual1[i, :batch.len1[i]].tobytes() + b"\n" for i in range(n))
    amb_any = matched.sum(axis=0) > 1
    want = [b""] * (3 + count)
    want_counts = [0] * (3 + count)
    for i in range(n):
        if amb_any[i]:
            continue
        own = int(want_bc[i]) if want_bc[i] != abi.CS_DEMUX_NONE else 0
        ln = int(batch.len1[i])
        route, rec = hostfmt.format_single(names[i], batch.seq1[i, :ln].tobytes(), batch.qual1[i, :ln].tobytes(), runs[own][i], None, tp)
        want[3 + own if route == 0 else route] += rec
        want_counts[3 + own if route == 0 else route] += 1
    with TrimEngine(tp, device=0, slots=0) as eng:
        with textpath.TextEngine(eng, slots=1, max_text_bytes=len(text1) + 1024, max_records=n, stride=batch.stride, bins=count) as te:
            got, counts = te.run(text1, n)
    assert sum(counts) == n and sum(counts[3:]) > n // 4
    if not amb_any.any():
        assert counts == want_counts
        for route in range(3 + count):
            assert got[route][0] == want[route], route


def test_demux_table_matches_the_oracle_on_every_prefix():
    """The device-built table against the CPU oracle's PrefixAdapter on all 5^0 + ... + 5^(m+k) prefixes."""
    rng = random.Random(3)
    codes = barcode_set(rng, 5, 5, 3)
    op = planmod.DemuxOp(codes, 0.2, abi.CS_F_INLINE)
    op.table = demux.build_table(op, 0)
    seq, lens = demux.all_prefixes(op.m + op.k)
    batch = util.batch_from_reads([(bytes(seq[i, : lens[i]]).decode(), "I" * int(lens[i])) for i in range(len(lens))])
    n_match = np.zeros(len(lens), dtype=int)
    for index, code in enumerate(codes):
        one = planmod.TrimPlan(r1=planmod.MateChain([planmod.prefix(code, 0.2, abi.CS_F_INLINE)]), r2=None, has_umi=False,
                               min_length=0, untrimmed_filter=False)
        (o1, _, _), _ = util.oracle_run(one, batch, threads=8)
        hit = (o1["flags"] & abi.CS_F_INLINE) != 0
        n_match += hit
        mine = (op.table & 0xFF) == index
        assert np.all(hit[mine])  # whatever the table assigns to this barcode, the oracle matches too ...
        assert np.array_equal((op.table[mine] >> 8) & 0xF, o1["start"][mine])  # ... and removes as many bases
    assert np.array_equal((op.table & 0xFF) == abi.CS_DEMUX_NONE, n_match == 0)
    assert np.array_equal((op.table & 0x4000) != 0, n_match > 1)


@pytest.mark.parametrize("switch", ["", "CUTSEQ_TEXT_PATH=0", "CUTSEQ_GPU_DEFLATE=0"])
def test_cli_demultiplexes_like_independent_runs(tmp_path, monkeypatch, switch):
    """cutseq --demux-barcodes through the whole host path (two engines, several chunks): every barcode's
    trimmed files hold exactly what an --ensure-inline-barcode run with that one barcode would have written
    to its trimmed files; the untrimmed files hold the pairs no barcode claims."""
    import gzip
    import json
    import zlib

    from cutseq_amd import run as cli
    import hostfmt

    rng = random.Random(11)
    length, count, n = 8, 16, 40_000
    codes = barcode_set(rng, count, length, 5)
    names = [f"bc{i:02d}" for i in range(count)]
    batch = synth.generate_pairs(n, 150, scheme_with(codes[0]), seed=21)
    plant_barcodes(rng, batch, codes, length)
    names1 = [f"SIM:{i} 1:N:0:X".encode() for i in range(n)]
    names2 = [f"SIM:{i} 2:N:0:X".encode() for i in range(n)]
    in1, in2 = str(tmp_path / "d_R1.fastq.gz"), str(tmp_path / "d_R2.fastq.gz")
    util.write_fastq(in1, names1, batch.seq1, batch.qual1, batch.len1, gz_members=7000)
    util.write_fastq(in2, names2, batch.seq2, batch.qual2, batch.len2, gz_members=9000)
    table = tmp_path / "barcodes.tsv"
    table.write_text("# name\tsequence\n" + "".join(f"{a}\t{b}\n" for a, b in zip(names, codes)))
    monkeypatch.setenv("CUTSEQ_DEVICES", "0,0")
    monkeypatch.setenv("CUTSEQ_CHUNK_READS", "6000")
    if switch:  # (the host parser / formatter; gzip members deflated on the host)
        monkeypatch.setenv(*switch.split("="))
    prefix = str(tmp_path / "dm")
    cli.main(["-a", scheme_with(codes[0]), "--demux-barcodes", str(table), "-O", prefix, "--json-file",
              str(tmp_path / "r.json"), in1, in2])

    def gunzip(path):
        with gzip.open(path, "rb") as fh:
            return fh.read()

    st = planmod.CutadaptConfig()
    st.ensure_inline_barcode = True
    claimed = np.zeros(n, dtype=int)
    total_trimmed = 0
    for name, code in zip(names, codes):
        one = planmod.compile_paired(BarcodeConfig(scheme_with(code)), st)
        (o1, _, _), (o2, _, _) = util.oracle_run(one, batch, threads=8)
        recs = util.format_batch(one, batch, names1, names2, o1, None, o2)
        want1 = b"".join(x[1] for x in recs if x[0] == hostfmt.ROUTE_TRIMMED)
        want2 = b"".join(x[2] for x in recs if x[0] == hostfmt.ROUTE_TRIMMED)
        assert gunzip(f"{prefix}_{name}_trimmed_R1.fastq.gz") == want1, name
        assert gunzip(f"{prefix}_{name}_trimmed_R2.fastq.gz") == want2, name
        claimed += np.array([x[0] == hostfmt.ROUTE_TRIMMED for x in recs])
        total_trimmed += sum(1 for x in recs if x[0] == hostfmt.ROUTE_TRIMMED)
        last = recs
    assert claimed.max() <= 1  # unambiguous barcode set: no pair is trimmed by two runs
    # pairs no run trims and that are not too short: identical (untrimmed) in every run -> the untrimmed files
    unassigned = [x for x in last if x[0] == hostfmt.ROUTE_UNTRIMMED]
    got_untr = gunzip(f"{prefix}_untrimmed_R1.fastq.gz")
    rep = json.loads((tmp_path / "r.json").read_text())
    assert sum(rep["engine"]["demultiplexed"].values()) == total_trimmed == rep["read_counts"]["output"]
    assert got_untr.count(b"\n") // 4 == n - total_trimmed - rep["read_counts"]["filtered"]["too_short"]
    assert zlib.crc32(got_untr) != 0 and len(unassigned) >= got_untr.count(b"\n") // 4


def test_cli_demultiplexes_under_ranks_like_one_process(tmp_path, monkeypatch):
    """--ranks 2 with --demux-barcodes: every rank writes a part of every barcode's files, the parent concatenates them
    in rank order -- same bytes (decompressed) and same report as the one-process run."""
    import gzip
    import json

    from cutseq_amd import run as cli

    rng = random.Random(12)
    length, count, n = 8, 12, 30_000
    codes = barcode_set(rng, count, length, 5)
    names = [f"bc{i:02d}" for i in range(count)]
    batch = synth.generate_pairs(n, 150, scheme_with(codes[0]), seed=22)
    plant_barcodes(rng, batch, codes, length)
    names1 = [f"SIM:{i} 1:N:0:X".encode() for i in range(n)]
    names2 = [f"SIM:{i} 2:N:0:X".encode() for i in range(n)]
    in1, in2 = str(tmp_path / "d_R1.fastq"), str(tmp_path / "d_R2.fastq.gz")
    util.write_fastq(in1, names1, batch.seq1, batch.qual1, batch.len1)
    util.write_fastq(in2, names2, batch.seq2, batch.qual2, batch.len2, gz_members=4000)
    table = tmp_path / "barcodes.tsv"
    table.write_text("".join(f"{a}\t{b}\n" for a, b in zip(names, codes)))
    monkeypatch.setenv("CUTSEQ_CHUNK_READS", "5000")
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    common = ["-a", scheme_with(codes[0]), "--demux-barcodes", str(table)]
    cli.main(common + ["-O", one, "--json-file", str(tmp_path / "one.json"), in1, in2])
    monkeypatch.setenv("CUTSEQ_DEVICES", "0,0")
    cli.main(common + ["-O", two, "--json-file", str(tmp_path / "two.json"), "--ranks", "2", in1, in2])

    def gunzip(path):
        with gzip.open(path, "rb") as fh:
            return fh.read()

    kinds = [f"{name}_trimmed" for name in names] + ["short", "untrimmed"]
    for kind in kinds:
        for mate in (1, 2):
            assert gunzip(f"{two}_{kind}_R{mate}.fastq.gz") == gunzip(f"{one}_{kind}_R{mate}.fastq.gz"), (kind, mate)
    a, b = json.loads((tmp_path / "one.json").read_text()), json.loads((tmp_path / "two.json").read_text())
    assert a["read_counts"] == b["read_counts"] and a["engine"]["demultiplexed"] == b["engine"]["demultiplexed"]
    assert sum(b["engine"]["demultiplexed"].values()) > n // 2 and not list(tmp_path.glob("*.part*"))
