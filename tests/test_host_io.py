"""Host side of the CLI (no GPU): native FASTQ parser / formatter against the Python record logic,
chunking, error behaviour, and the argument handling the reference's main() performs."""
import contextlib
import gzip
import io
import json
from pathlib import Path

import numpy as np
import pytest

from cutseq_amd import abi, fastq, plan as planmod, run as cli
import hostfmt
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig

import util


def write_fq(path, records, gz=True, eol=b"\n", final_newline=True):
    body = b"".join(b"@" + n + eol + s + eol + b"+" + eol + q + eol for n, s, q in records)
    if not final_newline:
        body = body[: -len(eol)]
    if gz:
        with gzip.open(path, "wb") as fh:
            fh.write(body)
    else:
        Path(path).write_bytes(body)
    return path


def test_chunks_roundtrip_fixture(tmp_path):
    rec1 = util.read_fastq_gz(util.GOLDEN / "fixture1k_R1.fq.gz")
    rec2 = util.read_fastq_gz(util.GOLDEN / "fixture1k_R2.fq.gz")
    chunks = list(fastq.read_chunks(str(util.GOLDEN / "fixture1k_R1.fq.gz"), str(util.GOLDEN / "fixture1k_R2.fq.gz"),
                                    chunk_reads=300))
    assert [c.n for c in chunks] == [300, 300, 300, 100]
    i = 0
    for c in chunks:
        assert c.stride == 160 and c.paired
        for j in range(c.n):
            name = c.raw1[c.name_off1[j]: c.name_off1[j] + c.name_len1[j]]
            assert (name, util.row_bytes(c.seq1, c.len1, j), util.row_bytes(c.qual1, c.len1, j)) == rec1[i]
            name2 = c.raw2[c.name_off2[j]: c.name_off2[j] + c.name_len2[j]]
            assert (name2, util.row_bytes(c.seq2, c.len2, j), util.row_bytes(c.qual2, c.len2, j)) == rec2[i]
            i += 1
    assert i == 1000


@pytest.mark.parametrize("eol,final_newline,gz", [(b"\n", True, True), (b"\r\n", True, False), (b"\n", False, False)])
def test_parser_line_endings(tmp_path, eol, final_newline, gz):
    recs = [(b"r1 c", b"ACGTN", b"IIII#"), (b"r2", b"", b""), (b"r3/1", b"GG", b"#I")]
    p = write_fq(tmp_path / "a.fq", recs, gz=gz, eol=eol, final_newline=final_newline)
    (c,) = list(fastq.read_chunks(str(p)))
    assert c.n == 3 and list(c.len1) == [5, 0, 2]
    assert c.raw1[c.name_off1[2]: c.name_off1[2] + c.name_len1[2]] == b"r3/1"


def test_parser_errors(tmp_path):
    p = tmp_path / "bad.fq"
    p.write_bytes(b"@r1\nACGT\n+\nIII\n")  # quality shorter than sequence
    with pytest.raises(fastq.FastqFormatError):
        list(fastq.read_chunks(str(p)))
    p.write_bytes(b"r1\nACGT\n+\nIIII\n")  # no '@'
    with pytest.raises(fastq.FastqFormatError):
        list(fastq.read_chunks(str(p)))
    p.write_bytes(b"@r1\nACGT\n+\nIIII\n@r2\nAC\n")  # truncated last record
    with pytest.raises(fastq.FastqFormatError):
        list(fastq.read_chunks(str(p)))
    a = write_fq(tmp_path / "a.fq", [(b"x", b"A", b"I"), (b"y", b"C", b"I")], gz=False)
    b = write_fq(tmp_path / "b.fq", [(b"x", b"A", b"I")], gz=False)
    with pytest.raises(fastq.FastqFormatError):  # more reads in one file than in the other
        list(fastq.read_chunks(str(a), str(b)))


@pytest.mark.parametrize("preset,flags,paired", [
    ("TAKARAV3", {"trim_polyA": True}, True),
    ("SACSEQV3", {}, False),
    ("INLINE", {"ensure_inline_barcode": True}, True),
    ("TAKARAV3", {"auto_rc": True}, False),
])
def test_native_formatter_equals_python_record_logic(preset, flags, paired):
    """csh_format_chunk == hostfmt.format_pair/format_single, results taken from the oracle."""
    scheme = BUILDIN_ADAPTERS[preset]
    st = planmod.CutadaptConfig()
    for k, v in flags.items():
        setattr(st, k, v)
    tp = util.compile_plan(scheme, st, paired)
    rec1 = util.read_fastq_gz(util.GOLDEN / "fixture1k_R1.fq.gz")
    rec2 = util.read_fastq_gz(util.GOLDEN / "fixture1k_R2.fq.gz")
    rec1[3] = (b"weird/1", rec1[3][1], rec1[3][2])
    rec2[3] = (b"weird/2", rec2[3][1], rec2[3][2])
    rec1[4] = (b"tab\tcomment.1", rec1[4][1], rec1[4][2])
    rec2[4] = (b"tab\tcomment.2", rec2[4][1], rec2[4][2])
    chunks = list(fastq.read_chunks(str(util.GOLDEN / "fixture1k_R1.fq.gz"),
                                    str(util.GOLDEN / "fixture1k_R2.fq.gz") if paired else None))
    (c,) = chunks
    # patch the names in the raw buffers through a rebuilt chunk
    body1 = b"".join(b"@" + n + b"\n" + s + b"\n+\n" + q + b"\n" for n, s, q in rec1)
    body2 = b"".join(b"@" + n + b"\n" + s + b"\n+\n" + q + b"\n" for n, s, q in rec2)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        Path(d, "1.fq").write_bytes(body1)
        Path(d, "2.fq").write_bytes(body2)
        (c,) = list(fastq.read_chunks(str(Path(d, "1.fq")), str(Path(d, "2.fq")) if paired else None))
    batch = util.batch_from_records(rec1, rec2 if paired else None)
    (r1, cap2, _), m2 = util.oracle_run(tp, batch)
    r2 = m2[0] if m2 else None
    # same stride as the chunk (batch_from_records pads to a multiple of 4 as well)
    data, counts = fastq.format_chunk(c, tp, r1, cap2, r2)
    want = util.format_batch(tp, batch, [r[0] for r in rec1], [r[0] for r in rec2], r1, cap2, r2)
    for route in range(3):
        w1 = b"".join(x[1] for x in want if x[0] == route)
        assert data[route][0] == w1
        if paired:
            assert data[route][1] == b"".join(x[2] for x in want if x[0] == route)
        assert counts[route] == sum(1 for x in want if x[0] == route)


def test_mismatched_pair_ids_raise():
    st = planmod.CutadaptConfig()
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        write_fq(Path(d, "1.fq"), [(b"a 1", b"ACGT" * 10, b"I" * 40)], gz=False)
        write_fq(Path(d, "2.fq"), [(b"b 2", b"ACGT" * 10, b"I" * 40)], gz=False)
        (c,) = list(fastq.read_chunks(str(Path(d, "1.fq")), str(Path(d, "2.fq"))))
    res = np.zeros(1, dtype=abi.RESULT_DTYPE)
    with pytest.raises(ValueError, match="Input read IDs not identical"):
        fastq.format_chunk(c, tp, res, None, res.copy())


# ---------------------------------------------------------------- CLI argument handling (reference main())


def parse(argv):
    return cli.resolve_args(cli.build_parser().parse_args(argv))


def test_cli_output_naming_and_preset_resolution():
    a = parse(["-A", "takarav3", "x/sample_R1_001.fastq.gz", "x/sample_R2_001.fastq.gz"])
    assert a.adapter_scheme == BUILDIN_ADAPTERS["TAKARAV3"]
    assert a.output_file == ["x/sample_trimmed_R1.fastq.gz", "x/sample_trimmed_R2.fastq.gz"]
    assert a.short_file == ["x/sample_short_R1.fastq.gz", "x/sample_short_R2.fastq.gz"]
    assert a.untrimmed_file == [None, None]
    a = parse(["-a", "acgt acgt > tttt", "-O", "out/p", "in.fq"])
    assert a.adapter_scheme == "ACGTACGT>TTTT"
    assert a.output_file == ["out/p_trimmed_R1.fastq.gz"] and a.short_file == ["out/p_short_R1.fastq.gz"]
    a = parse(["-A", "INLINE", "--ensure-inline-barcode", "a_R1.fq.gz", "a_R2.fq.gz"])
    assert a.untrimmed_file == ["a_untrimmed_R1.fastq.gz", "a_untrimmed_R2.fastq.gz"]
    a = parse(["-A", "TAKARAV3", "--ensure-inline-barcode", "a_R1.fq.gz", "a_R2.fq.gz"])
    assert a.untrimmed_file == [None, None]  # scheme has no inline barcode
    a = parse(["-A", "AAAA>CCCC", "r.fq"])  # unknown preset name is reused as the scheme
    assert a.adapter_scheme == "AAAA>CCCC"
    a = parse(["-A", "TAKARAV3", "-a", "AAAA>CCCC", "r.fq"])  # explicit scheme wins
    assert a.adapter_scheme == "AAAA>CCCC"


def test_cli_error_exits():
    for argv in (["a", "b", "c", "-A", "TAKARAV3"], ["r.fq"], ["-A", "TAKARAV3", "-o", "only_one.fq", "a.fq", "b.fq"],
                 ["-A", "TAKARAV3"]):
        with pytest.raises(SystemExit) as e:
            parse(argv)
        assert e.value.code == 1, argv
    with pytest.raises(SystemExit) as e:  # invalid scheme -> BarcodeConfig exits 1
        cli.main(["-a", "AAAAXN>CCCC", "-n", "r.fq"])
    assert e.value.code == 1
    with pytest.raises(SystemExit) as e:
        cli.main([])
    assert e.value.code == 0
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), pytest.raises(SystemExit) as e:
        cli.main(["--list-adapters"])
    assert e.value.code == 0 and "TAKARAV3" in buf.getvalue()


def test_cli_dry_run_lists_the_chain(capsys):
    cli.main(["-A", "TAKARAV3", "--trim-polyA", "-n", "r_R1.fq.gz"])
    out = capsys.readouterr().out
    assert "Step 1: SuffixRemover('.1')" in out and "RightmostFrontAdapter" in out
    assert "QualityTrimmer(cutoff_front=0, cutoff_back=20" in out and "Renamer" in out
    cli.main(["-A", "TAKARAV3", "-n", "r_R1.fq.gz", "r_R2.fq.gz"])
    out = capsys.readouterr().out
    assert "p5: ACACGACGCTCTTCCGATCT (AGATCGGAAGAGCGTCGTGT)" in out and "strand: -" in out


def _step_kinds(tp):
    """The kind of every --dry-run step (per mate for paired plans), in the order they are printed."""
    import re as _re
    names = ("SuffixRemover", "RightmostFrontAdapter", "BackAdapter", "PrefixAdapter", "SuffixAdapter",
             "NonInternalBackAdapter", "NonInternalFrontAdapter", "UnconditionalCutter", "ConditionalCutter",
             "PairedEndRenamer", "Renamer", "QualityTrimmer", "ReverseComplementConverter")
    pat = _re.compile(r"\b(" + "|".join(names) + r")\(")
    return [tuple(pat.findall(step)) for step in cli.dry_run_steps(tp)]


def test_dry_run_prints_the_renamer_where_the_reference_has_it():
    """The reference's modifier list holds the (PairedEnd)Renamer right behind the UMI step and in front of the mask
    steps (cutseq/run.py:377-380 single-end, 642-645 paired), and --dry-run prints that list (run.py:429-432,
    747-748).  TAKARAV3 (UMI + masks), INLINE (inline barcode + two UMIs) and TAKARAV2 (no UMI)."""
    st = planmod.CutadaptConfig()
    S, R5, B3 = ("SuffixRemover",), ("RightmostFrontAdapter",), ("BackAdapter",)
    U, C, Q = ("UnconditionalCutter",), ("ConditionalCutter",), ("QualityTrimmer",)
    pair = lambda a, b: a + b
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)   # XXX<XXXXXXNNNNNNNN: umi3 8, mask5 3, mask3 6
    assert _step_kinds(tp) == [pair(S, S), pair(S, S), pair(R5, R5), pair(B3, B3), pair(C, U), ("PairedEndRenamer",),
                               pair(U, C), pair(C, U), pair(Q, Q)]  # the renamer is step 6 of 9
    tp = util.compile_plan(BUILDIN_ADAPTERS["INLINE"], st, True)     # NNNNN>NNNNN(ATCACG): inline3, umi5 5, umi3 5
    assert _step_kinds(tp) == [pair(S, S), pair(S, S), pair(R5, R5), pair(B3, B3), U + ("PrefixAdapter",), pair(U, C),
                               pair(C, U), ("PairedEndRenamer",), pair(Q, Q)]
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV2"], st, True)   # XXX<XXX: no UMI -> the renamer follows step 3
    assert _step_kinds(tp) == [pair(S, S), pair(S, S), pair(R5, R5), pair(B3, B3), ("PairedEndRenamer",), pair(U, C),
                               pair(C, U), pair(Q, Q)]
    assert "'{id}'" in cli.dry_run_steps(tp)[4] and "{r1.cut_prefix}" not in cli.dry_run_steps(tp)[4]
    # single-end: all cuts unconditional, Renamer in front of the masks, the reverse complement last (run.py:373-386, 420-426)
    st.auto_rc = True
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, False)
    assert _step_kinds(tp) == [S, S, R5, B3, U, ("Renamer",), U, U, Q, ("ReverseComplementConverter",)]
    tp = util.compile_plan(BUILDIN_ADAPTERS["INLINE"], st, False)    # strand '+': --auto-rc is ignored
    assert _step_kinds(tp) == [S, S, R5, B3, ("SuffixAdapter",), U, U, ("Renamer",), Q]
    tp = util.compile_plan(BUILDIN_ADAPTERS["SMALLRNA"], st, False)
    assert _step_kinds(tp) == [S, S, R5, B3, ("Renamer",), Q]


def test_one_version_source(capsys):
    """`cutseq -V`, the package attribute and the packaging metadata name ONE version."""
    import cutseq_amd
    with pytest.raises(SystemExit):
        cli.main(["-V"])
    assert capsys.readouterr().out.strip() == f"cutseq {cutseq_amd.__version__}"
    text = (util.GOLDEN.parent.parent / "pyproject.toml").read_text()
    assert 'dynamic = ["version"]' in text and 'attr = "cutseq_amd.__version__"' in text
    assert "\nversion = \"" not in text.split("[tool.")[0]


# ---------------------------------------------------------------- reports (run.py:222-302, 489, 810)


def test_reports_from_oracle_results():
    """TSV + JSON reports assembled from per-read results; counters cross-checked against a direct
    recount over the formatted records (results come from the oracle, no GPU involved)."""
    from cutseq_amd import report

    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    rec1 = util.read_fastq_gz(util.GOLDEN / "fixture1k_R1.fq.gz")
    rec2 = util.read_fastq_gz(util.GOLDEN / "fixture1k_R2.fq.gz")
    batch = util.batch_from_records(rec1, rec2)
    (r1, cap2, s1), (r2, _, s2) = util.oracle_run(tp, batch)
    totals = report.new_totals()
    half = 500  # two chunks, like the streaming loop
    for lo in (0, half):
        sl = slice(lo, lo + half)
        report.account_chunk(totals, tp, batch.len1[sl], r1[sl], batch.len2[sl], r2[sl])
    want = util.format_batch(tp, batch, [r[0] for r in rec1], [r[0] for r in rec2], r1, cap2, r2)
    for route in range(3):
        totals["routes"][route] = sum(1 for x in want if x[0] == route)
    totals["stats"] = [[s1.as_dict(), s2.as_dict()]]
    totals["devices"], totals["seconds"] = [0], 0.0

    def written_bp(mate):
        return sum(len(x[mate].split(b"\n")[1]) for x in want if x[0] == 0)

    head, vals = report.minimal_report(tp, totals).split("\n")
    row = dict(zip(head.split("\t"), vals.split("\t")))
    assert head.split("\t") == ["status", "in_reads", "in_bp", "too_short", "too_long", "too_many_n", "out_reads",
                                "w/adapters", "qualtrim_bp", "out_bp", "w/adapters2", "qualtrim2_bp", "out2_bp"]
    assert row["status"] == "OK" and int(row["in_reads"]) == 1000
    assert int(row["in_bp"]) == int(batch.len1.sum()) + int(batch.len2.sum())
    assert int(row["too_short"]) + int(row["out_reads"]) == 1000
    assert int(row["out_bp"]) == written_bp(1) and int(row["out2_bp"]) == written_bp(2)
    # first AdapterCutter of each mate only (run.py:59-73): R1's chain starts with the 5' adapter
    slot1, op1 = report.first_adapter(tp.r1)
    assert op1.kind_name == "RightmostFrontAdapter"
    assert int(row["w/adapters"]) == int(s1.op_matched[slot1])
    assert int(row["w/adapters"]) == int(np.count_nonzero(r1["flags"] & abi.CS_F_ADAPTER5))

    bc = BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"])
    rep = report.json_report(tp, totals, bc, "a_R1.fq.gz", "a_R2.fq.gz", "o1", "o2", "s1", "s2", None, None)
    json.dumps(rep)  # serialisable
    assert rep["tag"] == "Cutadapt report" and rep["input"]["paired"] is True
    assert rep["barcode"] == bc.to_dict()
    rc, bp = rep["read_counts"], rep["basepair_counts"]
    assert rc["input"] == 1000 and rc["output"] + rc["filtered"]["too_short"] == 1000
    assert set(rc["filtered"]) == set(report.FILTER_KEYS)
    assert bp["output"] == written_bp(1) + written_bp(2) and bp["input"] == int(row["in_bp"])
    assert bp["quality_trimmed"] == int(s1.qualtrim_bp) + int(s2.qualtrim_bp)
    (a,) = rep["adapters_read1"]
    assert a["five_prime_end"]["sequence"] == op1.sequence and a["three_prime_end"] is None
    assert a["five_prime_end"]["trimmed_lengths"] == []
    # cutadapt's ErrorRanges for a 20-nt adapter at rate 0.2: one error from 5 nt, two from 10, ... (guide: "error tolerance")
    assert a["five_prime_end"]["error_lengths"] == [4, 9, 14, 19, 20]  # "the last number is always the adapter length"
    assert rep["schema_version"] == [0, 3] and rep["cutadapt_version"].startswith("5.0+")

    def no_none_under(node, path=""):
        """every None left in the report is one cutadapt itself writes for an unused feature"""
        allowed = ("filtered.", "reverse_complemented", "poly_a_trimmed", "on_reverse_complement", "three_prime_end",
                   "adjacent_bases", "dominant_adjacent_base", "untrimmed", "is_untrimmed_any")
        if isinstance(node, dict):
            for k, v in node.items():
                no_none_under(v, f"{path}.{k}")
        elif isinstance(node, list):
            for v in node:
                no_none_under(v, path)
        else:
            assert node is not None or any(a in path for a in allowed), path

    no_none_under(rep)


# ---------------------------------------------------------------- threaded reader / worker finisher


def test_reader_restrides_when_a_later_chunk_is_longer(tmp_path):
    """The row stride is shared by the two parser threads and only grows; a half parsed with the
    smaller stride is parsed again, so both mates of a chunk always agree."""
    recs1 = [(b"r%d" % i, b"A" * 10, b"I" * 10) for i in range(8)]
    recs2 = [(b"r%d" % i, b"C" * (10 if i < 6 else 70), b"I" * (10 if i < 6 else 70)) for i in range(8)]
    p1 = write_fq(tmp_path / "1.fq", recs1, gz=False)
    p2 = write_fq(tmp_path / "2.fq.gz", recs2, gz=True)
    chunks = list(fastq.read_chunks(str(p1), str(p2), chunk_reads=3))
    assert [c.n for c in chunks] == [3, 3, 2]
    for c in chunks:
        assert c.seq1.shape == c.seq2.shape == (c.n, c.stride) and c.stride % 4 == 0
    assert chunks[-1].stride == 72 and util.row_bytes(chunks[-1].seq2, chunks[-1].len2, 1) == b"C" * 70
    assert util.row_bytes(chunks[-1].seq1, chunks[-1].len1, 1) == b"A" * 10
    for c in chunks:
        c.release()
    # released buffers come back from the arena: same storage for the next chunk of the same shape
    again = list(fastq.read_chunks(str(p1), str(p2), chunk_reads=3))
    assert [c.n for c in again] == [3, 3, 2]


def test_finish_chunk_and_ordered_writer(tmp_path):
    """finish_chunk (format + one gzip member per stream) through OutputFile.write_job gives the
    same decompressed text as the plain formatter, in chunk order, for gz and plain outputs."""
    from concurrent.futures import ThreadPoolExecutor

    st = planmod.CutadaptConfig()
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    r1p, r2p = str(util.GOLDEN / "fixture1k_R1.fq.gz"), str(util.GOLDEN / "fixture1k_R2.fq.gz")
    want1, want2, jobs = b"", b"", []
    outs = [fastq.OutputFile(str(tmp_path / "o1.fq.gz")), fastq.OutputFile(str(tmp_path / "o2.fq"))]
    gz = [[True, False], [None, None], [None, None]]
    with ThreadPoolExecutor(4) as pool:
        for c in fastq.read_chunks(r1p, r2p, chunk_reads=128):
            batch = SynthLike(c)
            (r1, cap2, _), (r2, _, _) = util.oracle_run(tp, batch)
            data, _ = fastq.format_chunk(c, tp, r1, cap2, r2)
            want1 += data[0][0]
            want2 += data[0][1]
            fut = pool.submit(fastq.finish_chunk, c, tp, r1, cap2, r2, gz)
            outs[0].write_job(fut, 0, 0)
            outs[1].write_job(fut, 0, 1)
            jobs.append(fut)
        for o in outs:
            o.close()
    assert gzip.decompress((tmp_path / "o1.fq.gz").read_bytes()) == want1
    assert (tmp_path / "o2.fq").read_bytes() == want2
    assert sum(f.result()[1][0] + f.result()[1][1] for f in jobs) == 1000


class SynthLike:
    """The arrays of a fastq.Chunk under the attribute names util.oracle_run expects."""

    def __init__(self, c):
        self.n, self.stride = c.n, c.stride
        self.seq1, self.qual1, self.len1 = c.seq1, c.qual1, c.len1
        self.seq2, self.qual2, self.len2 = c.seq2, c.qual2, c.len2


# ---------------------------------------------------------------- gzip codec (libdeflate / BGZF / zlib)


def _bgzf_block(data: bytes) -> bytes:
    import struct
    import zlib
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    body = c.compress(data) + c.flush()
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body +
            struct.pack("<II", zlib.crc32(data), len(data)))


def _codec_files(tmp_path):
    import random
    from cutseq_amd import codec
    rng = random.Random(4)
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGT") for _ in range(60)), b"I" * 60) for i in range(20000))
    files = {
        "single": gzip.compress(text, 1),
        "members": b"".join(codec.gzip_member(text[i:i + 300_000], 1) for i in range(0, len(text), 300_000)),
        "bgzf": b"".join(_bgzf_block(text[i:i + 65280]) for i in range(0, len(text), 65280)) + _bgzf_block(b""),
        "padded": gzip.compress(text, 1) + b"\0" * 4096,
        "bgzf_then_plain_member": _bgzf_block(text[:50000]) + gzip.compress(text[50000:], 1),
    }
    paths = {}
    for name, blob in files.items():
        paths[name] = tmp_path / f"{name}.gz"
        paths[name].write_bytes(blob)
    return text, paths


def _inflate(path, pool=None):
    from cutseq_amd import codec
    src = codec.GzipSource(str(path), pool)
    try:
        return b"".join(bytes(memoryview(arr)[:n]) for arr, n in src.blocks())
    finally:
        src.close()


def test_codec_every_container_and_both_backends(tmp_path, monkeypatch):
    """single member, this tool's own multi-member output, BGZF (parallel block inflate), zero padding, mixed
    containers -- through libdeflate and through the zlib fallback; compressor output readable by gzip."""
    from concurrent.futures import ThreadPoolExecutor
    from cutseq_amd import codec
    text, paths = _codec_files(tmp_path)
    with ThreadPoolExecutor(3) as pool:
        for name, path in paths.items():
            assert _inflate(path) == text, name
            assert _inflate(path, pool) == text, name
    assert gzip.decompress(codec.gzip_member(text, 1)) == text
    assert gzip.decompress(codec.gzip_member(b"", 1)) == b""
    if codec.libdeflate() is not None:
        monkeypatch.setattr(codec, "_lib", None)  # zlib only (a box without libdeflate.so)
        monkeypatch.setattr(codec, "_lib_tried", True)
        for name, path in paths.items():
            assert _inflate(path) == text, name
        assert gzip.decompress(codec.gzip_member(text, 1)) == text


def test_codec_truncated_and_corrupt_input(tmp_path):
    text, paths = _codec_files(tmp_path)
    cut = tmp_path / "cut.gz"
    cut.write_bytes(paths["single"].read_bytes()[:-200])
    with pytest.raises(OSError):
        _inflate(cut)
    bad = bytearray(paths["members"].read_bytes())
    bad[len(bad) // 2] ^= 0xFF
    broken = tmp_path / "bad.gz"
    broken.write_bytes(bytes(bad))
    with pytest.raises(OSError):
        _inflate(broken)
    empty = tmp_path / "empty.gz"
    empty.write_bytes(b"")
    assert _inflate(empty) == b""


def test_codec_members_inflate_ahead_of_their_boundaries(tmp_path):
    """With a pool, members are tried at every gzip magic behind the current one before their true starts are
    known.  Payloads full of false magics (stored blocks keep them verbatim), empty members, zero padding, a
    corrupt member and a truncated tail must come out exactly as from the serial walk."""
    import os
    import random
    from concurrent.futures import ThreadPoolExecutor
    from cutseq_amd import codec
    rng = random.Random(11)
    parts = []
    for i in range(30):
        noise = os.urandom(rng.randint(0, 120_000))  # incompressible: stored verbatim, magics and all
        parts.append(noise + b"\x1f\x8b\x08\x00" * rng.randint(0, 6) + b"ACGT" * rng.randint(0, 20_000))
    parts[7] = b""
    blob = b"".join(gzip.compress(x, 1) for x in parts)
    want = b"".join(parts)
    good = tmp_path / "speculative.gz"
    good.write_bytes(blob + b"\0" * 8192)
    with ThreadPoolExecutor(6) as pool:
        assert _inflate(good) == want
        for _ in range(3):
            assert _inflate(good, pool) == want
        bad = bytearray(blob)
        bad[len(bad) * 2 // 3] ^= 0x55
        broken = tmp_path / "speculative_bad.gz"
        broken.write_bytes(bytes(bad))
        with pytest.raises(OSError):
            _inflate(broken, pool)
        cut = tmp_path / "speculative_cut.gz"
        cut.write_bytes(blob[:-37])
        with pytest.raises(OSError):
            _inflate(cut, pool)


def test_reader_takes_bgzf_and_multi_member_input(tmp_path):
    """read_chunks on BGZF / multi-member files yields the same records as on the plain text."""
    text, paths = _codec_files(tmp_path)
    plain = tmp_path / "plain.fq"
    plain.write_bytes(text)
    def records(path):
        out = []
        for ch in fastq.read_chunks(str(path), chunk_reads=3000):
            for i in range(ch.n):
                out.append((bytes(ch.raw1[ch.name_off1[i]: ch.name_off1[i] + ch.name_len1[i]]),
                            bytes(ch.seq1[i, : ch.len1[i]])))
            ch.release()
        return out
    want = records(plain)
    assert len(want) == 20000
    for name in ("bgzf", "members", "bgzf_then_plain_member"):
        assert records(paths[name]) == want, name


def test_library_binding_is_thread_safe():
    """The reader threads of both mates reach the lazily bound helper libraries at the same moment.  In a fresh
    interpreter, eight threads released together must all get library objects whose functions carry their
    prototypes (an unbound csh_fastq_count takes its 64-bit buffer address as a C int: SIGSEGV), and there must
    be one library object and one worker pool per process."""
    import subprocess
    import sys
    code = r'''
import sys, threading
sys.setswitchinterval(1e-6)
from cutseq_amd import codec, fastq, synth
start = threading.Barrier(8)
seen, bad = [], []
def worker():
    start.wait()
    L = fastq._lib()
    if L.csh_fastq_count.argtypes is None or L.csh_fastq_parse.argtypes is None or L.csh_format_chunk.argtypes is None:
        bad.append("helper library without prototypes")
    D = codec.libdeflate()
    if D is not None and D.libdeflate_gzip_decompress_ex.argtypes is None:
        bad.append("libdeflate without prototypes")
    seen.append((id(L), id(synth.host_lib()), id(fastq._pool())))
threads = [threading.Thread(target=worker) for _ in range(8)]
[t.start() for t in threads]
[t.join() for t in threads]
assert not bad, bad
assert len(set(seen)) == 1, seen
print("ok")
'''
    root = str(Path(__file__).resolve().parent.parent)
    for _ in range(4):
        out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and out.stdout.strip() == "ok", (out.returncode, out.stdout, out.stderr[-500:])


def test_threads_flag_bounds_the_host_pool(monkeypatch):
    """-t/--threads (reference: make_runner(cores=N), cutseq/run.py:436, 753, 998-1003) caps the host thread pool;
    without it every usable core is used."""
    monkeypatch.setattr(fastq, "_POOL", None)
    monkeypatch.setattr(fastq, "_THREADS", None)
    everything = fastq.pool_size()
    assert everything >= 2
    fastq.set_threads(1)
    assert fastq.pool_size() == 1
    fastq.set_threads(3)
    assert fastq.pool_size() == min(3, everything)
    pool = fastq._pool()
    assert pool._max_workers == min(3, everything)
    with pytest.raises(RuntimeError):
        fastq.set_threads(5)  # the pool is running: its size is fixed
    pool.shutdown()
    args = cli.build_parser().parse_args(["-A", "TAKARAV3", "-t", "4", "x_R1.fq.gz"])
    assert args.threads == 4 and cli.build_parser().parse_args(["-A", "TAKARAV3", "x.fq"]).threads is None


def test_one_huge_gzip_member_is_streamed_not_reinflated(tmp_path, monkeypatch):
    """A member beyond the one-call cap is tried once at the cap and then streamed through zlib: same
    bytes, and the arena keeps no giant buffers."""
    from cutseq_amd import codec
    body = b"".join(b"@r%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(60_000))
    path = tmp_path / "big.fq.gz"
    path.write_bytes(gzip.compress(body, 1))
    monkeypatch.setattr(codec, "_MEMBER_CAP", 1 << 20)  # the ~1.7 MB member no longer fits one call
    takes = []

    def take(n):
        takes.append(n)
        return np.empty(n, dtype=np.uint8)

    src = codec.GzipSource(str(path), fastq._pool(), take, lambda a: None)
    got = b"".join(bytes(memoryview(b)[:n]) for b, n in src.blocks())
    src.close()
    assert got == body
    assert max(takes) <= 32 << 20 and sum(1 for t in takes if t >= (1 << 20)) <= 4  # no escalation ladder
    big = np.empty(fastq._Arena._KEEP_MAX + 1, dtype=np.uint8)
    fastq.ARENA.give(big)
    assert big.size not in fastq.ARENA._free or not any(a is big for a in fastq.ARENA._free[big.size])


def test_one_huge_gzip_member_decodes_in_parallel(tmp_path, monkeypatch):
    """csrc/pinflate.c + codec.GzipSource._parallel_member: the compressed bytes of ONE member are cut into chunks, every
    chunk is decoded from the block boundary found behind its start (references into the unknown 32 KB in front as
    markers), accepted only when the chain of proven bit positions arrives exactly there, and resolved afterwards.
    Same bytes as gzip.decompress for every compressor setting, with members behind the big one, with header fields;
    buffers all come back; damage is noticed."""
    import os
    import random
    import zlib
    from cutseq_amd import codec
    rng = random.Random(9)
    body = "".join(f"@SIM:{i} 1:N:0:X\n{''.join(rng.choice('ACGT') for _ in range(100))}\n+\n"
                   f"{''.join(rng.choice('FFFFFF:,#') for _ in range(100))}\n" for i in range(60_000)).encode()  # ~14 MB
    monkeypatch.setattr(codec, "_MEMBER_CAP", 1 << 20)
    pool = fastq._pool()
    live = {}

    def take(n):
        a = np.empty(n, dtype=np.uint8)
        live[id(a)] = a
        return a

    strict = [True]

    def give(a):
        known = live.pop(id(a), None) is not None
        assert known or not strict[0], "a buffer was given back twice (or never taken)"

    def read(blob, expect_parallel=True):
        path = tmp_path / "big.fq.gz"
        path.write_bytes(blob)
        src = codec.GzipSource(str(path), pool, take, give)
        out = bytearray()
        for arr, n in src.blocks():
            out += memoryview(arr)[:n]
            give(arr)
        stats = dict(src.stats)
        src.close()
        assert not live, f"{len(live)} buffers never came back"
        if expect_parallel:
            assert stats.get("chunks", 0) >= 3 and stats.get("serial", 0) <= 1, stats
        return bytes(out)

    for level in (1, 6, 9):
        assert read(gzip.compress(body, level)) == body
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):  # fixed-code blocks are not what the finder looks for
        c = zlib.compressobj(6, zlib.DEFLATED, 31, 8, strategy)
        assert read(c.compress(body) + c.flush(), expect_parallel=False) == body
    # what pigz writes (and this repo's tier inputs): ONE member made of pieces joined by sync flushes -- an empty stored
    # block every 128 KB of text.  A chunk that ends in front of such a block goes on through it: the chunk behind starts
    # at the dynamic block BEHIND the flush (round 5: a third of a pigz file's chunks fell back to the serial decoder)
    def pieces(nbytes):
        parts = []
        for lo in range(0, len(body), nbytes):
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            parts.append(c.compress(body[lo:lo + nbytes]) + c.flush(zlib.Z_FINISH if lo + nbytes >= len(body) else zlib.Z_SYNC_FLUSH))
        return (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x04\xff" + b"".join(parts) + zlib.crc32(body).to_bytes(4, "little")
                + (len(body) & 0xffffffff).to_bytes(4, "little"))
    for nbytes in (128 << 10, 40_000, 1 << 20):
        assert read(pieces(nbytes)) == body
    tail = b"@tail\nACGT\n+\nIIII\n"
    assert read(gzip.compress(body, 1) + gzip.compress(tail, 6)) == body + tail  # members behind the big one
    named = b"\x1f\x8b\x08\x08\0\0\0\0\0\xffreads.fq\0" + gzip.compress(body, 1)[10:]
    assert read(named) == body
    noise = os.urandom(3 << 20)  # stored blocks only: every chunk is decoded from the proven position, serially
    assert read(gzip.compress(noise, 1), expect_parallel=False) == noise
    zeros = bytes(40 << 20)  # ratio 1000: the symbol buffers have to grow
    assert read(gzip.compress(zeros, 6), expect_parallel=False) == zeros
    monkeypatch.setenv("CUTSEQ_PARALLEL_INFLATE", "0")  # the zlib stream is still there (its blocks are its own arrays)
    strict[0] = False
    assert read(gzip.compress(body, 1), expect_parallel=False) == body
    strict[0] = True
    monkeypatch.delenv("CUTSEQ_PARALLEL_INFLATE")
    for damage in (lambda b: b[:-6] + bytes([b[-6] ^ 0x55]) + b[-5:],          # CRC-32 in the trailer
                   lambda b: b[:len(b) // 2] + bytes([b[len(b) // 2] ^ 0xFF]) + b[len(b) // 2 + 1:],  # a byte in the middle
                   lambda b: b[:len(b) // 2]):                                   # truncated
        with pytest.raises(OSError):
            read(damage(gzip.compress(body, 1)), expect_parallel=False)
        live.clear()  # (what an aborted read still holds goes with the garbage collector)


def test_damaged_huge_member_never_passes_silently(tmp_path, monkeypatch):
    """Random damage to a single-member file (bit flips, truncation, overwritten and deleted stretches) through the
    parallel decoder: every case ends in OSError (or, for damage the format cannot see, in the right bytes) -- never in
    wrong text, never in a hang."""
    import os
    import random
    from cutseq_amd import codec
    rng = random.Random(11)
    body = "".join(f"@SIM:{i} 1:N:0:X\n{''.join(rng.choice('ACGT') for _ in range(100))}\n+\n"
                   f"{''.join(rng.choice('FFFFFF:,#') for _ in range(100))}\n" for i in range(30_000)).encode()
    good = gzip.compress(body, 1)
    monkeypatch.setattr(codec, "_MEMBER_CAP", 1 << 20)
    path = tmp_path / "damaged.fq.gz"
    errors = 0
    for it in range(60):
        blob = bytearray(good)
        kind = it % 4
        if kind == 0:
            for _ in range(rng.randrange(1, 4)):
                blob[rng.randrange(len(blob))] ^= 1 << rng.randrange(8)
        elif kind == 1:
            blob = blob[:rng.randrange(20, len(blob))]
        elif kind == 2:
            a = rng.randrange(len(blob))
            b = min(len(blob), a + rng.randrange(1, 5000))
            blob[a:b] = os.urandom(b - a)
        else:
            a = rng.randrange(len(blob))
            del blob[a:a + rng.randrange(1, 2000)]
        path.write_bytes(bytes(blob))
        src = codec.GzipSource(str(path), fastq._pool())
        try:
            got = b"".join(bytes(memoryview(a)[:n]) for a, n in src.blocks())
            assert got == body, f"damage of kind {kind} (iteration {it}) went through with different text"
        except OSError:
            errors += 1
        finally:
            src.close()
    assert errors >= 55


def test_barcode_names_must_be_usable_in_file_names(tmp_path):
    from cutseq_amd import demux
    good = tmp_path / "ok.tsv"
    good.write_text("a1\tACGTAC\nb2\tTTGACC\n")
    assert demux.read_barcode_file(str(good)) == (["a1", "b2"], ["ACGTAC", "TTGACC"])
    for bad in ("../x\tACGTAC\n", "a/b\tACGTAC\n", "..\tACGTAC\n"):
        p = tmp_path / "bad.tsv"
        p.write_text(bad)
        with pytest.raises(ValueError):
            demux.read_barcode_file(str(p))


def test_replayed_counters_need_matching_kernel_sources(tmp_path):
    """bench.py replays profiles/*_pmc_summary.json only when it was collected on the kernels it runs."""
    import bench
    from cutseq_amd.build import kernel_source_hash
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps({"pairs_per_launch": 1000, "kernel_source_sha256": "0" * 64, "hbm_bytes_per_launch": 1}))
    got, why = bench.replayed_counters(str(f), 1000)
    assert got is None and "other kernel sources" in why
    f.write_text(json.dumps({"pairs_per_launch": 1000, "kernel_source_sha256": kernel_source_hash(), "hbm_bytes_per_launch": 7}))
    got, why = bench.replayed_counters(str(f), 1000)
    assert why is None and got["hbm_bytes_per_launch"] == 7
    got, why = bench.replayed_counters(str(f), 2000)
    assert got is None and "pairs per launch" in why
    assert bench.replayed_counters(str(tmp_path / "nope.json"), 1)[0] is None


# ---------------------------------------------------------------- text path, host side (reader / writer; no GPU)


@pytest.fixture
def unpinned(monkeypatch):
    """The text path keeps its blocks in page-locked memory; without a GPU an ordinary arena stands in."""
    monkeypatch.setattr(fastq, "PINNED", fastq._Arena())


def _drain(reader):
    got, sizes = b"", []
    while True:
        b = reader.get()
        if b is None:
            break
        got += bytes(memoryview(b.buf)[: b.nbytes])
        sizes.append(b.n)
        b.release()
    reader.close()
    return got, sizes


@pytest.mark.parametrize("container", ["plain", "plain-no-final-newline", "crlf+blank-tail", "gz-members", "gz-one-member"])
def test_text_reader_cuts_blocks_of_whole_records(tmp_path, unpinned, container):
    """textio.TextReader: exactly chunk_reads records per block (the last one shorter), bytes untouched, for every
    container the CLI reads (reference: dnaio through runner.run, cutseq/run.py:434-441)."""
    from cutseq_amd import textio
    recs = [b"@r%d x\n%s\n+\n%s\n" % (i, b"ACGT" * (5 + i % 40), b"IIII" * (5 + i % 40)) for i in range(25_001)]
    body = b"".join(recs)
    path = tmp_path / ("in.fq.gz" if container.startswith("gz") else "in.fq")
    data = body
    if container == "plain-no-final-newline":
        data = body[:-1]
    elif container == "crlf+blank-tail":
        data = body.replace(b"\n", b"\r\n") + b"\r\n\n"
    if container == "gz-members":
        with open(path, "wb") as fh:
            for lo in range(0, len(recs), 3000):
                fh.write(gzip.compress(b"".join(recs[lo:lo + 3000]), 1))
    elif container == "gz-one-member":
        path.write_bytes(gzip.compress(body, 1))
    else:
        path.write_bytes(data)
    got, sizes = _drain(textio.TextReader(str(path), 7000))
    assert sizes == [7000, 7000, 7000, 4001]
    assert got == data.rstrip(b"\r\n")  # (only the line end of the very last record may go)
    # one record too few lines: the reader says so instead of handing a ragged block on
    bad = tmp_path / "bad.fq"
    bad.write_bytes(body + b"@x\nACGT\n")
    with pytest.raises(fastq.FastqFormatError):
        _drain(textio.TextReader(str(bad), 7000))
    empty = tmp_path / "empty.fq"
    empty.write_bytes(b"")
    assert _drain(textio.TextReader(str(empty), 7000)) == (b"", [])


@pytest.mark.parametrize("gz", [False, True])
@pytest.mark.parametrize("tail,blocks", [
    (b"@e\n\n+\n\n", b"@e\n\n+\n\n"),            # a zero-length final record (already-trimmed inputs hold them)
    (b"@e\n\n+\n", b"@e\n\n+\n\n"),               # ... whose empty quality line has no line end: one is supplied
    (b"@e\n\n+\n\n\n\n", b"@e\n\n+\n\n"),         # ... with blank lines behind it: only whole blank lines go
    (b"@e\r\n\r\n+\r\n\r\n", b"@e\r\n\r\n+\r\n\r\n"),
    (b"@e\nAC\n+\nII\n\n \n", b"@e\nAC\n+\nII"),    # an ordinary last record with a blank tail
])
def test_text_reader_zero_length_final_record(tmp_path, unpinned, gz, tail, blocks):
    """the end-of-input branch stripped ALL trailing white space before counting lines, so a file whose
    last record is empty ("@id\\n\\n+\\n\\n", which dnaio reads and trimmed files contain) lost its quality line and
    the run died with 'truncated FASTQ record'.  The last non-blank line decides now."""
    from cutseq_amd import textio
    head = b"".join(b"@r%d\nACGT\n+\nIIII\n" % i for i in range(10))
    path = tmp_path / ("in.fq.gz" if gz else "in.fq")
    path.write_bytes(gzip.compress(head + tail, 1) if gz else head + tail)
    got, sizes = _drain(textio.TextReader(str(path), 7000))
    assert sizes == [11]
    assert got == head + blocks
    # the same record in the MIDDLE of a file (blocks there are cut by newline count) and a '+' line with nothing behind it
    path.write_bytes(gzip.compress(head + b"@e\n\n+\n\n" + head, 1) if gz else head + b"@e\n\n+\n\n" + head)
    got, sizes = _drain(textio.TextReader(str(path), 7000))
    assert sizes == [21] and got == (head + b"@e\n\n+\n\n" + head)[:-1]
    path.write_bytes(gzip.compress(head + b"@e\n\n+", 1) if gz else head + b"@e\n\n+")
    with pytest.raises(fastq.FastqFormatError):
        _drain(textio.TextReader(str(path), 7000))


def test_fasta_input_is_reshaped_into_four_line_records(tmp_path, unpinned):
    """FASTA input (the reference takes whatever dnaio detects, cutseq/run.py:437-441, 754-758): names as they are,
    sequence lines joined, white space at line ends stripped, blank lines and '#' comments skipped, a quality line no
    cutoff trims -- in blocks of whole records, whatever the container; anything else in front of the first '>' is an
    error with its line number."""
    import bz2
    import lzma
    from cutseq_amd import textio
    recs = [(b"r%d some comment" % i, b"ACGTNacgt" * (1 + i % 30)) for i in range(30_001)]
    want = b"".join(b"@" + n + b"\n" + s + b"\n+\n" + b"~" * len(s) + b"\n" for n, s in recs)
    wrapped = b"# a comment line\n\n" + b"".join(
        b">" + n + b"  \r\n" + b"\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + b"\n\n" for n, s in recs)
    for name, blob in (("in.fa", wrapped), ("in.fa.gz", gzip.compress(wrapped, 1)), ("in.fa.bz2", bz2.compress(wrapped, 1)),
                       ("in.fa.xz", lzma.compress(wrapped, preset=0)), ("noext", wrapped[:-2])):
        path = tmp_path / name
        path.write_bytes(blob)
        reader = textio.TextReader(str(path), 7000)
        assert reader.fasta
        got, sizes = _drain(reader)
        assert sizes == [7000, 7000, 7000, 7000, 2001], name
        assert got == want[:-1], name
    bad = tmp_path / "bad.fa"
    bad.write_bytes(b"# comment\nACGT\n>x\nAC\n")
    with pytest.raises(ValueError, match="line 2"):
        _drain(textio.TextReader(str(bad), 7000))


def test_fastq_through_the_sequential_containers(tmp_path, unpinned):
    """bzip2 / xz FASTQ files (xopen opens them for the reference): same blocks as the plain file."""
    import bz2
    import lzma
    from cutseq_amd import textio
    body = b"".join(b"@r%d x\n%s\n+\n%s\n" % (i, b"ACGT" * (5 + i % 40), b"IIII" * (5 + i % 40)) for i in range(25_001))
    for name, blob in (("in.fq.bz2", bz2.compress(body, 1)), ("in.fq.xz", lzma.compress(body, preset=0))):
        path = tmp_path / name
        path.write_bytes(blob)
        reader = textio.TextReader(str(path), 7000)
        assert not reader.fasta and reader.sequential
        got, sizes = _drain(reader)
        assert sizes == [7000, 7000, 7000, 4001] and got == body[:-1]


def test_output_format_rules():
    """dnaio's rules for output files (OutputFiles(qualities=has_qualities()), cutseq/run.py:437-441, 754-758)."""
    from cutseq_amd import textio
    assert textio.output_format(["a.fastq.gz", "b.fq"], True) is False
    assert textio.output_format(["a.fasta.gz", "b.fa"], True) is True      # FASTQ in, FASTA out: the name decides
    assert textio.output_format(["a.txt"], False) is True                   # no extension to go by: follows the input
    assert textio.output_format(["a.txt"], True) is False
    with pytest.raises(fastq.FastqFormatError, match="no quality values"):
        textio.output_format(["a_trimmed_R1.fastq.gz"], False)
    with pytest.raises(ValueError):
        textio.output_format(["a.fasta", "b.fastq"], True)


def test_stream_writer_sequential_containers(tmp_path, unpinned):
    import bz2
    import lzma
    from cutseq_amd import textio
    parts = [bytes([65 + i % 20]) * n for i, n in enumerate((10, 3_000_000, 0, 9_000_000, 77))]
    for name, undo in (("o.fq.bz2", bz2.decompress), ("o.fq.xz", lzma.decompress)):
        w = textio.StreamWriter(str(tmp_path / name))
        for p in parts:
            buf = fastq.PINNED.take(max(len(p), 1))
            buf[:len(p)] = np.frombuffer(p, dtype=np.uint8)
            w.put(memoryview(buf)[:len(p)], textio._Shared([buf], 1, lambda: None))
        w.close()
        assert undo((tmp_path / name).read_bytes()) == b"".join(parts)


def test_stream_writer_orders_pieces_and_releases_buffers(tmp_path, unpinned):
    from cutseq_amd import textio
    rng = np.random.default_rng(1)
    parts = [rng.integers(65, 90, size=n, dtype=np.uint8) for n in (10, 5_000_000, 0, 17_000_000, 123)]
    for name in ("o.fq", "o.fq.gz"):
        released = []
        w = textio.StreamWriter(str(tmp_path / name))
        for i, p in enumerate(parts):
            if p.size:
                buf = fastq.PINNED.take(p.size)
                buf[: p.size] = p
                w.put(memoryview(buf)[: p.size], textio._Shared([buf], 1, lambda i=i: released.append(i)))
        w.close()
        want = b"".join(p.tobytes() for p in parts)
        raw = (tmp_path / name).read_bytes()
        assert (gzip.decompress(raw) if name.endswith(".gz") else raw) == want
        assert sorted(released) == [0, 1, 3, 4]
    w = textio.StreamWriter(str(tmp_path / "none.fq.gz"))
    w.close()
    assert gzip.decompress((tmp_path / "none.fq.gz").read_bytes()) == b""


def test_rank_shares_reproduce_the_one_process_block_stream(tmp_path, unpinned):
    """ranks.split_inputs: record-index shares of a plain file (no final newline) and of a multi-member gzip file, read
    back share by share, give the same bytes and the same record counts as one reader over the whole file."""
    from cutseq_amd import ranks, textio
    recs1 = [b"@r%d x\n%s\n+\n%s\n" % (i, b"ACGT" * (10 + i % 30), b"IIII" * (10 + i % 30)) for i in range(40_003)]
    recs2 = [b"@r%d y\n%s\n+\n%s\n" % (i, b"TTGA" * (5 + i % 17), b"FFFF" * (5 + i % 17)) for i in range(40_003)]
    p1, p2 = tmp_path / "a_R1.fq", tmp_path / "a_R2.fq.gz"
    p1.write_bytes(b"".join(recs1)[:-1])
    with open(p2, "wb") as fh:
        for lo in range(0, len(recs2), 3500):
            fh.write(gzip.compress(b"".join(recs2[lo:lo + 3500]), 1))
    shares, total = ranks.split_inputs([str(p1), str(p2)], 3)
    assert total == 40_003 and len(shares) == 3
    assert [s[0]["max_records"] for s in shares] == [13334, 13334, None] == [s[1]["max_records"] for s in shares]
    for f, (path, data) in enumerate(((p1, b"".join(recs1)), (p2, b"".join(recs2)))):
        got, sizes = b"", []
        for r in range(3):
            part, ns = _drain(textio.TextReader(str(path), 5000, **shares[r][f]))
            got += part
            sizes.append(sum(ns))
        assert sizes == [13334, 13334, 13335]
        assert got == data.rstrip(b"\n")
    # one gzip member cannot be entered in the middle: the caller is told (and falls back to one process)
    single = tmp_path / "one.fq.gz"
    single.write_bytes(gzip.compress(b"".join(recs2), 1))
    assert ranks.split_inputs([str(p1), str(single)], 2)[0] is None
    assert ranks._strip_option(["-A", "X", "--ranks", "4", "a", "--ranks=2", "b"], "--ranks") == ["-A", "X", "a", "b"]


def test_member_decoder_in_byte_mode_equals_zlib_and_fails_safely():
    """csh_inflate_stream (csrc/pinflate.c, the block loop of pinflate_loop.h instantiated for plain bytes): what a
    multi-member file's members go through.  Same bytes as zlib for every block type and compressor setting; a buffer
    that is too small says so and is never overrun; truncated input and starts in the middle of nowhere are errors."""
    import ctypes as C
    import random
    import zlib
    from cutseq_amd import build
    H = C.CDLL(str(build.build_host()))
    H.csh_inflate_stream.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    rng = random.Random(21)

    def text(kind, n):
        if kind == 0:
            return bytes(rng.choice(b"ACGT") for _ in range(n))
        if kind == 1:
            return rng.randbytes(n)
        if kind == 2:  # long runs: distances below eight, lengths up to 258
            return (b"F" * rng.randrange(1, 600) + b"CG" * rng.randrange(1, 300) + bytes(rng.choice(b"ACGTN") for _ in range(50))) * max(1, n // 900)
        return b"".join(b"@r%d/1\n" % i + bytes(rng.choice(b"ACGT") for _ in range(100)) + b"\n+\n" +
                        bytes(rng.choice(b"FFFFFF:,#") for _ in range(100)) + b"\n" for i in range(n // 200 + 1))

    end, got = C.c_int64(), C.c_int64()
    cases = 0
    for _ in range(40):
        data = text(rng.randrange(4), rng.choice([0, 1, 7, 300, 5000, 70_000, 300_000]))
        for level, strategy in ((1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (1, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (0, 0)):
            c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
            raw = c.compress(data) + c.flush()
            buf = np.frombuffer(raw + bytes(64), dtype=np.uint8).copy()
            cap = len(data) + 400  # (the loop wants 320 bytes of room)
            out = np.zeros(cap, dtype=np.uint8)
            assert H.csh_inflate_stream(buf.ctypes.data, len(raw), 0, out.ctypes.data, cap, C.byref(end), C.byref(got)) == 0
            assert out[:got.value].tobytes() == data and (end.value + 7) // 8 == len(raw)
            cases += 1
            if len(data) > 1000:
                half = len(data) // 2
                guard = np.full(half + 64, 0xAB, dtype=np.uint8)
                assert H.csh_inflate_stream(buf.ctypes.data, len(raw), 0, guard.ctypes.data, half, C.byref(end), C.byref(got)) == -2
                assert (guard[half:] == 0xAB).all()
            if len(raw) > 50:
                cut = np.frombuffer(raw[: len(raw) // 2] + bytes(64), dtype=np.uint8).copy()
                assert H.csh_inflate_stream(cut.ctypes.data, len(raw) // 2, 0, out.ctypes.data, cap, C.byref(end), C.byref(got)) != 0
                H.csh_inflate_stream(buf.ctypes.data, len(raw), rng.randrange(8, len(raw) * 4), out.ctypes.data, cap, C.byref(end), C.byref(got))  # any answer, no crash
    assert cases == 280


def test_multi_member_input_through_either_member_decoder(tmp_path, monkeypatch):
    """GzipSource on a multi-member file: the host library's byte-mode decoder (default) and libdeflate
    (CUTSEQ_OWN_INFLATE=0) hand out the same blocks; a damaged member raises with either."""
    import random
    from cutseq_amd import codec
    rng = random.Random(4)
    members = [b"".join(b"@m%d_%d\n" % (k, i) + bytes(rng.choice(b"ACGT") for _ in range(80)) + b"\n+\n" + b"F" * 80 + b"\n"
                        for i in range(rng.randrange(1, 4000))) for k in range(12)]
    named = b"\x1f\x8b\x08\x08\0\0\0\0\0\xffx.fq\0" + gzip.compress(members[0], 1)[10:]  # a header with a file name
    blob = named + b"".join(gzip.compress(m, rng.choice([1, 6, 9])) for m in members[1:]) + gzip.compress(b"")
    path = tmp_path / "multi.fq.gz"
    pool = fastq._pool()

    def read(data):
        path.write_bytes(data)
        src = codec.GzipSource(str(path), pool)
        out = b"".join(bytes(memoryview(arr)[:n]) for arr, n in src.blocks())
        src.close()
        return out

    want = b"".join(members)
    for own in ("1", "0"):
        monkeypatch.setenv("CUTSEQ_OWN_INFLATE", own)
        assert read(blob) == want
        for damage in (lambda b: b[:-28] + bytes([b[-28] ^ 0x40]) + b[-27:],                         # CRC-32 of the last full member (the empty one behind it is 20 bytes)
                       lambda b: b[: len(b) // 2] + bytes([b[len(b) // 2] ^ 0xFF]) + b[len(b) // 2 + 1:]):  # deflate data in the middle
            with pytest.raises(OSError):
                read(damage(blob))


def test_progress_writes_one_done_line_like_the_reference(monkeypatch):
    """report.Progress: the reference hands cutadapt's Progress() to runner.run (cutseq/run.py:473, 794); off a terminal
    only the final line appears."""
    import io
    from cutseq_amd import report
    buf = io.StringIO()
    p = report.Progress(buf)
    p.t0 -= 5.0
    p.update(3_000_000)
    p.update(5_000_000)
    p.close()
    line = buf.getvalue()
    assert line.count("\n") == 1 and "\r" not in line
    assert line.startswith("Done") and "8,000,000 reads @" in line and "µs/read;" in line and line.rstrip().endswith("M reads/minute")
    monkeypatch.setenv("CUTSEQ_PROGRESS", "0")
    buf = io.StringIO()
    p = report.Progress(buf)
    p.update(10)
    p.close()
    assert buf.getvalue() == ""


def test_a_ranks_share_ends_at_its_member_even_through_the_zlib_fallback(tmp_path, monkeypatch):
    """two concatenated members too large for one libdeflate call (`cat L001.fq.gz L002.fq.gz`), parallel
    inflate off -> the zlib fallback.  It used to decode on across the member boundary to the end of the file with
    offsets of -1, so a rank whose share ended at the boundary also emitted the next rank's records.  Now the fallback
    stops behind ITS member and the walk goes on with real offsets."""
    from cutseq_amd import codec, textio
    a = b"".join(b"@a%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(60_000))
    b = b"".join(b"@b%d\nTTGTACGTAC\n+\nFFFFFFFFFF\n" % i for i in range(50_000))
    za, zb = gzip.compress(a, 1), gzip.compress(b, 1)
    path = tmp_path / "two.fq.gz"
    path.write_bytes(za + zb)
    monkeypatch.setattr(codec, "_MEMBER_CAP", 1 << 20)  # neither ~1.5 MB member fits one call
    for env in ({"CUTSEQ_PARALLEL_INFLATE": "0"}, {}, {"CUTSEQ_OWN_INFLATE": "0", "CUTSEQ_PARALLEL_INFLATE": "0"}):
        for key in ("CUTSEQ_PARALLEL_INFLATE", "CUTSEQ_OWN_INFLATE"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        for pool in (fastq._pool(), None):
            def share(start, stop):
                src = codec.GzipSource(str(path), pool)
                try:
                    return b"".join(bytes(memoryview(arr)[:n]) for arr, n in textio.TextReader._members_until(src, start, stop))
                finally:
                    src.close()
            assert share(0, len(za)) == a, (env, pool)
            assert share(len(za), len(za) + len(zb)) == b, (env, pool)
            src = codec.GzipSource(str(path), pool)
            offs = [off for off, arr, n in src.indexed_blocks()]
            src.close()
            assert len(za) in offs or offs.count(-1) == len(offs)  # real offsets again behind the first member ...
            src = codec.GzipSource(str(path), pool)
            assert b"".join(bytes(memoryview(arr)[:n]) for arr, n in src.blocks()) == a + b  # ... and nothing lost
            src.close()


def test_fasta_record_longer_than_many_blocks_is_converted_once(tmp_path):
    """FastaSource kept the open record in a carry and converted it again from its start with every new
    4 MB block -- quadratic in the record's length.  Now the open record's pieces wait until the block with the next
    record start arrives.  A 40 MB contig between short records: same text as the straight-line conversion, and the
    converter sees every input byte once."""
    from cutseq_amd import codec, textio
    rng = np.random.default_rng(5)
    contig = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=40 << 20).tobytes()
    lines = b"\n".join(contig[i:i + 70] for i in range(0, len(contig), 70))
    text = b">a first\nACGT\nAC\n>big one\n" + lines + b"\n>c\nGGGG\n>d\nTT"
    want = b"@a first\nACGTAC\n+\n~~~~~~\n@big one\n" + contig + b"\n+\n" + b"~" * len(contig) + b"\n@c\nGGGG\n+\n~~~~\n@d\nTT\n+\n~~\n"
    seen = []

    def convert(data, out, final):
        seen.append(len(data))
        return textio.TextReader._fasta_convert(data, out, final)

    class Inner:  # blocks of 1 MB, one of them starting exactly with a record's '>'
        def blocks(self, start=0):
            cuts = sorted({0, len(text)} | set(range(1 << 20, len(text), 1 << 20)) | {text.index(b">c")})
            for lo, hi in zip(cuts, cuts[1:]):
                yield np.frombuffer(text[lo:hi], dtype=np.uint8), hi - lo

        def close(self):
            pass

    src = codec.FastaSource(Inner(), convert, lambda n: np.empty(n, dtype=np.uint8), lambda a: None, "t.fa")
    got = b"".join(bytes(memoryview(arr)[:n]) for arr, n in src.blocks())
    assert got == want
    assert sum(seen) == len(text), (sum(seen), len(text))  # every byte converted once (round 4: ~20 times the contig)


def test_a_chance_member_magic_near_a_huge_member_costs_no_second_inflate(tmp_path, monkeypatch):
    """Round 5: the bytes 1f 8b 08 turn up by chance in compressed data; one of them within reach of a huge member's start
    used to count as "another member starts not far behind it" and the whole member was inflated once more into the
    largest buffer before the parallel decoder got it (0.2 s of a 0.4 s run).  A candidate only counts when its own
    inflate got somewhere.  Here the false magic sits in the header's file-name field."""
    from cutseq_amd import codec
    body = b"@r\nACGTACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIIIIIII\n" * 900_000  # 45 MB: beyond the first trial's 32 MB
    raw = gzip.compress(body, 1)
    blob = b"\x1f\x8b\x08\x08\0\0\0\0\0\xff" + b"a\x1f\x8b\x08\x01b.fq\0" + raw[10:]
    path = tmp_path / "big.fq.gz"
    path.write_bytes(blob)
    monkeypatch.setattr(codec, "_MEMBER_CAP", 40 << 20)  # the member does not fit the largest buffer either
    calls = []
    real = codec.GzipSource._inflate_member_at

    def spy(self, pos, cap, grow):
        calls.append((pos, cap))
        return real(self, pos, cap, grow)

    monkeypatch.setattr(codec.GzipSource, "_inflate_member_at", spy)
    src = codec.GzipSource(str(path), fastq._pool())
    got = 0
    for b, n in src.blocks():
        assert bytes(memoryview(b)[:n]) == body[got:got + n]
        got += n
    src.close()
    assert got == len(body)
    assert any(pos == 11 for pos, _ in calls), calls               # the false candidate was tried ...
    assert [c for c in calls if c[0] == 0] == [(0, 32 << 20)], calls  # ... and the member itself was tried ONCE, at 32 MB
