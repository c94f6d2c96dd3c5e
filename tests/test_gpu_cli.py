"""End to end on the GPU: the cutseq-compatible CLI on the 1000-pair slice of the reference's
input data (BASELINE.json config 1), decompressed output compared byte for byte with the
string-level restatement."""
import gzip
import json

import numpy as np

import pytest

from cutseq_amd import plan as planmod, run as cli
from cutseq_amd.common import BUILDIN_ADAPTERS

import util

pytestmark = pytest.mark.gpu

R1 = str(util.GOLDEN / "fixture1k_R1.fq.gz")
R2 = str(util.GOLDEN / "fixture1k_R2.fq.gz")


def gunzip(path):
    with gzip.open(path, "rb") as fh:
        return fh.read()


def expected(scheme, flags, paired, untrimmed_requested=False):
    st = planmod.CutadaptConfig()
    for k, v in flags.items():
        setattr(st, k, v)
    rec1 = util.read_fastq_gz(R1)
    rec2 = util.read_fastq_gz(R2) if paired else None
    batch = util.batch_from_records(rec1, rec2)
    out = util.pyref_run(scheme, st, batch, [r[0] for r in rec1], [r[0] for r in rec2] if paired else None,
                         untrimmed_requested)
    streams = {}
    for route, name in enumerate(("trimmed", "short", "untrimmed")):
        streams[name] = (b"".join(x[1] for x in out if x[0] == route),
                         b"".join(x[2] for x in out if x[0] == route) if paired else None)
    return streams


# Every switch the product reads from the environment gets the command line run through it:
# the host parser / formatter instead of the text path, the scan kernel without the merged adapter-pair walk and without
# the existence-only 5' scan, gzip outputs deflated on the host instead of on the device.
SWITCHES = ["", "CUTSEQ_FAST_RECODE=0", "CUTSEQ_TEXT_PATH=0", "CUTSEQ_PAIR=0", "CUTSEQ_EXISTS=0", "CUTSEQ_CUT_RUNS=0", "CUTSEQ_GPU_DEFLATE=0", "CUTSEQ_GPU_LZ=0",
            "CUTSEQ_ITEM_SLOTS=1", "CUTSEQ_LOG_ALWAYS=1"]


@pytest.fixture(params=SWITCHES)
def switch(request, monkeypatch):
    if request.param:
        monkeypatch.setenv(*request.param.split("="))
    return request.param


def test_cli_paired_takarav3(tmp_path, capsys, switch):
    prefix = str(tmp_path / "out")
    cli.main(["-A", "TAKARAV3", "--trim-polyA", "-O", prefix, "--json-file", str(tmp_path / "r.json"), R1, R2])
    want = expected(BUILDIN_ADAPTERS["TAKARAV3"], {"trim_polyA": True}, True)
    for kind in ("trimmed", "short"):
        assert gunzip(f"{prefix}_{kind}_R1.fastq.gz") == want[kind][0]
        assert gunzip(f"{prefix}_{kind}_R2.fastq.gz") == want[kind][1]
    rep = json.loads((tmp_path / "r.json").read_text())
    assert rep["read_counts"]["input"] == 1000
    assert rep["read_counts"]["output"] + rep["read_counts"]["filtered"]["too_short"] == 1000
    assert rep["basepair_counts"]["output_read1"] == sum(len(l) for l in want["trimmed"][0].split(b"\n")[1::4])
    assert rep["tag"] == "Cutadapt report" and rep["engine"]["name"] == "cutseq_amd"
    err = capsys.readouterr().err
    assert "status\tin_reads\tin_bp\ttoo_short\ttoo_long\ttoo_many_n\tout_reads" in err and "\nOK\t1000\t" in err


def test_cli_paired_auto_rc_swaps_outputs(tmp_path):
    o1, o2 = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")  # plain (no .gz): written uncompressed
    s1, s2 = str(tmp_path / "s1.fq.gz"), str(tmp_path / "s2.fq.gz")
    cli.main([R1, R2, "-A", "TAKARAV3", "--auto-rc", "-o", o1, o2, "-s", s1, s2])  # positionals first: -s is nargs="+"
    want = expected(BUILDIN_ADAPTERS["TAKARAV3"], {"auto_rc": True}, True)
    assert open(o1, "rb").read() == want["trimmed"][1]  # '-' strand library: R2 goes to the first file
    assert open(o2, "rb").read() == want["trimmed"][0]
    assert gunzip(s1) == want["short"][0] and gunzip(s2) == want["short"][1]


def test_cli_single_end_rc_and_untrimmed(tmp_path, switch):
    out, short, untr = (str(tmp_path / f"{n}.fastq.gz") for n in ("o", "s", "u"))
    scheme = "ACACGACGCTCTTCCGATCT(GGG)NNN<XXXAGATCGGAAGAGCACACGTC"
    cli.main([R1, "-a", scheme, "--auto-rc", "--ensure-inline-barcode", "--trim-polyA", "-o", out, "-s", short, "-u", untr])
    want = expected(scheme, {"auto_rc": True, "ensure_inline_barcode": True, "trim_polyA": True}, False, True)
    assert gunzip(out) == want["trimmed"][0]
    assert gunzip(short) == want["short"][0]
    assert gunzip(untr) == want["untrimmed"][0]
    assert len(want["untrimmed"][0]) > 0


@pytest.mark.parametrize("gz", [False, True])
def test_cli_zero_length_final_record(tmp_path, gz):
    """A file whose LAST record is empty ("@id\\n\\n+\\n\\n": dnaio reads it, already-trimmed inputs hold such records)
    used to die with 'truncated FASTQ record at end of file' (textio.TextReader); with and without the
    empty quality line's own line end, plain and gzip."""
    rec = util.read_fastq_gz(R1, limit=200) + [(b"last 1:N:0:X", b"", b"")]
    body = b"".join(b"@" + n + b"\n" + s + b"\n+\n" + q + b"\n" for n, s, q in rec)
    batch = util.batch_from_records(rec)
    st = planmod.CutadaptConfig()
    st.min_length = 0
    want = util.pyref_run(BUILDIN_ADAPTERS["TAKARAV3"], st, batch, [r[0] for r in rec])
    want = b"".join(x[1] for x in want if x[0] == 0)
    assert want.endswith(b"@last_\n\n+\n\n")  # (the read is empty, and so is the UMI behind the id)
    for k, text in enumerate((body, body[:-1], body + b"\n\n")):
        src = str(tmp_path / (f"in{k}.fq.gz" if gz else f"in{k}.fq"))
        with open(src, "wb") as fh:
            fh.write(gzip.compress(text, 1) if gz else text)
        out = str(tmp_path / f"o{k}.fq")
        cli.main([src, "-A", "TAKARAV3", "-m", "0", "-o", out, "-s", str(tmp_path / f"s{k}.fq")])
        assert open(out, "rb").read() == want, k


def _fasta_expected(rec, scheme, st):
    """The FASTA counterpart of `expected`: the oracle sees the reads with a quality no cutoff trims (the
    QualityTrimmer step does nothing without qualities), the records leave as '>id\\nsequence\\n'."""
    batch = util.batch_from_records([(n, s, b"~" * len(s)) for n, s, _ in rec])
    out = util.pyref_run(scheme, st, batch, [r[0] for r in rec])
    streams = {}
    for route, name in enumerate(("trimmed", "short", "untrimmed")):
        recs = [x[1] for x in out if x[0] == route]
        streams[name] = b"".join(b">" + r.split(b"\n")[0][1:] + b"\n" + r.split(b"\n")[1] + b"\n" for r in recs)
    return streams


def test_cli_fasta_and_the_other_containers(tmp_path):
    """What dnaio / xopen give the reference for free (cutseq/run.py:434-441, 751-758): FASTA input (here with wrapped
    sequence lines) -> FASTA output, quality trimming a no-op; FASTQ input to a file named .fasta; bzip2 / xz files on
    both sides; FASTQ output names for a FASTA input are refused as dnaio refuses them."""
    import bz2
    import lzma
    rec = util.read_fastq_gz(R1, limit=400)
    st = planmod.CutadaptConfig()
    scheme = BUILDIN_ADAPTERS["TAKARAV3"]
    want = _fasta_expected(rec, scheme, st)
    fasta = b"".join(b">" + n + b"\n" + b"\n".join(s[i:i + 70] for i in range(0, len(s), 70)) + b"\n" for n, s, _ in rec)
    (tmp_path / "in.fa").write_bytes(fasta)
    (tmp_path / "in.fasta.bz2").write_bytes(bz2.compress(fasta))
    for k, (src, dst, undo) in enumerate((("in.fa", "o.fasta", bytes), ("in.fasta.bz2", "o.fa.xz", lzma.decompress),
                                          ("in.fa", "o.fa.gz", gzip.decompress))):
        out, short = str(tmp_path / f"{k}_{dst}"), str(tmp_path / f"{k}_s_{dst}")
        cli.main([str(tmp_path / src), "-A", "TAKARAV3", "-o", out, "-s", short])
        assert undo(open(out, "rb").read()) == want["trimmed"], (src, dst)
        assert undo(open(short, "rb").read()) == want["short"], (src, dst)
    from cutseq_amd import fastq
    with pytest.raises(fastq.FastqFormatError, match="no quality values"):  # (an exception, as from dnaio in the reference)
        cli.main([str(tmp_path / "in.fa"), "-A", "TAKARAV3", "-O", str(tmp_path / "pre")])  # -> *_trimmed_R1.fastq.gz
    # FASTQ input, FASTA-named output: sequences of the ordinary run (quality trimming included), no quality lines
    fq = expected(scheme, {}, False)
    out, short = str(tmp_path / "q.fa.bz2"), str(tmp_path / "q_s.fa.bz2")
    cli.main([R1, "-A", "TAKARAV3", "-o", out, "-s", short])
    lines = fq["trimmed"][0].split(b"\n")
    as_fasta = b"".join(b">" + lines[i][1:] + b"\n" + lines[i + 1] + b"\n" for i in range(0, len(lines) - 1, 4))
    assert bz2.decompress(open(out, "rb").read()) == as_fasta
    # xz FASTQ in, bzip2 FASTQ out
    (tmp_path / "in.fq.xz").write_bytes(lzma.compress(gunzip(R1), preset=0))
    out, short = str(tmp_path / "x.fq.bz2"), str(tmp_path / "x_s.fq.bz2")
    cli.main([str(tmp_path / "in.fq.xz"), "-A", "TAKARAV3", "-o", out, "-s", short])
    assert bz2.decompress(open(out, "rb").read()) == fq["trimmed"][0]
    assert bz2.decompress(open(short, "rb").read()) == fq["short"][0]


def test_cli_standard_input_and_output(tmp_path):
    """'-' as input / output file name (xopen): FASTQ text and a gzip stream on standard input, the trimmed records on
    standard output, in a child process."""
    import subprocess
    import sys
    want = expected(BUILDIN_ADAPTERS["TAKARAV3"], {}, False)
    root = str(util.GOLDEN.parents[1])
    for blob in (gunzip(R1), open(R1, "rb").read()):
        short = str(tmp_path / "s.fq")
        r = subprocess.run([sys.executable, "-m", "cutseq_amd.run", "-", "-A", "TAKARAV3", "-o", "-", "-s", short],
                           input=blob, capture_output=True, cwd=root, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        assert r.stdout == want["trimmed"][0]
        assert open(short, "rb").read() == want["short"][0]


# ---------------------------------------------------------------- the runner beyond one chunk


def crc_streams(streams):
    import zlib
    return {k: tuple(None if s is None else (len(s), zlib.crc32(s)) for s in v) for k, v in streams.items()}


def oracle_streams(tp, batch, names1, names2):
    """trimmed / short / untrimmed byte streams from the C oracle's results + the host formatter."""
    (r1, cap2, _), m2 = util.oracle_run(tp, batch, threads=8)
    recs = util.format_batch(tp, batch, names1, names2, r1, cap2, m2[0] if m2 else None)
    out = {}
    for route, name in enumerate(("trimmed", "short", "untrimmed")):
        out[name] = (b"".join(x[1] for x in recs if x[0] == route),
                     b"".join(x[2] for x in recs if x[0] == route) if tp.paired else None)
    return out


def read_streams(prefix, paired, kinds=("trimmed", "short")):
    return {k: (gunzip(f"{prefix}_{k}_R1.fastq.gz"), gunzip(f"{prefix}_{k}_R2.fastq.gz") if paired else None)
            for k in kinds}


def test_cli_many_chunks_two_engines_on_one_gpu(tmp_path, monkeypatch):
    """200 000 synthetic pairs = four chunks of 65 536 through TWO engines on GPU 0 (CUTSEQ_DEVICES=0,0):
    round-robin over the device workers, both staging slots of each engine reused, results re-ordered
    behind the GPUs.  Input as multi-member gzip.  Streams compared by length + CRC-32 with the oracle's
    results run through the host formatter (counterpart of make_runner(cores=N), cutseq/run.py:436, 753)."""
    from cutseq_amd import synth
    n = 200_000
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    batch = synth.generate_pairs(n, 150, seed=77, poly_fraction=0.05, art5_fraction=0.01, indel_frac=0.1)
    names1 = [f"SIM:{i} 1:N:0:X".encode() for i in range(n)]
    names2 = [f"SIM:{i} 2:N:0:X".encode() for i in range(n)]
    in1, in2 = str(tmp_path / "in_R1.fastq.gz"), str(tmp_path / "in_R2.fastq.gz")
    util.write_fastq(in1, names1, batch.seq1, batch.qual1, batch.len1, gz_members=30_000)
    util.write_fastq(in2, names2, batch.seq2, batch.qual2, batch.len2, gz_members=50_000)
    monkeypatch.setenv("CUTSEQ_DEVICES", "0,0")
    monkeypatch.setenv("CUTSEQ_CHUNK_READS", "65536")  # (the text path's default block is 262 144 records)
    prefix = str(tmp_path / "out")
    cli.main(["-A", "TAKARAV3", "--trim-polyA", "-O", prefix, "--json-file", str(tmp_path / "r.json"), in1, in2])
    got = read_streams(prefix, True)
    want = oracle_streams(tp, batch, names1, names2)
    want.pop("untrimmed")
    assert crc_streams(got) == crc_streams(want)
    rep = json.loads((tmp_path / "r.json").read_text())
    assert rep["read_counts"]["input"] == n and rep["engine"]["devices"] == [0, 0]
    assert len(rep["engine"]["per_device"]) == 2  # both engines saw chunks


def test_cli_full_reference_input_in_small_chunks(tmp_path, monkeypatch):
    """BASELINE.json config 1 at full size: the reference's own 10 000-pair input (tests/golden/fixture10k_*,
    a copy of test/input_R{1,2}.fq.gz), defaults, in chunks of 1 500 records so that seven chunks cycle
    through the two staging slots.  Byte for byte against the string-level restatement."""
    r1, r2 = str(util.GOLDEN / "fixture10k_R1.fq.gz"), str(util.GOLDEN / "fixture10k_R2.fq.gz")
    monkeypatch.setenv("CUTSEQ_CHUNK_READS", "1500")
    prefix = str(tmp_path / "full")
    cli.main(["-A", "TAKARAV3", "-O", prefix, r1, r2])
    rec1, rec2 = util.read_fastq_gz(r1), util.read_fastq_gz(r2)
    assert len(rec1) == len(rec2) == 10_000
    batch = util.batch_from_records(rec1, rec2)
    st = planmod.CutadaptConfig()
    out = util.pyref_run(BUILDIN_ADAPTERS["TAKARAV3"], st, batch, [r[0] for r in rec1], [r[0] for r in rec2])
    for route, kind in enumerate(("trimmed", "short")):
        assert gunzip(f"{prefix}_{kind}_R1.fastq.gz") == b"".join(x[1] for x in out if x[0] == route)
        assert gunzip(f"{prefix}_{kind}_R2.fastq.gz") == b"".join(x[2] for x in out if x[0] == route)


def test_cli_longer_reads_appear_in_a_later_chunk(tmp_path, monkeypatch):
    """The row stride has to grow in the middle of a run (a 300-nt read in the third chunk): the engine is
    rebuilt only after everything it still has in flight came back, and its counters are kept."""
    from cutseq_amd import synth
    n = 6000
    st = planmod.CutadaptConfig()
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    short = synth.generate_pairs(n, 100, seed=5)
    long_ = synth.generate_pairs(n, 300, seed=6)
    reads1, reads2 = [], []
    for i in range(n):
        src = long_ if (i >= 2500 and i % 7 == 0) else short
        reads1.append((util.row_bytes(src.seq1, src.len1, i).decode(), util.row_bytes(src.qual1, src.len1, i).decode()))
        reads2.append((util.row_bytes(src.seq2, src.len2, i).decode(), util.row_bytes(src.qual2, src.len2, i).decode()))
    batch = util.batch_from_reads(reads1, reads2)
    names1 = [f"r{i}/1".encode() for i in range(n)]
    names2 = [f"r{i}/2".encode() for i in range(n)]
    in1, in2 = str(tmp_path / "a_R1.fq"), str(tmp_path / "a_R2.fq")
    util.write_fastq(in1, names1, batch.seq1, batch.qual1, batch.len1)
    util.write_fastq(in2, names2, batch.seq2, batch.qual2, batch.len2)
    monkeypatch.setenv("CUTSEQ_CHUNK_READS", "1000")
    prefix = str(tmp_path / "o")
    cli.main(["-A", "TAKARAV3", "-O", prefix, "--json-file", str(tmp_path / "r.json"), in1, in2])
    want = oracle_streams(tp, batch, names1, names2)
    want.pop("untrimmed")
    assert crc_streams(read_streams(prefix, True)) == crc_streams(want)
    rep = json.loads((tmp_path / "r.json").read_text())
    (o1, _, s1), _ = util.oracle_run(tp, batch)
    # the 5' adapter count of the report comes from the device counters of BOTH engines the run has used
    assert rep["read_counts"]["read1_with_adapter"] == int(s1.op_matched[0])
    assert rep["basepair_counts"]["quality_trimmed_read1"] == int(s1.qualtrim_bp)


def test_cli_reads_of_any_length(tmp_path, monkeypatch):
    """The reference streams records of any length (cutseq/run.py:434, 751).  5 k / 20 k / 1 537-nt reads mixed into a
    150-nt library: the long ones leave the tile kernels' rows and walk their chain in the long-read kernel, the
    output (all three routes) equals the oracle's through the record logic; also a mate pair where only ONE mate
    is long, adapters inside long reads, poly tails and quality tails on them."""
    from cutseq_amd import synth
    from cutseq_amd.common import reverse_complement as rc
    rng = np.random.default_rng(17)
    n = 3000
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    bc = tp  # noqa: F841
    base = synth.generate_pairs(n, 150, seed=41, poly_fraction=0.05)
    p7, p5rc = "AGATCGGAAGAGCACACGTC", rc("ACACGACGCTCTTCCGATCT")
    reads1 = [(util.row_bytes(base.seq1, base.len1, i).decode(), util.row_bytes(base.qual1, base.len1, i).decode()) for i in range(n)]
    reads2 = [(util.row_bytes(base.seq2, base.len2, i).decode(), util.row_bytes(base.qual2, base.len2, i).decode()) for i in range(n)]

    def dna(k):
        return "".join("ACGT"[x] for x in rng.integers(0, 4, size=k))

    def quals(k, tail):
        return "I" * (k - tail) + "#" * tail

    specials = {
        5: (dna(5000), dna(5000)),                                                  # nothing to find in 5 kb
        700: (dna(12000) + "TTT" + dna(8) + p7 + dna(300), dna(14) + dna(400)),     # 3' adapter deep inside a 12 kb read; mate short
        701: (dna(150), "T" * 30 + dna(20000) + p5rc[:13]),                         # only mate 2 long: poly-T head, partial adapter at the end
        1500: (dna(1537), dna(1600)),                                               # just over the tile limit
        2999: (dna(3) + dna(2500) + "A" * 60 + dna(14) + p7, dna(8) + dna(6) + dna(2500)),
    }
    for i, (a, b) in specials.items():
        reads1[i] = (a, quals(len(a), 40 if i % 2 else 0))
        reads2[i] = (b, quals(len(b), 0 if i % 2 else 25))
    batch = util.batch_from_reads(reads1, reads2)
    names1 = [f"r{i} 1:N".encode() for i in range(n)]
    names2 = [f"r{i} 2:N".encode() for i in range(n)]
    in1, in2 = str(tmp_path / "a_R1.fq"), str(tmp_path / "a_R2.fq")
    util.write_fastq(in1, names1, batch.seq1, batch.qual1, batch.len1)
    util.write_fastq(in2, names2, batch.seq2, batch.qual2, batch.len2)
    monkeypatch.setenv("CUTSEQ_CHUNK_READS", "1000")
    prefix = str(tmp_path / "o")
    cli.main(["-A", "TAKARAV3", "--trim-polyA", "-O", prefix, "--json-file", str(tmp_path / "r.json"), in1, in2])
    want = oracle_streams(tp, batch, names1, names2)
    want.pop("untrimmed")
    got = read_streams(prefix, True)
    assert crc_streams(got) == crc_streams(want)
    rep = json.loads((tmp_path / "r.json").read_text())
    (o1, _, s1), (o2, _, s2) = util.oracle_run(tp, batch)
    assert rep["read_counts"]["input"] == n
    assert rep["basepair_counts"]["input_read1"] == int(batch.len1.sum()) and rep["basepair_counts"]["input_read2"] == int(batch.len2.sum())
    assert rep["basepair_counts"]["quality_trimmed_read1"] == int(s1.qualtrim_bp)
    assert int(o1["stop"][700]) < 12100 and int(o1["flags"][700]) & 2  # the adapter inside the 12 kb read was found


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_ranks_rehearsal_on_one_gpu(launcher):
    """bench.py's N > 1 path (one process per GPU, shard by global read index, no data-path collective) with two
    ranks on GPU 0 and gloo for the barrier (CUTSEQ_BENCH_REHEARSAL=1; RCCL wants one device per rank).  The CPU
    sample at N = 1 checks the device results against the oracle; here the two-rank run must report twice the
    work of one rank and a positive rate.  Both launch forms: plain ``python bench.py --gpus 2`` (the script spawns
    its own ranks, the way the driver starts --gpus 1) and under torch.distributed.run."""
    import os
    import socket
    import subprocess
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CUTSEQ_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    tail = [str(util.GOLDEN.parents[1] / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "300000"]
    if launcher == "self":
        cmd = [sys.executable] + tail
        for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
            env.pop(name, None)
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + tail
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rep = json.loads(line)
    assert rep["n_gpus"] == 2 and rep["scaling"] == "weak" and rep["value"] > 0
    assert rep["config"]["parallelism"] == "shard2" and rep["config"]["pairs_per_step_per_gpu"] == 300000
    assert "cpu_baseline" not in rep  # rank 0 at N = 1 only
    # what a reader of an N-GPU line asks of it: every rank's step time and event-timed scan kernel,
    # the device each rank ran on, the world as the collective backend reports it, the per-rank parity gate
    pr = rep["per_rank"]
    assert len(pr["ms_per_step"]["each"]) == 2 and pr["ms_per_step"]["min"] <= pr["ms_per_step"]["median"] <= pr["ms_per_step"]["max"]
    assert abs(pr["ms_per_step"]["max"] - rep["ms_per_step"]) < 1e-3  # the line's figure is the slowest rank's
    assert len(pr["scan_kernel_ms"]) == 2 and all(x > 0 for x in pr["scan_kernel_ms"]) and len(pr["resolve_kernel_ms"]) == 2
    assert pr["rccl_world"] == 2 and pr["backend"] == "gloo"  # (the rehearsal's backend; RCCL on a real node)
    assert len(pr["devices"]) == 2 and all("MI355X" in dv.get("name", "") or dv.get("name") for dv in pr["devices"])
    assert len({dv["pid"] for dv in pr["devices"]}) == 2 and pr["distinct_devices"] == 1  # two ranks on GPU 0: visible
    assert rep["all_ranks_identical"] is True
    assert rep["batch_generator"]["where"].startswith("device") and rep["batch_generator"]["device_bytes_equal_host_bytes_on_head"] is True


@pytest.mark.parametrize("layout", ["plain+gz", "gz-members-in-lockstep", "gz-members-out-of-step"])
def test_cli_two_ranks_equal_one_process(tmp_path, monkeypatch, layout):
    """--ranks 2 (two processes, both on GPU 0 here), every rank with its own reader, engine and output part; the
    concatenated output and the report equal the one-process run (counterpart of make_runner(cores=N),
    cutseq/run.py:436, 753).  Layouts: one mate plain text, the other multi-member gzip (split at record indices from
    a counting pass); both mates gzip with the same records per member (split by member index, nothing inflated
    twice); gzip members that do not line up (the parent's id check at the split point sends it to the exact split)."""
    from cutseq_amd import synth
    n = 50_000
    batch = synth.generate_pairs(n, 150, seed=31, poly_fraction=0.05)
    names1 = [s.encode() for s in synth.headers(n, 1)]
    names2 = [s.encode() for s in synth.headers(n, 2)]
    in1 = str(tmp_path / ("in_R1.fastq" if layout == "plain+gz" else "in_R1.fastq.gz"))
    in2 = str(tmp_path / "in_R2.fastq.gz")
    util.write_fastq(in1, names1, batch.seq1, batch.qual1, batch.len1, gz_members=6_000)
    util.write_fastq(in2, names2, batch.seq2, batch.qual2, batch.len2, gz_members=6_250 if layout.endswith("out-of-step") else 6_000)
    monkeypatch.setenv("CUTSEQ_CHUNK_READS", "8000")
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    cli.main(["-A", "TAKARAV3", "--trim-polyA", "-O", one, "--json-file", str(tmp_path / "one.json"), in1, in2])
    monkeypatch.setenv("CUTSEQ_DEVICES", "0,0")
    cli.main(["-A", "TAKARAV3", "--trim-polyA", "-O", two, "--json-file", str(tmp_path / "two.json"), "--ranks", "2", in1, in2])
    for kind in ("trimmed", "short"):
        for mate in (1, 2):
            assert gunzip(f"{two}_{kind}_R{mate}.fastq.gz") == gunzip(f"{one}_{kind}_R{mate}.fastq.gz"), (kind, mate)
    a, b = json.loads((tmp_path / "one.json").read_text()), json.loads((tmp_path / "two.json").read_text())
    assert a["read_counts"] == b["read_counts"] and a["basepair_counts"] == b["basepair_counts"]
    assert b["read_counts"]["input"] == n and len(b["engine"]["per_device"]) == 2
    assert not list(tmp_path.glob("*.part*"))
    assert b["engine"]["ranks"] == 2 and b["engine"]["threads_per_rank"] >= 1
    assert b["engine"]["ranks_split"].startswith("gzip members" if layout == "gz-members-in-lockstep" else "record indices")
