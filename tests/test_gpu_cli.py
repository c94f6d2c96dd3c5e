"""End to end on the GPU: the cutseq-compatible CLI on the 1000-pair slice of the reference's
input data (BASELINE.json config 1), decompressed output compared byte for byte with the
string-level restatement."""
import gzip
import json

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


def test_cli_paired_takarav3(tmp_path, capsys):
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


def test_cli_single_end_rc_and_untrimmed(tmp_path):
    out, short, untr = (str(tmp_path / f"{n}.fastq.gz") for n in ("o", "s", "u"))
    scheme = "ACACGACGCTCTTCCGATCT(GGG)NNN<XXXAGATCGGAAGAGCACACGTC"
    cli.main([R1, "-a", scheme, "--auto-rc", "--ensure-inline-barcode", "--trim-polyA", "-o", out, "-s", short, "-u", untr])
    want = expected(scheme, {"auto_rc": True, "ensure_inline_barcode": True, "trim_polyA": True}, False, True)
    assert gunzip(out) == want["trimmed"][0]
    assert gunzip(short) == want["short"][0]
    assert gunzip(untr) == want["untrimmed"][0]
    assert len(want["untrimmed"][0]) > 0
