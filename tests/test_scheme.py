"""Scheme parser / presets / filename helpers against the golden table captured from the
reference's cutseq.common (tools/make_scheme_golden.py)."""
import contextlib
import io
import json

import pytest

from cutseq_amd import common

FIELDS = ("p5", "p7", "inline5", "inline3", "umi5", "umi3", "mask5", "mask3")


@pytest.fixture(scope="module")
def golden(golden_dir):
    return json.loads((golden_dir / "scheme_golden.json").read_text())


def describe(cfg):
    d = cfg.to_dict()
    for f in FIELDS:
        part = getattr(cfg, f)
        d[f + "_rc"], d[f + "_len"], d[f + "_repr"] = part.rc, part.len, repr(part)
    return d


def test_preset_table_matches_reference(golden):
    assert list(common.BUILDIN_ADAPTERS) == golden["preset_order"]
    assert len(common.BUILDIN_ADAPTERS) == 18
    for name, g in golden["presets"].items():
        assert common.BUILDIN_ADAPTERS[name] == g["scheme"]
        assert describe(common.BarcodeConfig(g["scheme"])) == g["parsed"], name


def test_extra_schemes(golden):
    for scheme, parsed in golden["schemes"].items():
        assert describe(common.BarcodeConfig(scheme)) == parsed, scheme
    assert describe(common.BarcodeConfig()) == golden["empty_config"]


def test_invalid_schemes_exit_1(golden):
    for scheme, outcome in golden["invalid"].items():
        assert outcome == "exit:1"
        with pytest.raises(SystemExit) as e:
            common.BarcodeConfig(scheme)
        assert e.value.code == 1, scheme


def test_reverse_complement(golden):
    for s, rc in golden["rc"].items():
        assert common.reverse_complement(s) == rc


def test_remove_fq_suffix(golden):
    for s, out in golden["fq_suffix"].items():
        assert common.remove_fq_suffix(s) == out, s


def test_list_adapters_output(golden):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        common.print_builtin_adapters()
    assert buf.getvalue() == golden["list_adapters_stdout"]


def test_takarav3_readme_narrative():
    """reference README.md:13-26: TAKARAV3 = 3 nt mask / 6 nt mask + 8 nt UMI, '-' strand."""
    b = common.BarcodeConfig(common.BUILDIN_ADAPTERS["TAKARAV3"])
    assert (b.p5.fw, b.p7.fw) == ("ACACGACGCTCTTCCGATCT", "AGATCGGAAGAGCACACGTC")
    assert (b.p5.rc, b.p7.rc) == ("AGATCGGAAGAGCGTCGTGT", "GACGTGTGCTCTTCCGATCT")
    assert (b.umi5.len, b.umi3.len, b.mask5.len, b.mask3.len, b.strand) == (0, 8, 3, 6, "-")
