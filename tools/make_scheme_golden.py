#!/usr/bin/env python3
"""Generate tests/golden/scheme_golden.json by importing the reference's
``cutseq.common`` from /root/reference (works in the build container only; the
reference never travels to the GPU box, only this JSON does).

The fixture holds inputs and the reference's outputs (data), no reference source.
Run:  python tools/make_scheme_golden.py
"""
import contextlib
import io
import json
import sys
from pathlib import Path

sys.path.insert(0, "/root/reference")
from cutseq import common as ref  # noqa: E402

OUT = Path(__file__).resolve().parents[1] / "tests" / "golden" / "scheme_golden.json"

EXTRA_SCHEMES = [
    "AAAA>CCCC",
    "AAAA>CCCCZZZ",
    "aaaa>cccc",
    "AAAA(GG)NNXX-XN(TT)CCCC",
    "ACGT(ACG)NNNNNNNN>ACGTACGT",
    "ACGTACGTAC(ATCACG)NNNNNNNN>AGATCGGAAGAGC",
    "ACGTACGTAC<NNNN(TTAGGC)AGATCGGAAGAGC",
    "ACACGACGCTCTTCCGATCT(ATCACG)NNNNNNNNXX<XXXNNNN(CGATGT)AGATCGGAAGAGCACACGTC",
    "A-C",
    "ACGTNNNN>TTTT",
]
INVALID_SCHEMES = ["AAAAXN>CCCC", "AAAA()>CCCC", "AAAA>", "NAAAA>CCCC", "AAAA(GN)>CCCC", "", ">AAAA", "AAAA"]
RC_CASES = ["", "A", "ACGT", "ACGTNacgtnX", "AGATCGGAAGAGCACACGTC", "ACACGACGCTCTTCCGATCT", "UuRYKM-*"]
SUFFIX_CASES = [
    "a_R1_001.fastq.gz", "a_R2.fq", "a.fastq", "a_R1.txt", "x/y_R1.fq.gz", "a_R1_001.fq.gz.fq",
    "my_sample_R1.fastq.gz", "another_file.fq", "no_suffix_here", "s_R2_001.fq.gz", "t.fq.gz",
    "_R1.fastq", ".fq", "a_R1_001.fastq", "dir.fq/b_R2.fastq.gz",
]


def describe(cfg):
    d = cfg.to_dict()
    for f in ("p5", "p7", "inline5", "inline3", "umi5", "umi3", "mask5", "mask3"):
        part = getattr(cfg, f)
        d[f + "_rc"] = part.rc
        d[f + "_len"] = part.len
        d[f + "_repr"] = repr(part)
    return d


def main():
    g = {"presets": {}, "schemes": {}, "invalid": {}, "rc": {}, "fq_suffix": {}}
    g["preset_order"] = list(ref.BUILDIN_ADAPTERS)
    for name, scheme in ref.BUILDIN_ADAPTERS.items():
        g["presets"][name] = {"scheme": scheme, "parsed": describe(ref.BarcodeConfig(scheme))}
    for s in EXTRA_SCHEMES:
        g["schemes"][s] = describe(ref.BarcodeConfig(s))
    for s in INVALID_SCHEMES:
        try:
            ref.BarcodeConfig(s)
            g["invalid"][s] = "ok"
        except SystemExit as e:
            g["invalid"][s] = f"exit:{e.code}"
    g["empty_config"] = describe(ref.BarcodeConfig())
    for s in RC_CASES:
        g["rc"][s] = ref.reverse_complement(s)
    for s in SUFFIX_CASES:
        g["fq_suffix"][s] = ref.remove_fq_suffix(s)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ref.print_builtin_adapters()
    g["list_adapters_stdout"] = buf.getvalue()
    OUT.write_text(json.dumps(g, indent=1, sort_keys=True) + "\n")
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes)")


if __name__ == "__main__":
    main()
