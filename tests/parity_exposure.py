#!/usr/bin/env python3
"""(python tests/parity_exposure.py)  How much of "byte-identical FASTQ" rests on each recalled cutadapt rule.

The per-read arithmetic of the reference lives in cutadapt (reference pyproject.toml:17), which is absent
here: four of its rules are restated from recollection and each is a switch (include/cutseq_hip.h):
    select_rule  CS_SELECT_LEFTMOST (default, cutadapt >= 4.0)  vs  CS_SELECT_SCORE (3.x)
    shortcut     CS_SHORTCUT_NONE   (default, cutadapt >= 3)    vs  CS_SHORTCUT_FIND (<= 2.x str.find)
    case_rule    CS_CASE_FOLD       (default, sequence.upper()) vs  CS_CASE_SENSITIVE
    indel_tie    CS_TIE_INSERTION   (default, SURVEY B.2 order) vs  CS_TIE_DELETION
This script counts, with the CPU oracle, the read pairs whose output record (interval, UMI, route) changes
when ONE switch is flipped away from the default, on
    * the reference's own 10 000-pair input (tests/golden/fixture10k_*, -A TAKARAV3, defaults), and
    * 1 000 000 synthetic pairs (seed 0xC0FFEE, TAKARAV3 + --trim-polyA, the bench workload),
and writes profiles/parity_exposure.json.  Lives under tests/ because it drives the oracle.
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402

import util  # noqa: E402
from cutseq_amd import abi, plan as planmod, synth  # noqa: E402
from cutseq_amd.common import BUILDIN_ADAPTERS  # noqa: E402

SWITCHES = {
    "select_rule": abi.CS_SELECT_SCORE,
    "shortcut": abi.CS_SHORTCUT_FIND,
    "case_rule": abi.CS_CASE_SENSITIVE,
    "indel_tie": abi.CS_TIE_DELETION,
}


def results(batch, flags, **rules):
    st = planmod.CutadaptConfig()
    for k, v in {**flags, **rules}.items():
        setattr(st, k, v)
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    (r1, _, _), (r2, _, _) = util.oracle_run(tp, batch, threads=8)
    return r1, r2


def exposure(batch, flags):
    base1, base2 = results(batch, flags)
    out = {}
    for name, value in SWITCHES.items():
        r1, r2 = results(batch, flags, **{name: value})
        changed = (r1 != base1) | (r2 != base2)
        out[name] = {"pairs_changed": int(changed.sum()), "fraction": float(changed.mean()),
                     "mate1_changed": int((r1 != base1).sum()), "mate2_changed": int((r2 != base2).sum())}
    return out


def main():
    rep = {"note": "pairs whose output record changes when ONE recalled cutadapt rule is flipped away from the default; "
                   "CPU oracle (oracle/cutseq_oracle.c); see tests/parity_exposure.py"}
    rec1 = util.read_fastq_gz(util.GOLDEN / "fixture10k_R1.fq.gz")
    rec2 = util.read_fastq_gz(util.GOLDEN / "fixture10k_R2.fq.gz")
    fixture = util.batch_from_records(rec1, rec2)
    rep["reference_input_10k_pairs"] = {"pairs": fixture.n, "command": "-A TAKARAV3 (defaults)",
                                        "switches": exposure(fixture, {})}
    n = 1_000_000
    batch = synth.generate_pairs(n, 150)
    rep["synthetic_1M_pairs"] = {"pairs": n, "command": "-A TAKARAV3 --trim-polyA, seed 0xC0FFEE (bench.py workload)",
                                 "switches": exposure(batch, {"trim_polyA": True})}
    masked = synth.generate_pairs(200_000, 150)
    util.soft_mask(masked, 0.2)
    rep["synthetic_200k_pairs_soft_masked"] = {
        "pairs": masked.n, "command": "same, a random stretch of 20 % of the reads in lower case",
        "switches": {"case_rule": exposure(masked, {"trim_polyA": True})["case_rule"]}}
    (ROOT / "profiles" / "parity_exposure.json").write_text(json.dumps(rep, indent=1) + "\n")
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
