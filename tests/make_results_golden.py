#!/usr/bin/env python3
"""(python tests/make_results_golden.py)  Freeze the oracle's results on a seeded synthetic batch as CRC-32 values (tests/golden/synth_results_crc.json).
The CPU suite checks the oracle still reproduces them, the GPU suite checks the device does: a change to
either side that alters results is caught even when both change together."""
import json
import sys
import zlib
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))  # lives under tests/: it drives the oracle, which only tests may do
import util  # noqa: E402
from cutseq_amd import plan as planmod, synth  # noqa: E402
from cutseq_amd.common import BUILDIN_ADAPTERS  # noqa: E402

CASES = [("TAKARAV3", {"trim_polyA": True}, 0), ("TAKARAV3", {"trim_polyA": True}, 1),
         ("SACSEQV3", {"trim_polyA": True}, 0), ("INLINE", {"ensure_inline_barcode": True}, 0)]
N, SEED = 200_000, 20260101


def crc_case(name, flags, rule, run):
    st = planmod.CutadaptConfig()
    for k, v in flags.items():
        setattr(st, k, v)
    st.select_rule = rule
    tp = util.compile_plan(BUILDIN_ADAPTERS[name], st, True)
    batch = synth.generate_pairs(N, 150, BUILDIN_ADAPTERS[name], seed=SEED, poly_fraction=0.05, art5_fraction=0.01,
                                 indel_frac=0.1)
    r1, r2 = run(tp, batch)
    return {"scheme": name, "flags": flags, "rule": rule, "n": N, "seed": SEED,
            "crc_r1": zlib.crc32(r1.tobytes()), "crc_r2": zlib.crc32(r2.tobytes())}


def oracle_results(tp, batch):
    (r1, _, _), (r2, _, _) = util.oracle_run(tp, batch, threads=8)
    return r1, r2


if __name__ == "__main__":
    out = [crc_case(n, f, r, oracle_results) for n, f, r in CASES]
    (ROOT / "tests" / "golden" / "synth_results_crc.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))
