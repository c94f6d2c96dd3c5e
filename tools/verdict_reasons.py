#!/usr/bin/env python3
"""Diagnostic: why the leading walk's verdict sends reads to the exact DP (library built with -DCS_EXP_REASON).
    CUTSEQ_HIP_LIB=tools/ab/reason.so python tools/verdict_reasons.py [pairs]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from cutseq_amd import plan as planmod, synth
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig
from cutseq_amd.engine import TrimEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
st = planmod.CutadaptConfig()
st.trim_polyA = True
tp = planmod.compile_paired(BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"]), st)
b = synth.generate_pairs(n, 150)
BASE = ["exact hit, overlap with an older hit unclear", "outside the settle block (read start / short read)",
        "no attempt possible (conditions of either cell fail)", "end row tried, not substitution-only",
        "column cell tried, not substitution-only, no hit one column on", "hit one column on tried, failed"]
with TrimEngine(tp, device=0, slots=1, max_reads=n, max_stride=b.stride) as eng:
    eng.trim(b.seq1, b.qual1, b.len1, b.seq2, b.qual2, b.len2)
    for mate, s in enumerate(eng.stats(), 1):
        print(f"mate {mate}: {s.n_reads} reads, exact-DP requests {s.n_exact_dp} ({100 * s.n_exact_dp / s.n_reads:.2f} %)")
        for q in range(24):
            w = int(s.op_matched[8 + q % 12])
            c = (w >> 32) if q >= 12 else (w & 0xffffffff)
            if c:
                grp = q // 6
                print(f"   {100 * c / s.n_reads:5.2f} %  {BASE[q % 6]:70s} candidates: {'columns ' if grp & 1 else ''}{'end rows' if grp & 2 else ''}")
