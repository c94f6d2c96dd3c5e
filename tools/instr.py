import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from cutseq_amd import plan as planmod, synth
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig
from cutseq_amd.engine import TrimEngine
st = planmod.CutadaptConfig(); st.trim_polyA = True
tp = planmod.compile_paired(BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"]), st)
b = synth.generate_pairs(1_000_000, 150)
with TrimEngine(tp, device=0, slots=1, max_reads=b.n, max_stride=b.stride) as eng:
    eng.trim(b.seq1, b.qual1, b.len1, b.seq2, b.qual2, b.len2)
    s1, s2 = eng.stats()
for s in (s1, s2):
    steps, passes = s._reserved, s.op_matched[15]
    print("exact items", s.n_exact_dp, "passes", passes, "sum max-steps", steps, "avg steps/pass", steps / max(1, passes), "items/pass", s.n_exact_dp / max(1, passes))
