#!/usr/bin/env python3
"""Diagnostic: where a wave's wall-clock time goes, phase by phase.

  python tools/phase_timers.py build      # here: hipcc -DCS_PHASE_TIMERS -> tools/ab/libcutseq_hip_timers.so
  python tools/phase_timers.py [pairs]    # on the GPU box: run BASELINE config 3 once and print the split

The instrumented library reports cycles (in units of 64) through op_matched[16..23]; it is never
the product library.
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
WHICH = os.environ.get("CS_TIMERS_KERNEL", "1")  # 1 = scan kernel, 2 = resolve kernel
LIB = ROOT / "tools" / "ab" / f"libcutseq_hip_timers{WHICH}.so"
PHASES = ["stage tile (loads + recode + barrier)", "filter: column scan", "filter: end rows / short cuts / verdict",
          "exact DP: queue, pass set-up, end rows, result", "poly-A/T closed form", "quality trim (+ fixed cuts)",
          "results, statistics, barrier", "exact DP: strip step loop"]

if len(sys.argv) > 1 and sys.argv[1] == "build":
    LIB.parent.mkdir(exist_ok=True)
    src = ROOT / "cutseq_amd" / "csrc" / "cutseq_hip.hip"
    for which in ("1", "2"):
        lib = ROOT / "tools" / "ab" / f"libcutseq_hip_timers{which}.so"
        subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
                        f"-DCS_PHASE_TIMERS={which}", "-o", str(lib), str(src)], check=True, cwd=str(src.parent))
        print(lib)
    sys.exit(0)

os.environ["CUTSEQ_HIP_LIB"] = str(LIB)
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

from cutseq_amd import abi, plan as planmod, synth
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig
from cutseq_amd.engine import TrimEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dev = torch.device("cuda", 0)
batch = synth.generate_pairs(n, 150)
up = lambda a: torch.from_numpy(a).to(dev)
keep = [up(batch.seq1), up(batch.qual1), up(batch.len1.view(np.int16)), up(batch.seq2), up(batch.qual2),
        up(batch.len2.view(np.int16))]
out1 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
out2 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
r1 = abi.cs_reads(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), out1.data_ptr(), None)
r2 = abi.cs_reads(keep[3].data_ptr(), keep[4].data_ptr(), keep[5].data_ptr(), out2.data_ptr(), None)
st = planmod.CutadaptConfig()
st.trim_polyA = True
tp = planmod.compile_paired(BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"]), st)
VARIANT = os.environ.get("CS_PLAN", "full")  # full | only_5prime | only_3prime | no_5prime
if VARIANT != "full":
    A = planmod.AdapterOp
    pred = {"only_5prime": lambda o: isinstance(o, A) and o.rightmost,
            "only_3prime": lambda o: isinstance(o, A) and o.kind_name == "BackAdapter",
            "no_5prime": lambda o: not (isinstance(o, A) and o.rightmost)}[VARIANT]
    tp.r1.ops = [o for o in tp.r1.ops if pred(o)]
    tp.r2.ops = [o for o in tp.r2.ops if pred(o)]
eng = TrimEngine(tp, device=0, slots=0)
eng.trim_device(r1, r2, n, batch.stride)
torch.cuda.synchronize()
eng.stats(reset=True)
eng.trim_device(r1, r2, n, batch.stride)
torch.cuda.synchronize()
ms = eng.last_kernel_ms()
stats = eng.stats()
print(f"plan {VARIANT}: {n} pairs, both kernels {ms:.3f} ms (instrumented build), reporting: {'scan' if WHICH == '1' else 'resolve'} kernel")
for mate, s in enumerate(stats):
    t = np.array([int(s.op_matched[16 + i]) for i in range(8)], dtype=np.float64) * 64
    tot = t.sum()
    print(f"mate {mate + 1}: {tot / 1e9:.3f} G wave-cycles")
    for i, name in enumerate(PHASES):
        print(f"  {100 * t[i] / tot:5.1f} %   {name}")
    passes, steps, items = (int(s.op_matched[13 + i]) for i in range(3))
    print(f"  strip passes {passes} ({passes / (n / 64):.2f} per tile), {steps / max(passes, 1):.1f} steps and "
          f"{items / max(passes, 1):.1f} survivors per pass")
eng.close()
