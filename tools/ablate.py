#!/usr/bin/env python3
"""GPU ablation: time plan variants of the two kernels on one resident batch (kernel-only)."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from cutseq_amd import abi, plan as planmod, synth
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig
from cutseq_amd.engine import TrimEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dev = torch.device("cuda", 0)
import os
batch = synth.generate_pairs(n, 150, **({"adapter_fraction": float(os.environ["CS_ADAPTER_FRACTION"]), "partial_fraction": float(os.environ.get("CS_PARTIAL_FRACTION", "0.09"))} if "CS_ADAPTER_FRACTION" in os.environ else {}))
up = lambda a: torch.from_numpy(a).to(dev)
d = dict(seq1=up(batch.seq1), qual1=up(batch.qual1), len1=up(batch.len1.view(np.int16)),
         seq2=up(batch.seq2), qual2=up(batch.qual2), len2=up(batch.len2.view(np.int16)))
out1 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
out2 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
r1 = abi.cs_reads(d["seq1"].data_ptr(), d["qual1"].data_ptr(), d["len1"].data_ptr(), out1.data_ptr(), None)
r2 = abi.cs_reads(d["seq2"].data_ptr(), d["qual2"].data_ptr(), d["len2"].data_ptr(), out2.data_ptr(), None)
bc = BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"])


def full(polyA=True, use_filter=True):
    st = planmod.CutadaptConfig()
    st.trim_polyA = polyA
    tp = planmod.compile_paired(bc, st)
    tp.use_filter = use_filter
    return tp


def keep(tp, pred):
    tp.r1.ops = [o for o in tp.r1.ops if pred(o)]
    tp.r2.ops = [o for o in tp.r2.ops if pred(o)]
    return tp


A, Cu, Q = planmod.AdapterOp, planmod.CutOp, planmod.QTrimOp
variants = {
    "full": full(),
    "full_nofilter": full(use_filter=False),
    "no_polyA": full(polyA=False),
    "no_qtrim": keep(full(), lambda o: not isinstance(o, Q)),
    "no_5prime": keep(full(), lambda o: not (isinstance(o, A) and o.rightmost)),
    "no_3prime": keep(full(), lambda o: not (isinstance(o, A) and o.kind_name == "BackAdapter")),
    "no_cuts": keep(full(), lambda o: not isinstance(o, Cu)),
    "only_5prime": keep(full(), lambda o: isinstance(o, A) and o.rightmost),
    "only_3prime": keep(full(), lambda o: isinstance(o, A) and o.kind_name == "BackAdapter"),
    "only_3prime_mo10": None,
    "only_5prime_mo3": None,
    "only_poly": keep(full(), lambda o: isinstance(o, A) and o.kind_name.startswith("NonInternal")),
    "only_cuts": keep(full(), lambda o: isinstance(o, Cu)),
    "only_qtrim": keep(full(), lambda o: isinstance(o, Q)),
}
def with_overlap(tp, mo):
    for ch in (tp.r1, tp.r2):
        for o in ch.ops:
            o.min_overlap = mo
    return tp


variants["only_3prime_mo10"] = with_overlap(keep(full(), lambda o: isinstance(o, A) and o.kind_name == "BackAdapter"), 10)
variants["only_5prime_mo3"] = with_overlap(keep(full(), lambda o: isinstance(o, A) and o.rightmost), 3)
stream = torch.cuda.Stream(device=dev)
sh = C.c_void_p(stream.cuda_stream)
for name, tp in variants.items():
    eng = TrimEngine(tp, device=0, slots=0)
    for _ in range(2):
        eng.trim_device(r1, r2, n, batch.stride, stream=sh)
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        eng.trim_device(r1, r2, n, batch.stride, stream=sh)
        ms.append(eng.last_kernel_ms())
        split = eng.last_kernel_split_ms()
    s1, s2 = eng.stats()
    frac = (s1.n_exact_dp + s2.n_exact_dp) / max(1, s1.n_reads + s2.n_reads)
    t = float(np.median(ms))
    print(f"{name:16s} {t:9.3f} ms  {n / t / 1e3:9.1f} M pairs/s  {616 * n / t / 1e6:8.1f} GB/s  exact-DP/read {frac:.3f}  scan {split[0]:.3f} resolve {split[1]:.3f}", flush=True)
    eng.close()
