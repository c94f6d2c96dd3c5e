#!/usr/bin/env python3
"""How much would a lean scan kernel (staging + the two Myers scans only) save?  Times, on one resident batch,
the full TAKARAV3 + poly-A plan against its two halves: the leading adapter ops alone and everything behind them
alone (kernel-only, serial form: scan and resolve kernel on one stream; and the pipelined bench form).
Run per library build (CUTSEQ_HIP_LIB): tools/lean_probe.sh compares scan kernels built for 5 / 6 / 7 waves per SIMD."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from cutseq_amd import abi, plan as planmod, workloads
from cutseq_amd.engine import TrimEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda", 0)
batch = workloads.make_batch("config3", n)
up = lambda a: torch.from_numpy(a).to(dev)
d = dict(seq1=up(batch.seq1), qual1=up(batch.qual1), len1=up(batch.len1.view(np.int16)),
         seq2=up(batch.seq2), qual2=up(batch.qual2), len2=up(batch.len2.view(np.int16)))
sets = []
for _ in range(3):
    o1 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
    o2 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
    sets.append((abi.cs_reads(d["seq1"].data_ptr(), d["qual1"].data_ptr(), d["len1"].data_ptr(), o1.data_ptr(), None, None),
                 abi.cs_reads(d["seq2"].data_ptr(), d["qual2"].data_ptr(), d["len2"].data_ptr(), o2.data_ptr(), None, None), o1, o2))


def keep(tp, pred):
    tp.r1.ops = [o for o in tp.r1.ops if pred(o)]
    tp.r2.ops = [o for o in tp.r2.ops if pred(o)]
    return tp


def lead(o):
    return isinstance(o, planmod.AdapterOp) and not o.kind_name.startswith("NonInternal")


variants = {
    "full": workloads.make_plan("config3"),
    "adapters_only": keep(workloads.make_plan("config3"), lead),
    "rest_only": keep(workloads.make_plan("config3"), lambda o: not lead(o)),
}
stream = torch.cuda.Stream(device=dev)
sh = C.c_void_p(stream.cuda_stream)
for name, tp in variants.items():
    eng = TrimEngine(tp, device=0, slots=0)
    for pipelined in (False, True):
        for i in range(3):
            eng.trim_device(sets[i % 3][0], sets[i % 3][1], n, batch.stride, stream=sh, pipelined=pipelined)
        eng.join(sh)
        torch.cuda.synchronize()
        eng.kernel_time_totals(reset=True)
        steps = 12
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            e0.record()
        for i in range(steps):
            eng.trim_device(sets[i % 3][0], sets[i % 3][1], n, batch.stride, stream=sh, pipelined=pipelined)
        eng.join(sh)
        with torch.cuda.stream(stream):
            e1.record()
        torch.cuda.synchronize()
        calls, scan_ms, res_ms = eng.kernel_time_totals()
        step_ms = e0.elapsed_time(e1) / steps
        print(f"{name:14s} {'pipelined' if pipelined else 'serial   '} step {step_ms:7.3f} ms  scan {scan_ms / calls:7.3f}  "
              f"resolve {res_ms / calls:7.3f}  {n / step_ms / 1e3:8.1f} M pairs/s", flush=True)
    eng.close()
