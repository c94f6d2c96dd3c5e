#!/usr/bin/env python3
"""Where does the finish kernel of the split form spend its time?  Serial form (scan, then finish + resolve, one
stream) on plan variants, CUTSEQ_LEAN=0 / 1 in separate processes (the engine reads the knob when it is created)."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def child():
    import numpy as np
    import torch
    from cutseq_amd import abi, plan as planmod, workloads
    from cutseq_amd.engine import TrimEngine
    n = 4_000_000
    dev = torch.device("cuda", 0)
    batch = workloads.make_batch("config3", n)
    up = lambda a: torch.from_numpy(a).to(dev)
    d = dict(seq1=up(batch.seq1), qual1=up(batch.qual1), len1=up(batch.len1.view(np.int16)),
             seq2=up(batch.seq2), qual2=up(batch.qual2), len2=up(batch.len2.view(np.int16)))
    o1 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
    o2 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
    r1 = abi.cs_reads(d["seq1"].data_ptr(), d["qual1"].data_ptr(), d["len1"].data_ptr(), o1.data_ptr(), None, None)
    r2 = abi.cs_reads(d["seq2"].data_ptr(), d["qual2"].data_ptr(), d["len2"].data_ptr(), o2.data_ptr(), None, None)

    def keep(tp, pred):
        tp.r1.ops = [o for o in tp.r1.ops if pred(o)]
        tp.r2.ops = [o for o in tp.r2.ops if pred(o)]
        return tp

    A, Cu, Q = planmod.AdapterOp, planmod.CutOp, planmod.QTrimOp
    poly = lambda o: isinstance(o, A) and o.kind_name.startswith("NonInternal")
    variants = {
        "full": workloads.make_plan("config3"),
        "no_poly": keep(workloads.make_plan("config3"), lambda o: not poly(o)),
        "no_qtrim": keep(workloads.make_plan("config3"), lambda o: not isinstance(o, Q)),
        "cuts_only_tail": keep(workloads.make_plan("config3"), lambda o: not poly(o) and not isinstance(o, Q)),
    }
    stream = torch.cuda.Stream(device=dev)
    sh = C.c_void_p(stream.cuda_stream)
    for name, tp in variants.items():
        eng = TrimEngine(tp, device=0, slots=0)
        for pipelined in (False, True):
            for _ in range(3):
                eng.trim_device(r1, r2, n, batch.stride, stream=sh, pipelined=pipelined)
            eng.join(sh)
            torch.cuda.synchronize()
            eng.kernel_time_totals(reset=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record()
            for _ in range(10):
                eng.trim_device(r1, r2, n, batch.stride, stream=sh, pipelined=pipelined)
            eng.join(sh)
            with torch.cuda.stream(stream):
                e1.record()
            torch.cuda.synchronize()
            calls, a, b = eng.kernel_time_totals()
            print(f"LEAN={os.environ.get('CUTSEQ_LEAN')} {name:16s} {'pipelined' if pipelined else 'serial   '} step {e0.elapsed_time(e1) / 10:6.3f} "
                  f"scan {a / calls:6.3f} finish+resolve {b / calls:6.3f}", flush=True)
        eng.close()


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        for lean in ("0", "1"):
            subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, CUTSEQ_LEAN=lean), check=False)
