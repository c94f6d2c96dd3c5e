#!/usr/bin/env python3
"""Kernel time of the TAKARAV3 chain at other read lengths (the LDS tile, and with it the blocks per CU, follow the
stride):  CUTSEQ_HIP_LIB=... python tools/readlen_ab.py [pairs]"""
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
st = planmod.CutadaptConfig()
st.trim_polyA = True
tp = planmod.compile_paired(BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"]), st)
for read_len in (75, 100, 150, 200, 250, 300):
    batch = synth.generate_pairs(n, read_len)
    up = lambda a: torch.from_numpy(a).to(dev)
    d = [up(batch.seq1), up(batch.qual1), up(batch.len1.view(np.int16)), up(batch.seq2), up(batch.qual2), up(batch.len2.view(np.int16))]
    out1 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
    out2 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
    r1 = abi.cs_reads(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out1.data_ptr(), None)
    r2 = abi.cs_reads(d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(), out2.data_ptr(), None)
    stream = torch.cuda.Stream(device=dev)
    sh = C.c_void_p(stream.cuda_stream)
    with TrimEngine(tp, device=0, slots=0) as eng:
        for _ in range(3):
            eng.trim_device(r1, r2, n, batch.stride, stream=sh)
        torch.cuda.synchronize()
        ms = []
        for _ in range(7):
            eng.trim_device(r1, r2, n, batch.stride, stream=sh)
            torch.cuda.synchronize()
            ms.append(eng.last_kernel_split_ms())
        scan = float(np.median([m[0] for m in ms]))
        res = float(np.median([m[1] for m in ms]))
    print(f"read length {read_len:3d}  stride {batch.stride:3d}  scan {scan:.3f} ms  resolve {res:.3f} ms  {n / (scan + res) / 1e3:8.1f} M pairs/s (one stream)", flush=True)
    del d, out1, out2
