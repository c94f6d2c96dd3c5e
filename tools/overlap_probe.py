#!/usr/bin/env python3
"""Does the resolve kernel of one batch hide under the scan kernel of the next?

Config 3 of bench.py on E engines (each with its own stream, deferral queue and result arrays, the inputs
shared), steps handed to the engines in turn.  E=1 is bench.py's own loop.  Prints M pairs/s per setting.
    python3 tools/overlap_probe.py [pairs] [steps]
"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cutseq_amd import abi, synth  # noqa: E402
import bench  # noqa: E402
from cutseq_amd.engine import TrimEngine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    tp = bench.make_plan("config3", True)
    batch = synth.generate_pairs(n, 150)
    up = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    d = {k: up(getattr(batch, k)) for k in ("seq1", "qual1", "seq2", "qual2")}
    d["len1"], d["len2"] = up(batch.len1.view(np.int16)), up(batch.len2.view(np.int16))
    for n_eng in (1, 2, 3):
        engines, streams, reads, keep = [], [], [], []
        for _ in range(n_eng):
            o1 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
            o2 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
            keep += [o1, o2]
            reads.append((abi.cs_reads(d["seq1"].data_ptr(), d["qual1"].data_ptr(), d["len1"].data_ptr(), o1.data_ptr(), None, None),
                          abi.cs_reads(d["seq2"].data_ptr(), d["qual2"].data_ptr(), d["len2"].data_ptr(), o2.data_ptr(), None, None)))
            engines.append(TrimEngine(tp, device=0, slots=0))
            s = torch.cuda.Stream(device=dev)
            streams.append((s, C.c_void_p(s.cuda_stream)))
        for rep in range(2):
            for i in range(2 * n_eng):
                e = i % n_eng
                engines[e].trim_device(reads[e][0], reads[e][1], n, batch.stride, stream=streams[e][1])
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(steps):
                e = i % n_eng
                engines[e].trim_device(reads[e][0], reads[e][1], n, batch.stride, stream=streams[e][1])
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            print(f"engines {n_eng}: {steps * n / dt / 1e6:.1f} M pairs/s  ({dt / steps * 1e3:.3f} ms/step)", flush=True)
        same = all(torch.equal(keep[0], keep[2 * e]) and torch.equal(keep[1], keep[2 * e + 1]) for e in range(n_eng))
        print("  results identical across engines:", same, flush=True)
        del engines
    # the engine's own pipeline: one stream, resolve kernels on the engine's resolve stream (what bench.py times)
    eng = TrimEngine(tp, device=0, slots=0)
    st = torch.cuda.Stream(device=dev)
    sh = C.c_void_p(st.cuda_stream)
    sets = []
    for _ in range(3):
        o1 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
        o2 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
        sets.append((abi.cs_reads(d["seq1"].data_ptr(), d["qual1"].data_ptr(), d["len1"].data_ptr(), o1.data_ptr(), None, None),
                     abi.cs_reads(d["seq2"].data_ptr(), d["qual2"].data_ptr(), d["len2"].data_ptr(), o2.data_ptr(), None, None), o1, o2))
    for rep in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            eng.trim_device(sets[i % 3][0], sets[i % 3][1], n, batch.stride, stream=sh, pipelined=True)
        eng.join(sh)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        print(f"pipelined, one engine: {steps * n / dt / 1e6:.1f} M pairs/s  ({dt / steps * 1e3:.3f} ms/step)", flush=True)


if __name__ == "__main__":
    main()
