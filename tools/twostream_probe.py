#!/usr/bin/env python3
"""The bench loop (one engine, cs_trim_device_pipelined, three result sets) with the steps handed to S caller streams
in turn: with S > 1 the scan kernel of step i + 1 may start on the SIMD slots the draining blocks of step i's scan kernel
give back, instead of behind its last tile.
    python3 tools/twostream_probe.py [pairs] [steps]
"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cutseq_amd import abi, workloads  # noqa: E402
from cutseq_amd.engine import TrimEngine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    dev = torch.device("cuda:0")
    tp = workloads.make_plan("config3", True)
    batch = workloads.make_batch("config3", n, 0, 8)
    up = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    d = {k: up(getattr(batch, k)) for k in ("seq1", "qual1", "seq2", "qual2")}
    d["len1"], d["len2"] = up(batch.len1.view(np.int16)), up(batch.len2.view(np.int16))
    eng = TrimEngine(tp, device=0, slots=0)
    sets = []
    for _ in range(3):
        o1 = torch.zeros((n, 8), dtype=torch.uint8, device=dev)
        o2 = torch.zeros((n, 8), dtype=torch.uint8, device=dev)
        sets.append((abi.cs_reads(d["seq1"].data_ptr(), d["qual1"].data_ptr(), d["len1"].data_ptr(), o1.data_ptr(), None, None),
                     abi.cs_reads(d["seq2"].data_ptr(), d["qual2"].data_ptr(), d["len2"].data_ptr(), o2.data_ptr(), None, None), o1, o2))
    ref = None
    for n_streams in (1, 2, 3, 1, 2):
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
        handles = [C.c_void_p(s.cuda_stream) for s in streams]
        for rep in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(steps):
                eng.trim_device(sets[i % 3][0], sets[i % 3][1], n, batch.stride, stream=handles[i % n_streams], pipelined=True)
            for h in handles:
                eng.join(h)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            print(f"{n_streams} caller stream(s): {steps * n / dt / 1e6:.1f} M pairs/s  ({dt / steps * 1e3:.3f} ms/step)", flush=True)
        got = [sets[0][2].clone(), sets[0][3].clone()]
        if ref is None:
            ref = got
        print("  results identical to the one-stream run:", all(torch.equal(a, b) for a, b in zip(ref, got)), flush=True)
    ms = eng.kernel_time_totals() if hasattr(eng, "kernel_time_totals") else None
    print("kernel time totals:", ms)


if __name__ == "__main__":
    main()
