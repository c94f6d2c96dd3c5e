#!/usr/bin/env python3
"""Tier T (pinned host -> H2D -> kernel -> D2H) and tier E (gz FASTQ -> gz FASTQ) rates, SURVEY.md 8d.
These are NOT the headline (bench.py reports the HBM-resident kernel rate); they go into DESIGN.md."""
import ctypes as C
import json
import subprocess
import sys
import tempfile
import os
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np

from cutseq_amd import capi, plan as planmod, synth
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig
from cutseq_amd.engine import TrimEngine


def pinned_like(L, arr):
    # (TIER_T_HUGE=1: page-locked memory on huge pages, cs_alloc_pinned_huge -- same 53 GB/s, see include/cutseq_hip.h)
    p = (L.cs_alloc_pinned_huge if os.environ.get("TIER_T_HUGE") == "1" else L.cs_alloc_pinned)(arr.nbytes)
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(arr.nbytes,)).view(arr.dtype).reshape(arr.shape)
    out[...] = arr
    return out


def tier_t(n=1 << 20, rounds=8):
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    tp = planmod.compile_paired(BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"]), st)
    L = capi.load()
    b = synth.generate_pairs(n, 150)
    arrs = [pinned_like(L, a) for a in (b.seq1, b.qual1, b.len1, b.seq2, b.qual2, b.len2)]
    with TrimEngine(tp, device=0, slots=2, max_reads=n, max_stride=b.stride) as eng:
        for s in (0, 1):
            eng.submit(s, *arrs)
            eng.wait(s)
        t0 = time.perf_counter()
        for r in range(rounds):  # two slots in flight: copies of one overlap the kernel of the other
            eng.submit(r & 1, *arrs)
            if r:
                eng.wait((r - 1) & 1)
        eng.wait((rounds - 1) & 1)
        dt = time.perf_counter() - t0
    return {"pairs": n * rounds, "seconds": dt, "M_pairs_per_s": n * rounds / dt / 1e6,
            "host_GBps": n * rounds * (4 * 152 + 4 + 16) / dt / 1e9}


def tier_e(n=2_000_000):
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        subprocess.run([sys.executable, str(ROOT / "tools" / "make_fastq.py"), str(n), f"{d}/syn"], check=True)
        t0 = time.perf_counter()
        subprocess.run([sys.executable, "-m", "cutseq_amd.run", f"{d}/syn_R1.fastq.gz", f"{d}/syn_R2.fastq.gz",
                        "-A", "TAKARAV3", "--trim-polyA", "-O", f"{d}/out"], check=True, cwd=str(ROOT),
                       stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        out = {"pairs": n, "seconds": dt, "M_pairs_per_s": n / dt / 1e6}
        # the same run on uncompressed text in and out (no codec on either side)
        import gzip
        import shutil
        for m in (1, 2):
            with gzip.open(f"{d}/syn_R{m}.fastq.gz", "rb") as src, open(f"{d}/plain_R{m}.fastq", "wb") as dst:
                shutil.copyfileobj(src, dst, 1 << 24)
        t0 = time.perf_counter()
        subprocess.run([sys.executable, "-m", "cutseq_amd.run", f"{d}/plain_R1.fastq", f"{d}/plain_R2.fastq",
                        "-A", "TAKARAV3", "--trim-polyA", "-o", f"{d}/o1.fastq", f"{d}/o2.fastq",
                        "-s", f"{d}/s1.fastq", f"{d}/s2.fastq"], check=True, cwd=str(ROOT), stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        out["plain_text"] = {"seconds": dt, "M_pairs_per_s": n / dt / 1e6}
    return out


if __name__ == "__main__":
    print(json.dumps({"tier_T": tier_t(), "tier_E": tier_e()}))
