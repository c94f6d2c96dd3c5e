#!/usr/bin/env python3
"""Text-path soak on a GPU box (not collected by pytest): the chain presets through cs_text_* on fresh synthetic reads with
seeds nobody looked at -- ragged lengths, odd batch sizes, plain and device-compressed output, names with and without
mate suffixes -- against the oracle's results formatted by the record logic, byte for byte, until the time budget is spent.

    python tests/soak_text.py [seconds] [first_seed]
"""
import gzip
import random
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np  # noqa: E402

import test_gpu_text as tt  # noqa: E402
import util  # noqa: E402
from cutseq_amd import plan as planmod, synth, textpath  # noqa: E402
from cutseq_amd.common import BUILDIN_ADAPTERS  # noqa: E402
from cutseq_amd.engine import TrimEngine  # noqa: E402
from test_oracle import CHAIN_CASES  # noqa: E402


def bins_case(seed: int) -> None:
    """One demultiplexing batch through the text path (one route per barcode), expected from the array API's results and
    barcode indices (held to the oracle by tests/test_gpu_demux.py) formatted by the record logic."""
    from cutseq_amd import abi
    import hostfmt
    from cutseq_amd.common import BarcodeConfig
    from test_gpu_demux import barcode_set, plant_barcodes, scheme_with
    rng = random.Random(seed)
    length = rng.choice([6, 8, 8, 10, 12])
    count = rng.choice([2, 5, 24, 60])
    paired = rng.random() < 0.7
    compress = rng.random() < 0.4
    codes = barcode_set(rng, count, length, 3 if length <= 6 else 4)
    st = planmod.CutadaptConfig()
    st.trim_polyA = rng.random() < 0.5
    st.min_length = rng.choice([20, 40])
    st.demux_barcodes = codes
    n = rng.choice([1, 64, 257, 1000, 9000])
    batch = synth.generate_pairs(n, rng.choice([100, 150]), scheme_with(codes[0]), seed=seed, adapter_fraction=0.5, single_end=not paired)
    plant_barcodes(rng, batch, codes, length)
    tp = (planmod.compile_paired if paired else planmod.compile_single)(BarcodeConfig(scheme_with(codes[0])), st)
    names1 = [f"B{seed}:{i} 1:N:0:X".encode() for i in range(n)]
    names2 = [f"B{seed}:{i} 2:N:0:X".encode() for i in range(n)]
    text1 = tt.fastq_text(names1, batch.seq1, batch.qual1, batch.len1)
    text2 = tt.fastq_text(names2, batch.seq2, batch.qual2, batch.len2) if paired else None
    bc = np.empty(n, dtype=np.uint8)
    with TrimEngine(tp, device=0, slots=1, max_reads=n, max_stride=batch.stride) as eng:
        r1, cap2, r2 = eng.submit(0, batch.seq1, batch.qual1, batch.len1, batch.seq2, batch.qual2, batch.len2, bc=bc)
        eng.wait(0)
    want = [[b"", b""] for _ in range(3 + count)]
    want_counts = [0] * (3 + count)
    for i in range(n):
        n1 = int(batch.len1[i])
        if paired:
            n2 = int(batch.len2[i])
            route, rec1, rec2 = hostfmt.format_pair(names1[i], batch.seq1[i, :n1].tobytes(), batch.qual1[i, :n1].tobytes(), r1[i],
                                                    names2[i], batch.seq2[i, :n2].tobytes(), batch.qual2[i, :n2].tobytes(), r2[i], tp)
        else:
            route, rec1 = hostfmt.format_single(names1[i], batch.seq1[i, :n1].tobytes(), batch.qual1[i, :n1].tobytes(), r1[i],
                                                cap2[i] if cap2 is not None else None, tp)
            rec2 = b""
        if route == 0:
            assert bc[i] != abi.CS_DEMUX_NONE, seed
            route = 3 + int(bc[i])
        want[route][0] += rec1
        want[route][1] += rec2
        want_counts[route] += 1
    short_rows = rng.random() < 0.15 and n <= 1000
    with TrimEngine(tp, device=0, slots=0) as eng:
        with textpath.TextEngine(eng, slots=2, max_text_bytes=max(len(text1), len(text2 or b"")) + 1024, max_records=n,
                                 stride=64 if short_rows else batch.stride, compress=compress, bins=count) as te:
            got, counts = te.run(text1, n, text2, slot=rng.randrange(2))
    assert counts == want_counts, (seed, counts, want_counts)
    for route in range(3 + count):
        for m in range(2 if paired else 1):
            data = got[route][m]
            if compress and data:
                data = gzip.decompress(data)
            assert data == want[route][m], (seed, route, m)


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "bins":  # python tests/soak_text.py seconds first_seed bins
        budget, seed, t0, done = float(sys.argv[1]), int(sys.argv[2]), time.time(), 0
        while time.time() - t0 < budget:
            bins_case(seed)
            seed += 1
            done += 1
            if done % 25 == 0:
                print(f"{time.time() - t0:6.0f} s  seed {seed}  demultiplexing batches {done}", flush=True)
        print(f"text soak (bins) ok: {done} batches, seeds up to {seed}, {time.time() - t0:.0f} s")
        return
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
    cases = [c for c in CHAIN_CASES if "shortcut" not in c[1]]
    t0 = time.time()
    done = 0
    while time.time() - t0 < budget:
        rng = random.Random(seed)
        name, flags, paired = cases[rng.randrange(len(cases))]
        scheme = BUILDIN_ADAPTERS.get(name, name)
        st = planmod.CutadaptConfig()
        for k, v in flags.items():
            setattr(st, k, v)
        st.min_length = rng.choice([0, 20, 60])
        read_len = rng.choice([36, 75, 100, 150, 151, 250])
        n = rng.choice([1, 63, 64, 65, 257, 3000, 20_000])
        batch = synth.generate_pairs(n, read_len, scheme, seed=seed, single_end=not paired,
                                     poly_fraction=rng.choice([0.02, 0.2]), art5_fraction=rng.choice([0.001, 0.05]),
                                     adapter_fraction=rng.choice([0.2, 0.6]))
        nrng = np.random.default_rng(seed)
        cut = nrng.random(n) < 0.15
        batch.len1[cut] = nrng.integers(0, read_len, size=int(cut.sum())).astype(np.uint16)
        if paired:
            batch.len2[cut] = nrng.integers(0, read_len, size=int(cut.sum())).astype(np.uint16)
        tp = util.compile_plan(scheme, st, paired, untrimmed_requested="INLINE" in name)
        style = rng.randrange(3)
        names1 = [(f"S{seed}:{i}/1" if style == 0 else f"S{seed}:{i} 1:N:0:X" if style == 1 else f"S{seed}.{i}.1").encode() for i in range(n)]
        names2 = [(f"S{seed}:{i}/2" if style == 0 else f"S{seed}:{i} 2:N:0:X" if style == 1 else f"S{seed}.{i}.2").encode() for i in range(n)] if paired else None
        want, want_counts = tt.expected_streams(tp, batch, names1, names2)
        eol = rng.choice([b"\n", b"\r\n"])
        # (a file that ends in an EMPTY quality line without its line end cannot be told from a truncated record: keep the
        # final line end when the last read is empty)
        text1 = tt.fastq_text(names1, batch.seq1, batch.qual1, batch.len1, eol, rng.random() < 0.5 or int(batch.len1[-1]) == 0)
        text2 = tt.fastq_text(names2, batch.seq2, batch.qual2, batch.len2, eol, True) if paired else None
        compress = rng.random() < 0.4
        with TrimEngine(tp, device=0, slots=0) as eng:
            with textpath.TextEngine(eng, slots=2, max_text_bytes=max(len(text1), len(text2 or b""), 1) + 1024, max_records=max(n, 1),
                                     stride=rng.choice([batch.stride, batch.stride + 8, 64 if read_len > 64 else batch.stride]),
                                     compress=compress) as te:
                got, counts = te.run(text1, n, text2, slot=rng.randrange(2))
        assert counts == want_counts, (seed, name, counts, want_counts)
        for route in range(3):
            for m in range(2 if paired else 1):
                data = got[route][m]
                if compress and data:
                    data = gzip.decompress(data)
                assert data == want[route][m], (seed, name, route, m)
        done += 1
        seed += 1
        if done % 50 == 0:
            print(f"{time.time() - t0:6.0f} s  seed {seed}  batches {done}", flush=True)
    print(f"text soak ok: {done} batches, seeds up to {seed}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
