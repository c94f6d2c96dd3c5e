"""The synthetic generator's device form (csrc/synth_device.hip) against its host form (csrc/cutseq_host.c).

bench.py fills its resident batch on the device (no 61 GB host buffer, no pageable copy, start-up
independent of the number of ranks); what it times must be the very reads the parity tests hold to the oracle, so the
two forms have to write the same bytes for the same global pair indices -- any first index, any split of the range.
"""
import numpy as np
import pytest
import torch

from cutseq_amd import synth, workloads

pytestmark = pytest.mark.gpu


def device_batch(workload, n, first, dev, stream=None):
    paired = workloads.is_paired(workload)
    stride = synth._stride_for(workloads.READ_LEN)
    mk = lambda: (torch.full((n, stride), 0xAA, dtype=torch.uint8, device=dev),  # noqa: E731
                  torch.full((n, stride), 0xAA, dtype=torch.uint8, device=dev),
                  torch.full((n,), -1, dtype=torch.int16, device=dev))
    a = mk()
    b = mk() if paired else (None, None, None)
    ptrs = [t.data_ptr() if t is not None else None for t in a + b]
    assert workloads.fill_device(workload, n, ptrs, first_index=first, stream=stream) == stride
    torch.cuda.synchronize(dev)
    return [t.cpu().numpy() if t is not None else None for t in a + b]


@pytest.mark.parametrize("workload", ["config3", "config4", "config2"])
@pytest.mark.parametrize("first", [0, 123_456_789, (7 << 32) + 99])
def test_device_generator_writes_the_host_generators_bytes(workload, first):
    dev = torch.device("cuda:0")
    n = 1_000_000
    got = device_batch(workload, n, first, dev)
    want = workloads.make_batch(workload, n, first_index=first)
    for name, g in zip(("seq1", "qual1", "len1", "seq2", "qual2", "len2"), got):
        w = getattr(want, name)
        if w is None:
            assert g is None
            continue
        g = g.view(np.uint16) if name.startswith("len") else g
        if not np.array_equal(g, w):
            bad = int(np.flatnonzero((g != w).reshape(n, -1).any(axis=1))[0])
            raise AssertionError(f"{workload} first_index {first}: {name} differs first at pair {bad}: "
                                 f"{bytes(g[bad]) if g.ndim > 1 else g[bad]!r} != {bytes(w[bad]) if w.ndim > 1 else w[bad]!r}")


def test_device_generator_any_split_and_odd_shapes():
    """Two launches over adjacent index ranges equal one launch (ragged last block); other read lengths, rates that
    make every branch common (indels in every adapter, artefacts, poly stretches, degrading tails), '+' strand."""
    dev = torch.device("cuda:0")
    whole = device_batch("config3", 100_001, 5_000, dev)
    lo = device_batch("config3", 70_000, 5_000, dev)
    hi = device_batch("config3", 30_001, 75_000, dev)
    for w, a, b in zip(whole, lo, hi):
        assert np.array_equal(w, np.concatenate([a, b]))
    for read_len, scheme in ((75, None), (151, "ACACGACGCTCTTCCGATCT(ATCACG)NNNNN>NNN(GGTTAA)AGATCGGAAGAGCACACGTC"),
                             (250, "AGTTCTACAGTCCGACGATCNNNNN-NNNNN(ATCACG)AGATCGGAAGAGCACACGTC"), (36, None)):
        n = 50_000
        kw = dict(adapter_fraction=0.5, partial_fraction=0.2, poly_fraction=0.5, art5_fraction=0.3, sub_rate=0.05,
                  indel_frac=0.9, n_rate=0.02, seed=42)
        want = synth.generate_pairs(n, read_len, scheme, first_index=77, **kw)
        stride = want.stride
        t = [torch.zeros((n, stride), dtype=torch.uint8, device=dev) for _ in range(4)]
        ln = [torch.zeros(n, dtype=torch.int16, device=dev) for _ in range(2)]
        synth.generate_pairs_device(n, [t[0].data_ptr(), t[1].data_ptr(), ln[0].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                                        ln[1].data_ptr()], read_len, scheme, first_index=77, **kw)
        torch.cuda.synchronize(dev)
        for g, w in zip((t[0], t[1], t[2], t[3]), (want.seq1, want.qual1, want.seq2, want.qual2)):
            assert np.array_equal(g.cpu().numpy(), w), (read_len, scheme)
        assert np.array_equal(ln[0].cpu().numpy().view(np.uint16), want.len1)
        assert np.array_equal(ln[1].cpu().numpy().view(np.uint16), want.len2)


def test_device_generator_refuses_what_it_cannot_reproduce():
    with pytest.raises(ValueError):
        synth.generate_pairs_device(10, [1, 1, 1, None, None, None], 150)  # paired batch without mate 2 arrays
    with pytest.raises(ValueError):
        synth.generate_pairs_device(10, [1, 1, 1, 1, 1, 1], 150, stride=150)  # stride not a multiple of 4
