"""The device-pointer entry points (cs_trim_device, cs_trim_device_pipelined + cs_join) against the oracle.

Batches already resident in HBM, as bench.py drives them: joined calls, pipelined calls on one stream with
one set of result arrays per call in flight (more calls than the engine has lanes, so lanes get reused),
and joined calls alternating between two streams on one engine.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from cutseq_amd import abi, plan as planmod, synth
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig
from cutseq_amd.engine import TrimEngine

import util

pytestmark = pytest.mark.gpu


def takara_plan():
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    return planmod.compile_paired(BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"]), st)


class Resident:
    """One synthetic batch on the device + its own result arrays."""

    def __init__(self, batch, dev):
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        self.batch = batch
        self.t = [up(batch.seq1), up(batch.qual1), up(batch.len1.view(np.int16)),
                  up(batch.seq2), up(batch.qual2), up(batch.len2.view(np.int16))]
        self.out1 = torch.zeros((batch.n, 8), dtype=torch.uint8, device=dev)
        self.out2 = torch.zeros((batch.n, 8), dtype=torch.uint8, device=dev)
        p = [x.data_ptr() for x in self.t]
        self.r1 = abi.cs_reads(p[0], p[1], p[2], self.out1.data_ptr(), None, None)
        self.r2 = abi.cs_reads(p[3], p[4], p[5], self.out2.data_ptr(), None, None)

    def results(self):
        return (self.out1.cpu().numpy().view(abi.RESULT_DTYPE).reshape(-1),
                self.out2.cpu().numpy().view(abi.RESULT_DTYPE).reshape(-1))


def expect(tp, batch):
    (o1, _, st1), (o2, _, st2) = util.oracle_run(tp, batch, threads=8)
    return o1, o2, st1, st2


@pytest.mark.parametrize("form", ["joined", "pipelined", "two_streams"])
def test_resident_batches(form):
    dev = torch.device("cuda:0")
    tp = takara_plan()
    # different sizes: the queue of a reused lane has to grow, the last batch is a ragged tile
    sizes = [30_000, 50_000, 20_001, 64, 70_000, 1, 40_000]
    batches = [synth.generate_pairs(n, 150, first_index=1000 * i) for i, n in enumerate(sizes)]
    res = [Resident(b, dev) for b in batches]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    handles = [C.c_void_p(s.cuda_stream) for s in streams]
    with TrimEngine(tp, device=0, slots=0) as eng:
        for i, r in enumerate(res):
            if form == "joined":
                eng.trim_device(r.r1, r.r2, r.batch.n, r.batch.stride, stream=handles[0])
            elif form == "pipelined":
                eng.trim_device(r.r1, r.r2, r.batch.n, r.batch.stride, stream=handles[0], pipelined=True)
            else:
                eng.trim_device(r.r1, r.r2, r.batch.n, r.batch.stride, stream=handles[i % 2])
        if form == "pipelined":
            eng.join(handles[0])
            streams[0].synchronize()  # results are complete in stream order behind the join
        else:
            for s in streams:
                s.synchronize()
        gst1, gst2 = eng.stats()
        scan_ms, resolve_ms = eng.last_kernel_split_ms()
        assert scan_ms > 0 and resolve_ms > 0
    tot = {}
    for r in res:
        o1, o2, st1, st2 = expect(tp, r.batch)
        g1, g2 = r.results()
        assert (g1 == o1).all() and (g2 == o2).all()
        for mate, st in ((1, st1), (2, st2)):
            for k, v in st.as_dict().items():
                if k in ("n_exact_dp", "n_refiltered"):
                    continue
                key = (mate, k)
                tot[key] = (np.asarray(tot[key]) + np.asarray(v)).tolist() if key in tot else v
    for mate, st in ((1, gst1), (2, gst2)):
        for k, v in st.as_dict().items():
            if k not in ("n_exact_dp", "n_refiltered"):
                assert np.array_equal(np.asarray(v), np.asarray(tot[(mate, k)])), (mate, k)


def test_single_end_pipelined_with_second_capture_and_timing_totals():
    """Single-end scheme with a UMI on either side (second capture array), pipelined calls, and the per-kernel
    time totals the bench reads: one entry per timed call, reset on request."""
    dev = torch.device("cuda:0")
    st = planmod.CutadaptConfig()
    tp = planmod.compile_single(BarcodeConfig("ACACGACGCTCTTCCGATCTNNNNNN>NNNNAGATCGGAAGAGCACACGTC"), st)
    assert tp.needs_cap2
    batches = [synth.generate_pairs(n, 150, scheme="ACACGACGCTCTTCCGATCTNNNNNN>NNNNAGATCGGAAGAGCACACGTC", single_end=True,
                                    first_index=77 * i) for i, n in enumerate([20_000, 33_333, 5])]
    stream = torch.cuda.Stream(device=dev)
    sh = C.c_void_p(stream.cuda_stream)
    keep = []
    with TrimEngine(tp, device=0, slots=0) as eng:
        assert eng.kernel_time_totals(reset=True)[0] == 0
        for b in batches:
            t = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (b.seq1, b.qual1, b.len1.view(np.int16))]
            out = torch.zeros((b.n, 8), dtype=torch.uint8, device=dev)
            cap2 = torch.zeros((b.n, 4), dtype=torch.uint8, device=dev)
            keep.append((t, out, cap2))
            r1 = abi.cs_reads(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), out.data_ptr(), cap2.data_ptr(), None)
            eng.trim_device(r1, None, b.n, b.stride, stream=sh, pipelined=True)
        eng.join(sh)
        stream.synchronize()
        calls, scan_ms, resolve_ms = eng.kernel_time_totals(reset=True)
        assert calls == len(batches) and scan_ms > 0 and resolve_ms > 0
        assert eng.kernel_time_totals()[0] == 0
    for b, (_, out, cap2) in zip(batches, keep):
        (o1, ocap2, _), _m2 = util.oracle_run(tp, b, threads=8)
        assert (out.cpu().numpy().view(abi.RESULT_DTYPE).reshape(-1) == o1).all()
        assert (cap2.cpu().numpy().view(abi.CAP2_DTYPE).reshape(-1) == ocap2).all()


def test_four_slots_in_flight_on_three_lanes():
    """The host-buffer path with more staging slots in flight than the engine has lanes: uploads and scan kernels
    on the engine stream, resolve kernels and downloads on the resolve stream, a lane reused while its slot's
    results are still coming back.  Every batch must equal the oracle, in whatever order the slots are synced."""
    tp = takara_plan()
    sizes = [9_000, 12_345, 64, 20_000, 1, 15_000, 7_777, 12_000, 3]
    batches = [synth.generate_pairs(n, 150, first_index=5000 * i + 11) for i, n in enumerate(sizes)]
    with TrimEngine(tp, device=0, slots=4, max_reads=max(sizes), max_stride=batches[0].stride) as eng:
        results = [None] * len(batches)
        in_flight = []
        for i, b in enumerate(batches):
            slot = i % 4
            if len(in_flight) == 4:  # the oldest submission owns this slot
                j, s = in_flight.pop(0)
                assert s == slot
                eng.wait(s)
            results[i] = eng.submit(slot, b.seq1, b.qual1, b.len1, b.seq2, b.qual2, b.len2)
            in_flight.append((i, slot))
        for _, s in reversed(in_flight):  # the rest in reverse order
            eng.wait(s)
    for b, (g1, _, g2) in zip(batches, results):
        o1, o2, _, _ = expect(tp, b)
        assert (g1 == o1).all() and (g2 == o2).all()


@pytest.mark.parametrize("shift", [2, 4, 6])
def test_result_arrays_that_are_not_eight_byte_aligned(shift):
    """cs_result is a struct of halfwords and bytes: the C ABI promises its array two-byte alignment only.  The kernels
    write a record as ONE eight-byte store where the array allows it and field by field where it does not; both forms
    must give the same records (scan kernel and resolve kernel write through the same code)."""
    dev = torch.device("cuda:0")
    tp = takara_plan()
    batch = synth.generate_pairs(50_003, 150, first_index=4242, indel_frac=0.3)
    r = Resident(batch, dev)
    raw1 = torch.zeros(batch.n * 8 + 8, dtype=torch.uint8, device=dev)
    raw2 = torch.zeros(batch.n * 8 + 8, dtype=torch.uint8, device=dev)
    p = [x.data_ptr() for x in r.t]
    m1 = abi.cs_reads(p[0], p[1], p[2], raw1.data_ptr() + shift, None, None)
    m2 = abi.cs_reads(p[3], p[4], p[5], raw2.data_ptr() + shift, None, None)
    stream = torch.cuda.Stream(device=dev)
    sh = C.c_void_p(stream.cuda_stream)
    with TrimEngine(tp, device=0, slots=0) as eng:
        eng.trim_device(r.r1, r.r2, batch.n, batch.stride, stream=sh)
        eng.trim_device(m1, m2, batch.n, batch.stride, stream=sh)
        stream.synchronize()
    g1, g2 = r.results()
    o1, o2, _, _ = expect(tp, batch)
    assert (g1 == o1).all() and (g2 == o2).all()
    for raw, want in ((raw1, g1), (raw2, g2)):
        got = raw[shift:shift + batch.n * 8].cpu().numpy().view(abi.RESULT_DTYPE).reshape(-1)
        assert (got == want).all()
        assert int(raw[:shift].sum()) == 0 and int(raw[shift + batch.n * 8:].sum()) == 0  # nothing outside the array
