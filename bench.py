#!/usr/bin/env python3
"""Headline benchmark: M read-pairs/s of the trimming kernels (BASELINE.json metric).

    python bench.py --gpus 1 --steps 25 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one cs_trim_device_pipelined call = scan kernel + resolve kernel over
both mates) over a synthetic batch of ``--pairs`` 2x150 bp read pairs that is already resident in HBM.
Steps are pipelined the way the runner keeps batches in flight: the scan kernels follow each other on one
stream, each step's resolve kernel (3 % of the reads, a few latency-bound waves) runs on the engine's
resolve stream under the NEXT step's scan kernel, every step writes its own result arrays, and the timed
region ends with cs_join + a device synchronize, so all K steps are complete inside it (``--serial``: both
kernels of a step on one stream, nothing overlaps).  The
default 25 steps x 4 M pairs = the 100 M-pair workload of BASELINE.json config 3 (TAKARAV3 +
--trim-polyA: UMI + masks + poly-T/A + q-trim).  Reads shard across ranks with no
collective (weak scaling: every GPU gets its own 4 M-pair batch); torch.distributed is only
used for the barrier and the max-over-ranks of the elapsed time.

One JSON line on rank 0, with
  roofline      algorithmic bytes (616 B/pair = 2 x (150 seq + 150 qual + 8 result)) per
                launch / average duration of the dominant kernel (the scan kernel; --serial: both kernels)
                from HIP events on the launch stream, against the 8 TB/s HBM3E peak (``frac_step``: the
                same bytes over the whole step time); ``traffic`` (HBM bytes per launch) is REPLAYED
                from the committed counter passes (profiles/r02_pmc_summary.json), not measured in
                this run;
  roofline_valu the bound that actually governs: VALU wave-instructions per launch (same replayed
                counter file, both kernels) x the measured issue cost per instruction / (1024 SIMDs x
                clock x step time of THIS run);
  cpu_baseline  the CPU oracle (own scalar C restatement of the cutseq->cutadapt chain --
                cutadapt itself is not installable here) timed on this box's host cores on a
                bounded sample of the same batch; the GPU results for that sample are
                compared bit-for-bit while at it.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from cutseq_amd import abi, plan as planmod, shard, synth  # noqa: E402
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig  # noqa: E402
from cutseq_amd.engine import TrimEngine  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
READ_LEN = 150
N_SIMD, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, max clock (MI355X_MICROARCH.md)
# issue cost of one VALU wave-instruction on a SIMD, measured on this part for this kernel's mix of full-
# and half-rate opcodes (tools/micro/valu_ops.hip): ~85 % at 2.3 cycles, the rest at 4.5
VALU_CYCLES_PER_INSTR = 2.65
# BASELINE config 4: inline barcode + 8-nt UMI + dual adapters, --ensure-inline-barcode
CONFIG4_SCHEME = "ACACGACGCTCTTCCGATCT(ATCACG)NNNNNNNN>AGATCGGAAGAGCACACGTC"
# BASELINE config 5 (extension, not a reference capability): 96-plex inline-barcode demultiplex + dual adapters + q-trim
CONFIG5_PLEX, CONFIG5_LEN = 96, 8


def config5_barcodes(seed: int = 96):
    """96 barcodes of 8 nt, pairwise edit distance >= 3 (seeded greedy pick)."""
    import random
    rng = random.Random(seed)

    def dist(a, b):
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
            prev = cur
        return prev[-1]

    codes = []
    while len(codes) < CONFIG5_PLEX:
        c = "".join(rng.choice("ACGT") for _ in range(CONFIG5_LEN))
        if all(dist(c, o) >= 3 for o in codes):
            codes.append(c)
    return codes


def config5_scheme(codes):
    return f"ACACGACGCTCTTCCGATCT({codes[0]})>AGATCGGAAGAGCACACGTC"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=4_000_000, help="read pairs per step (resident batch)")
    ap.add_argument("--workload", choices=["config3", "config2", "config4", "config5"], default="config3")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="pairs timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-filter", action="store_true", help="ablation: exact DP on every read")
    ap.add_argument("--serial", action="store_true", help="both kernels of a step on one stream (no overlap between steps)")
    ap.add_argument("--no-copy-probe", action="store_true", help="skip the HBM copy-bandwidth probe")
    ap.add_argument("--traffic-json", type=str, default=str(ROOT / "profiles" / "r02_pmc_summary.json"),
                    help="JSON with HBM bytes per launch measured in separate rocprofv3 --pmc passes (tools/pmc.sh)")
    return ap.parse_args()


def make_plan(workload: str, use_filter: bool):
    if workload == "config2":
        tp = planmod.single_adapter_plan("AGATCGGAAGAGC", 0.1, 3, min_length=0)
    elif workload == "config4":
        st = planmod.CutadaptConfig()
        st.ensure_inline_barcode = True
        tp = planmod.compile_paired(BarcodeConfig(CONFIG4_SCHEME), st)
    elif workload == "config5":
        codes = config5_barcodes()
        st = planmod.CutadaptConfig()
        st.demux_barcodes = codes
        tp = planmod.compile_paired(BarcodeConfig(config5_scheme(codes)), st)
    else:
        st = planmod.CutadaptConfig()
        st.trim_polyA = True
        tp = planmod.compile_paired(BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"]), st)
    tp.use_filter = use_filter
    return tp


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the trimming engine has no CPU path")
    # rehearsal of the N > 1 path on a one-GPU box: CUTSEQ_BENCH_REHEARSAL=1 puts every rank on GPU 0 and
    # uses gloo for the barrier / max (RCCL refuses two ranks on one device).  Never set by the driver.
    rehearsal = os.environ.get("CUTSEQ_BENCH_REHEARSAL") == "1"
    gpu_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    paired = args.workload != "config2"
    tp = make_plan(args.workload, not args.no_filter)
    n = args.pairs
    # every rank trims its own shard of the read stream (weak scaling: world * n reads in all), no exchange step
    first, last = shard.shard_bounds(world * n, rank, world)
    assert last - first == n
    if args.workload == "config5":
        codes = config5_barcodes()
        batch = synth.generate_pairs(n, READ_LEN, config5_scheme(codes), first_index=first)
        # every pair gets one of the 96 barcodes (2 % get none of them), 1 % sequencing error per base on top
        rng = np.random.default_rng(first + 5)
        table = np.frombuffer("".join(codes).encode(), dtype=np.uint8).reshape(CONFIG5_PLEX, CONFIG5_LEN)
        pick = rng.integers(0, CONFIG5_PLEX, size=n)
        bc_bases = table[pick].copy()
        foreign = rng.random(n) < 0.02
        bc_bases[foreign] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(int(foreign.sum()), CONFIG5_LEN))]
        err = rng.random((n, CONFIG5_LEN)) < 0.01
        bc_bases[err] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(err.sum()))]
        batch.seq1[:, :CONFIG5_LEN] = bc_bases
    elif args.workload == "config4":
        batch = synth.generate_pairs(n, READ_LEN, CONFIG4_SCHEME, first_index=first)
    elif paired:
        batch = synth.generate_pairs(n, READ_LEN, first_index=first)
    else:
        batch = synth.generate_single_adapter(n, READ_LEN, first_index=first)
    stride = batch.stride

    def up(a):
        return torch.from_numpy(a).to(dev, non_blocking=False)

    d = {"seq1": up(batch.seq1), "qual1": up(batch.qual1), "len1": up(batch.len1.view(np.int16))}
    if paired:
        d.update(seq2=up(batch.seq2), qual2=up(batch.qual2), len2=up(batch.len2.view(np.int16)))
    # one set of result arrays per step in flight (the engine lets a call start once the call three before it is done)
    n_sets = 1 if args.serial else 3
    sets = []
    for _ in range(n_sets):
        out1 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
        bc1 = torch.empty(n, dtype=torch.uint8, device=dev) if args.workload == "config5" else None
        r1 = abi.cs_reads(d["seq1"].data_ptr(), d["qual1"].data_ptr(), d["len1"].data_ptr(), out1.data_ptr(), None,
                          bc1.data_ptr() if bc1 is not None else None)
        r2 = out2 = None
        if paired:
            out2 = torch.empty((n, 8), dtype=torch.uint8, device=dev)
            r2 = abi.cs_reads(d["seq2"].data_ptr(), d["qual2"].data_ptr(), d["len2"].data_ptr(), out2.data_ptr(), None, None)
        sets.append((r1, r2, out1, out2, bc1))
    out1, out2 = sets[0][2], sets[0][3]

    eng = TrimEngine(tp, device=gpu_index, slots=0)
    # an explicit (non-default) stream: its handle is what the C ABI launches on, and the
    # torch events below are recorded on the same stream, so they bracket the kernel itself
    stream = torch.cuda.Stream(device=dev)
    sh = C.c_void_p(stream.cuda_stream)
    assert sh.value, "expected a non-null hipStream_t"

    def step(i):
        r1, r2 = sets[i % n_sets][:2]
        eng.trim_device(r1, r2, n, stride, stream=sh, pipelined=not args.serial)

    def barrier():
        eng.join(sh)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    barrier()
    eng.stats(reset=True)
    eng.kernel_time_totals(reset=True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel durations of exactly these K steps: HIP events the library records around each kernel on the
    # stream it runs on (cs_kernel_time_totals; no further events on the launch stream)
    timed_calls, scan_ms_total, resolve_ms_total = eng.kernel_time_totals()
    assert timed_calls == args.steps, (timed_calls, args.steps)
    scan_ms, resolve_ms = scan_ms_total / args.steps, resolve_ms_total / args.steps
    kernel_ms = [scan_ms + resolve_ms] if args.serial else [scan_ms]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    units_per_step = n
    value = world * args.steps * units_per_step / elapsed / 1e6
    bytes_per_unit = (2 if paired else 1) * (2 * READ_LEN + 8) + (1 if args.workload == "config5" else 0)  # + barcode byte
    avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
    step_s = elapsed / args.steps
    achieved = bytes_per_unit * units_per_step / avg_kernel_s / 1e9
    traffic = valu_insts = None
    traffic_source = None
    if args.workload == "config3" and args.traffic_json and Path(args.traffic_json).exists():
        pm = json.loads(Path(args.traffic_json).read_text())
        if pm.get("pairs_per_launch") == n:  # counters were collected on this very launch shape
            traffic = pm.get("hbm_bytes_per_launch")
            valu_insts = pm.get("valu_insts_per_launch")
            traffic_source = f"replayed from {Path(args.traffic_json).name} (separate rocprofv3 --pmc passes, tools/pmc.sh)"

    st1, st2 = eng.stats()
    result = {
        "metric": "M read-pairs/sec" if paired else "M reads/sec",
        "value": round(value, 3),
        "unit": "M read-pairs/s" if paired else "M reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": {
                "config3": "BASELINE config 3: 2x150 bp pairs, full TAKARAV3 scheme + --trim-polyA (UMI+mask+polyT/A+q-trim), "
                           f"{args.steps} steps x {n} resident pairs per GPU",
                "config2": f"BASELINE config 2: 150 bp single-end, 3' adapter AGATCGGAAGAGC e=0.1, {args.steps} steps x {n} reads",
                "config4": "BASELINE config 4: 2x150 bp pairs, custom scheme with inline barcode + 8 nt UMI + dual adapters, "
                           f"--ensure-inline-barcode, {args.steps} steps x {n} resident pairs per GPU",
                "config5": "BASELINE config 5 (extension): 2x150 bp pairs, 96-plex 8-nt inline-barcode demultiplex + dual "
                           f"adapters + q-trim in one pass, {args.steps} steps x {n} resident pairs per GPU",
            }[args.workload],
            "pairs_per_step_per_gpu": n,
            "read_len": READ_LEN,
            "scheme": {"config3": "TAKARAV3", "config2": "-a AGATCGGAAGAGC", "config4": CONFIG4_SCHEME,
                       "config5": "P5(96 x 8 nt)>P7 --demux-barcodes"}[args.workload],
            "prefilter": not args.no_filter,
            "pipeline": "serial" if args.serial else "resolve kernel of step i under the scan kernel of step i+1 (2 streams, 3 result sets)",
            "parallelism": f"shard{world}" if world > 1 else "single",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 5),
            "traffic": traffic,
            "traffic_source": traffic_source,
            "frac_step": round(bytes_per_unit * units_per_step / step_s / 1e9 / HBM_PEAK_GBPS, 5),
            "kernel": ("csdev::trim_kernel<.., MODE_SCAN> + <.., MODE_RESOLVE> (one launch of each per step, one stream)"
                       if args.serial else
                       "csdev::trim_kernel<.., MODE_SCAN> (dominant; the step's <.., MODE_RESOLVE> launch runs on the "
                       "resolve stream under the next step's scan kernel)"),
            "kernel_ms_avg": round(avg_kernel_s * 1e3, 4),
            "kernel_ms_avg_each": {"scan": round(scan_ms, 4), "resolve": round(resolve_ms, 4)},
            "bytes_per_unit": bytes_per_unit,
        },
        "exact_dp_fraction": round((st1.n_exact_dp + st2.n_exact_dp) / max(1, st1.n_reads + st2.n_reads), 4),
    }
    if valu_insts:
        # the bound that governs: VALU issue slots.  achieved / peak in wave-instructions per second.
        peak = N_SIMD * CLOCK_HZ / VALU_CYCLES_PER_INSTR
        ach = valu_insts / (avg_kernel_s if args.serial else step_s)  # both kernels' instructions per step
        result["roofline_valu"] = {
            "bound": "valu_issue", "achieved": round(ach / 1e9, 2), "peak": round(peak / 1e9, 2),
            "unit": "G wave-instr/s", "frac": round(ach / peak, 4),
            "valu_insts_per_launch": valu_insts, "cycles_per_instr": VALU_CYCLES_PER_INSTR,
            "source": traffic_source,
        }

    if rank == 0 and world == 1 and not args.no_copy_probe:
        # measured HBM copy bandwidth of this box: a second denominator next to the 8 TB/s spec
        nbytes = 1 << 30
        src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        src.fill_(7)
        for _ in range(3):
            dst.copy_(src)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize(dev)
        copy_gbps = 2.0 * nbytes * reps / (e0.elapsed_time(e1) / 1e3) / 1e9  # read + write
        result["roofline"]["copy_measured_GBps"] = round(copy_gbps, 1)
        result["roofline"]["frac_of_copy_measured"] = round(achieved / copy_gbps, 5)
        del src, dst

    if rank == 0 and world == 1 and args.cpu_sample > 0 and args.workload != "config5":
        # (config 5 has no single CPU counterpart: its parity definition is 96 oracle runs, tests/test_gpu_demux.py)
        import oracle  # the checker, timed as the CPU baseline; never part of the product path
        m = min(args.cpu_sample, n)
        threads = oracle.host_threads()
        a1, n1, a2, n2 = tp.pack()
        params = tp.params()
        t0 = time.perf_counter()
        o1, _, _ = oracle.trim_mate(a1, n1, params, batch.seq1[:m], batch.qual1[:m], batch.len1[:m], threads=threads)
        o2 = None
        if paired:
            o2, _, _ = oracle.trim_mate(a2, n2, params, batch.seq2[:m], batch.qual2[:m], batch.len2[:m], threads=threads)
        cpu_s = time.perf_counter() - t0
        g1 = out1[:m].cpu().numpy().view(abi.RESULT_DTYPE).reshape(-1)
        same = bool(np.array_equal(g1, o1))
        if paired:
            g2 = out2[:m].cpu().numpy().view(abi.RESULT_DTYPE).reshape(-1)
            same = same and bool(np.array_equal(g2, o2))
        result["cpu_baseline"] = {
            "value": round(m / cpu_s / 1e6, 4),
            "unit": result["unit"],
            "cores": threads,
            "kind": "port",
            "sample": f"first {m} {'pairs' if paired else 'reads'} of the same batch, own scalar C restatement "
                      f"(oracle/cutseq_oracle.c), {threads} threads, {cpu_s:.2f} s wall = {cpu_s * threads:.0f} CPU-seconds; "
                      "cutadapt is not installable here",
            "gpu_results_identical_on_sample": same,
        }
        if not same:
            result["parity_error"] = "GPU results differ from the oracle on the CPU sample"
    if rank == 0:
        print(json.dumps(result))
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and result.get("parity_error"):
        sys.exit(3)


if __name__ == "__main__":
    main()
