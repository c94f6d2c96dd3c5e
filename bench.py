#!/usr/bin/env python3
"""Headline benchmark: M read-pairs/s of the trimming kernels (BASELINE.json metric).

    python bench.py --gpus 1 --steps 25 --warmup 10
    python bench.py --gpus N --steps K --warmup W        (spawns its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one cs_trim_device_pipelined call = scan kernel + resolve kernel over
both mates) over a synthetic batch of ``--pairs`` 2x150 bp read pairs that is already resident in HBM.
Steps are pipelined the way the runner keeps batches in flight: the scan kernels follow each other on one
stream, each step's resolve kernel (3 % of the reads, a few latency-bound waves) runs on the engine's
resolve stream under the NEXT step's scan kernel, every step writes its own result arrays, and the timed
region ends with cs_join + a device synchronize, so all K steps are complete inside it (``--serial``: both
kernels of a step on one stream, nothing overlaps).  The
default is 25 steps x 16 M pairs = four times the 100 M-pair workload of BASELINE.json config 3 (TAKARAV3 +
--trim-polyA: UMI + masks + poly-T/A + q-trim).  Batch size: a launch has a fixed cost of about 0.12 ms (the
kernel-to-kernel gap and the drain of the last tiles: a wave sees 25 tiles of 57 us each in a 4 M-pair launch), which
is 8 % of a 4 M-pair step and 2 % of a 16 M-pair step (2764 / 2960 / 3040 M pairs/s at 4 / 8 / 16 M pairs on one
box, profiles/r03_batch_size.log); 16 M pairs are 9.9 GB of the 288 GB a GPU has.  The default warm-up is ten steps:
the part needs ~50 ms under load before its clock settles (the first dozen steps run 4 % slower).  Reads shard across
ranks with no collective (weak scaling: every GPU gets its own 16 M-pair batch); torch.distributed is only
used for the barrier and the max-over-ranks of the elapsed time.

One JSON line on rank 0, with
  roofline      algorithmic bytes (616 B/pair = 2 x (150 seq + 150 qual + 8 result)) per launch over the launch
                duration, against the 8 TB/s HBM3E peak.  Pipelined form: the duration is the step time (the scan
                kernel plus what the overlapped resolve kernel adds), ``frac_scan_kernel`` / ``kernel_ms_avg`` keep
                the scan kernel alone (HIP events on the launch stream, cs_kernel_time_totals); --serial: both
                kernels, event-timed.  ``traffic`` (HBM bytes per launch) is REPLAYED from the newest committed
                counter passes (profiles/rNN_pmc_summary.json) and only when that file carries the sha256 of the
                kernel sources this run uses -- otherwise it is null and ``traffic_source`` says why;
  roofline_valu the bound that actually governs: VALU wave-instructions per launch (same replayed counter file, both
                kernels) x 2 cycles (spec issue cost of a wave64 instruction on a SIMD-32) / (1024 SIMDs x the clock
                measured in the counter passes x step time of THIS run), plus the same at this kernel's measured mix
                ceiling (2.3 cycles) and the scan kernel's SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES;
  cpu_baseline  the CPU oracle (own scalar C restatement of the cutseq->cutadapt chain -- cutadapt itself is not
                installable here; when it is, ``cpu_baseline_cutadapt`` times the real chain beside it) on this box's
                host cores on a bounded sample of the same batch; the GPU results for that sample are compared
                bit-for-bit while at it (``all_ranks_identical``).  With N > 1 every rank checks a 200 k-pair sample
                of its own shard against the oracle and rank 0 reports the MIN over ranks.
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

from cutseq_amd import abi, shard, synth, workloads  # noqa: E402
from cutseq_amd.build import kernel_source_hash  # noqa: E402
from cutseq_amd.engine import TrimEngine  # noqa: E402
from cutseq_amd.workloads import READ_LEN  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMD, MAX_CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, max clock (MI355X_MICROARCH.md)
# A wave64 VALU instruction occupies a SIMD-32 for 2 cycles (MI355X_MICROARCH.md, "Wave scheduling"): the spec
# ceiling.  tools/micro/valu_ops.hip reaches it for and/or/xor/add/sub/bitop3 once the MEASURED clock is used
# (0.42 instr/clk at the nominal 2.4 GHz = 0.50 at the ~2.0 GHz the part holds under this load); the half-rate
# opcodes of this kernel's mix (shifts left, v_perm, compares: ~15 %) make its own ceiling 2.3 cycles.
VALU_CYCLES_SPEC, VALU_CYCLES_MIX = 2.0, 2.3
PARITY_SAMPLE = 200_000  # pairs every rank checks against the oracle when world > 1
GEN_PIECE = 16_000_000   # pairs generated on the host and uploaded at a time


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)  # (the last step's resolve kernel has no scan kernel to hide under: 1/K of it per step)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=100_000_000,
                    help="read pairs per step = the resident batch (default: BASELINE config 3's 100 M pairs, 61 GB of HBM)")
    ap.add_argument("--workload", choices=["config3", "config2", "config4", "config5"], default="config3")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="pairs timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-filter", action="store_true", help="ablation: exact DP on every read")
    ap.add_argument("--serial", action="store_true", help="both kernels of a step on one stream (no overlap between steps)")
    ap.add_argument("--no-copy-probe", action="store_true", help="skip the HBM copy-bandwidth probe")
    ap.add_argument("--no-piece-check", action="store_true",
                    help="skip the piecewise launches behind the timed region (profiling runs: per-kernel averages stay clean)")
    ap.add_argument("--tier-pairs", type=int, default=4_000_000,
                    help="pairs of the tier T / tier E legs behind the timed region (config3, one GPU; 0 = skip)")
    ap.add_argument("--host-generator", action="store_true",
                    help="generate the batch on the host and upload it in pieces (rounds 1-4; default: on the device)")
    ap.add_argument("--traffic-json", type=str, default=str(default_traffic_json()),
                    help="JSON with HBM bytes per launch measured in separate rocprofv3 --pmc passes (tools/pmc.sh)")
    return ap.parse_args()


def default_traffic_json() -> Path:
    """The newest committed counter summary (profiles/rNN_pmc_summary.json)."""
    found = sorted((ROOT / "profiles").glob("r[0-9][0-9]_pmc_summary.json"))
    return found[-1] if found else ROOT / "profiles" / "r02_pmc_summary.json"


def replayed_counters(path: str, n: int):
    """Counters of the committed rocprofv3 --pmc passes, replayed only when they were collected on THESE kernel
    sources (sha256 written by tools/summarize_pmc.py) and on this launch shape -> (dict | None, why not)."""
    p = Path(path)
    if not p.exists():
        return None, f"{p.name} not found"
    pm = json.loads(p.read_text())
    if pm.get("pairs_per_launch") != n:
        return None, f"{p.name} was collected on {pm.get('pairs_per_launch')} pairs per launch, this run uses {n}"
    have, want = pm.get("kernel_source_sha256"), kernel_source_hash()
    if have != want:
        return None, (f"{p.name} was collected on other kernel sources (sha256 {str(have)[:12]}.. != {want[:12]}..): "
                      "re-run tools/pmc.sh + tools/summarize_pmc.py")
    return pm, None


def self_launch(n_ranks: int) -> int:
    """``python bench.py --gpus N`` without a launcher: N fresh child processes of this script, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set the way torch.distributed.run sets them (the counterpart of the
    reference's ``make_runner(cores=N)``, cutseq/run.py:436, 753).  The parent touches no GPU (children are started
    with subprocess, nothing is re-exec'ed), relays the children's output -- rank 0 prints the JSON line -- and
    returns non-zero as soon as any child does, stopping the others."""
    import socket
    import subprocess

    with socket.socket() as sock:  # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    children = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        children.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
    code = 0
    left = list(children)
    while left:
        for c in list(left):
            rc = c.poll()
            if rc is None:
                continue
            left.remove(c)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 1
                for other in left:  # one rank failed: the others would wait at the barrier for ever
                    other.terminate()
        if left:
            time.sleep(0.05)
    return code


def rccl_version():
    try:
        v = torch.cuda.nccl.version()
        return ".".join(map(str, v)) if isinstance(v, (tuple, list)) else str(v)
    except Exception as exc:  # (a local query: it must never take the N-rank line down)
        return f"unknown ({type(exc).__name__})"


def device_identity(index: int) -> dict:
    """Name, uuid and PCI bus id of the device a rank runs on (two ranks on one device must be visible in the line)."""
    out = {"index": index, "pid": os.getpid()}
    try:
        props = torch.cuda.get_device_properties(index)
        out["name"] = props.name
        for key in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id", "gcnArchName"):
            val = getattr(props, key, None)
            if val is not None:
                out[key] = str(val)
    except Exception as exc:  # the identity must never take the bench line down
        out["error"] = f"{type(exc).__name__}: {exc}"
    return out


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process has made no GPU call yet and becomes the launcher
        sys.exit(self_launch(args.gpus))
    if world != args.gpus:
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

    paired = workloads.is_paired(args.workload)
    tp = workloads.make_plan(args.workload, not args.no_filter)
    n = args.pairs
    # every rank trims its own shard of the read stream (weak scaling: world * n reads in all), no exchange step
    first, last = shard.shard_bounds(world * n, rank, world)
    assert last - first == n
    # The resident batch is generated ON THE DEVICE (the generator is keyed by the global pair index: csrc/synth_device.hip
    # writes what csrc/cutseq_host.c writes, tests/test_gpu_synth.py): no host buffer of the batch's size, no pageable
    # copy, start-up independent of the number of ranks.  The host only generates the head of the shard that the CPU
    # legs need (baseline sample, parity sample, tier data) -- and that head is compared with the device's bytes.
    # config 5 plants its barcodes with numpy: it keeps the host generator, in pieces.
    tiers_on = args.tier_pairs > 0 and args.workload == "config3" and world == 1
    keep = min(n, max(args.cpu_sample if world == 1 else 0, PARITY_SAMPLE, args.tier_pairs if tiers_on else 0))
    host_threads = max(1, synth.usable_cpus() // world)  # (the ranks share the host: the head of N shards at once)
    names = ("seq1", "qual1", "len1") + (("seq2", "qual2", "len2") if paired else ())
    stride = synth._stride_for(READ_LEN)
    d = {name: torch.empty((n,) if name.startswith("len") else (n, stride),
                           dtype=torch.int16 if name.startswith("len") else torch.uint8, device=dev) for name in names}
    t_gen = time.perf_counter()
    if args.workload == "config5" or args.host_generator:
        batch = None
        for at in range(0, n, GEN_PIECE):
            m = min(GEN_PIECE, n - at)
            piece = workloads.make_batch(args.workload, m, first_index=first + at, threads=host_threads)
            assert piece.stride == stride
            for name in names:
                arr = getattr(piece, name)
                d[name][at:at + m].copy_(torch.from_numpy(arr.view(np.int16) if name.startswith("len") else arr))
            if batch is None:
                batch = piece
                for name in names:
                    setattr(batch, name, getattr(piece, name)[:keep].copy())
            del piece
        generator = "host (csrc/cutseq_host.c), uploaded in pieces"
        head_equal = None
    else:
        ptrs = [d[name].data_ptr() for name in names] + [None] * (6 - len(names))
        assert workloads.fill_device(args.workload, n, ptrs, first_index=first) == stride
        batch = workloads.make_batch(args.workload, keep, first_index=first, threads=host_threads)  # the generator the parity tests use
        torch.cuda.synchronize(dev)
        head_equal = all(bool(torch.equal(d[name][:keep].cpu(), torch.from_numpy(
            getattr(batch, name).view(np.int16) if name.startswith("len") else getattr(batch, name)))) for name in names)
        generator = "device (csrc/synth_device.hip), one launch over the shard's global indices"
    gen_s = time.perf_counter() - t_gen
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
    per_rank = None
    if world > 1:
        # what a reader of the N-GPU line will ask: did every rank run at the same rate, and on its own device?
        # (gathered with the collective backend itself; the data path has no collective)
        comm_dev = "cpu" if rehearsal else dev
        mine = torch.tensor([elapsed / args.steps * 1e3, scan_ms, resolve_ms, gen_s], dtype=torch.float64, device=comm_dev)
        rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        ident = [None] * world
        dist.all_gather_object(ident, device_identity(gpu_index))
        rows = [[float(x) for x in r.cpu()] for r in rows]
        step_ms = sorted(r[0] for r in rows)
        per_rank = {
            "ms_per_step": {"min": round(step_ms[0], 4), "median": round(float(np.median(step_ms)), 4), "max": round(step_ms[-1], 4),
                            "each": [round(r[0], 4) for r in rows]},
            "scan_kernel_ms": [round(r[1], 4) for r in rows],     # HIP events around the scan kernel on each rank's stream
            "resolve_kernel_ms": [round(r[2], 4) for r in rows],
            "generator_seconds": [round(r[3], 2) for r in rows],
            "devices": ident,                                      # name / uuid / PCI bus id each rank ran on
            "distinct_devices": len({(i or {}).get("uuid") or (i or {}).get("pci_bus_id") or k for k, i in enumerate(ident)}),
            "rccl_world": dist.get_world_size(), "backend": dist.get_backend(),
            "rccl_version": None if rehearsal else rccl_version(),
        }
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    units_per_step = n
    value = world * args.steps * units_per_step / elapsed / 1e6
    bytes_per_unit = (2 if paired else 1) * (2 * READ_LEN + 8) + (1 if args.workload == "config5" else 0)  # + barcode byte
    avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
    step_s = elapsed / args.steps
    # Pipelined form: a step's bytes (both mates, every result) are spread over the scan kernel AND the resolve
    # kernel that hides under the next step's scan kernel, so the step time is the honest denominator; the
    # scan-kernel-only figure stays as a secondary field.  Serial form: both kernels, event-timed.
    launch_s = avg_kernel_s if args.serial else max(step_s, avg_kernel_s)
    achieved = bytes_per_unit * units_per_step / launch_s / 1e9
    achieved_scan = bytes_per_unit * units_per_step / avg_kernel_s / 1e9
    counters, why_not = (None, "counters are kept for the headline workload (config3) only")
    if args.workload == "config3" and args.traffic_json:
        counters, why_not = replayed_counters(args.traffic_json, n)
    traffic = counters.get("hbm_bytes_per_launch") if counters else None
    valu_insts = counters.get("valu_insts_per_launch") if counters else None
    traffic_source = (f"replayed from {Path(args.traffic_json).name} (separate rocprofv3 --pmc passes on these kernel "
                      f"sources, tools/pmc.sh)") if counters else f"null: {why_not}"

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
            "scheme": workloads.scheme_label(args.workload),
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
            "launch_ms": round(launch_s * 1e3, 4),
            "launch_ms_note": ("scan + resolve kernel, HIP events on the launch stream" if args.serial else
                               "whole step (scan kernel + what the overlapped resolve kernel adds); the scan kernel alone "
                               "is achieved_scan_kernel / frac_scan_kernel / kernel_ms_avg"),
            "achieved_scan_kernel": round(achieved_scan, 2),
            "frac_scan_kernel": round(achieved_scan / HBM_PEAK_GBPS, 5),
            "kernel": ("csdev::trim_kernel<.., MODE_SCAN> + <.., MODE_RESOLVE> (one launch of each per step, one stream)"
                       if args.serial else
                       "csdev::trim_kernel<.., MODE_SCAN> (dominant; the step's <.., MODE_RESOLVE> launch runs on the "
                       "resolve stream under the next step's scan kernel)"),
            "kernel_ms_avg": round(avg_kernel_s * 1e3, 4),
            "kernel_ms_avg_each": {"scan": round(scan_ms, 4), "resolve": round(resolve_ms, 4)},
            "bytes_per_unit": bytes_per_unit,
        },
        "batch_generator": {"where": generator, "seconds": round(gen_s, 2), "host_head_pairs": keep,
                            "device_bytes_equal_host_bytes_on_head": head_equal},
        "exact_dp_fraction": round((st1.n_exact_dp + st2.n_exact_dp) / max(1, st1.n_reads + st2.n_reads), 4),
        "refiltered_fraction": round((st1.n_refiltered + st2.n_refiltered) / max(1, st1.n_reads + st2.n_reads), 4),
    }
    if head_equal is False:
        result["parity_error"] = "the device generator's bytes differ from the host generator's on the head of the shard"
    if per_rank is not None:
        result["per_rank"] = per_rank
    if n > GEN_PIECE and args.workload != "config5" and not args.no_piece_check:
        # Full-size property (outside the timed region): the batch-sized launch must give, bit for bit, what launches of
        # GEN_PIECE reads over the same resident rows give -- the sizes the parity tests hold to the oracle.  (Tile
        # hand-out in big and small units, queue capacities and reservations all scale with the launch.)
        pieces_equal = True
        m_max = min(GEN_PIECE, n)
        po1 = torch.empty((m_max, 8), dtype=torch.uint8, device=dev)
        po2 = torch.empty((m_max, 8), dtype=torch.uint8, device=dev) if paired else None
        esz = d["seq1"].element_size() * stride
        for at in range(0, n, GEN_PIECE):
            m = min(GEN_PIECE, n - at)
            q1 = abi.cs_reads(d["seq1"].data_ptr() + at * esz, d["qual1"].data_ptr() + at * esz, d["len1"].data_ptr() + at * 2,
                              po1.data_ptr(), None, None)
            q2 = None
            if paired:
                q2 = abi.cs_reads(d["seq2"].data_ptr() + at * esz, d["qual2"].data_ptr() + at * esz,
                                  d["len2"].data_ptr() + at * 2, po2.data_ptr(), None, None)
            eng.trim_device(q1, q2, m, stride, stream=sh, pipelined=False)
            torch.cuda.synchronize(dev)
            pieces_equal = pieces_equal and bool(torch.equal(po1[:m], out1[at:at + m]))
            if paired:
                pieces_equal = pieces_equal and bool(torch.equal(po2[:m], out2[at:at + m]))
        result["full_size_launch_equals_piecewise_launches"] = pieces_equal
        if not pieces_equal:
            result["parity_error"] = "the batch-sized launch differs from launches of GEN_PIECE reads over the same rows"
        del po1, po2
    if valu_insts:
        # The bound that governs: VALU issue slots, priced with the clock MEASURED in the same counter passes
        # (GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 / kernel duration, MI355X_MICROARCH.md "DVFS give-back").
        scan = counters.get("per_kernel", {}).get("scan", {})
        clock_hz = counters.get("measured_clock_hz") or MAX_CLOCK_HZ
        ach = valu_insts / (avg_kernel_s if args.serial else step_s)  # both kernels' instructions per step
        peak_spec = N_SIMD * clock_hz / VALU_CYCLES_SPEC
        wave_cycles = scan.get("SQ_WAVE_CYCLES", {}).get("mean_per_launch")
        wait_inst = scan.get("SQ_WAIT_INST_ANY", {}).get("mean_per_launch")
        wait_any = scan.get("SQ_WAIT_ANY", {}).get("mean_per_launch")
        result["roofline_valu"] = {
            "bound": "valu_issue", "achieved": round(ach / 1e9, 2), "peak": round(peak_spec / 1e9, 2),
            "unit": "G wave-instr/s", "frac": round(ach / peak_spec, 4),
            "cycles_per_instr": VALU_CYCLES_SPEC,
            "clock_hz": round(clock_hz), "clock_source": counters.get("measured_clock_source", "nominal 2.4 GHz (no GRBM_GUI_ACTIVE pass)"),
            "frac_at_mix_ceiling": round(ach / (N_SIMD * clock_hz / VALU_CYCLES_MIX), 4),
            "mix_ceiling_cycles_per_instr": VALU_CYCLES_MIX,
            "valu_insts_per_launch": valu_insts,
            # per 64-pair tile and kernel (SQ_INSTS_VALU of the same counter passes): the figure instruction-count work is judged on
            "valu_insts_per_64_pair_tile": {
                "both": round(valu_insts / (n / 64), 1),
                "scan": round(scan.get("SQ_INSTS_VALU", {}).get("mean_per_launch", 0) / (n / 64), 1),
                "resolve": round(counters.get("per_kernel", {}).get("resolve", {}).get("SQ_INSTS_VALU", {}).get("mean_per_launch", 0) / (n / 64), 1),
            },
            "scan_wait_inst_any_over_wave_cycles": round(wait_inst / wave_cycles, 4) if wait_inst and wave_cycles else None,
            "scan_wait_any_over_wave_cycles": round(wait_any / wave_cycles, 4) if wait_any and wave_cycles else None,
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
        # (config 5 has no single CPU counterpart: its parity definition is 96 oracle runs, tests/test_gpu_workloads.py)
        import oracle  # the checker, timed as the CPU baseline; never part of the product path
        m = min(args.cpu_sample, n, batch.seq1.shape[0])  # (the host keeps the head of the first generated piece)
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
        result["all_ranks_identical"] = same
        if not same:
            result["parity_error"] = "GPU results differ from the oracle on the CPU sample"
        # when an image ships cutadapt, time the real cutseq -> cutadapt chain beside it (never here so far)
        try:
            from tools import pin_against_cutadapt as pin
            extra = pin.time_cutadapt_chain(args.workload, batch, min(m, 200_000)) if pin.cutadapt_available() else None
        except Exception as exc:  # the probe must never take the bench line down
            extra = {"error": f"{type(exc).__name__}: {exc}"}
        if extra:
            result["cpu_baseline_cutadapt"] = extra
    if world > 1 and args.workload != "config5":
        # An N-GPU rate without a parity gate is not a result: every rank checks a sample of ITS OWN shard against
        # the oracle (the host cores are shared by the ranks), rank 0 reports whether all of them agreed.
        import oracle
        m = min(PARITY_SAMPLE, n)
        threads = max(1, oracle.host_threads() // world)
        a1, n1, a2, n2 = tp.pack()
        params = tp.params()
        o1, _, _ = oracle.trim_mate(a1, n1, params, batch.seq1[:m], batch.qual1[:m], batch.len1[:m], threads=threads)
        same = bool(np.array_equal(out1[:m].cpu().numpy().view(abi.RESULT_DTYPE).reshape(-1), o1))
        if paired:
            o2, _, _ = oracle.trim_mate(a2, n2, params, batch.seq2[:m], batch.qual2[:m], batch.len2[:m], threads=threads)
            same = same and bool(np.array_equal(out2[:m].cpu().numpy().view(abi.RESULT_DTYPE).reshape(-1), o2))
        flag = torch.tensor([1 if same else 0], dtype=torch.int32, device="cpu" if rehearsal else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        result["all_ranks_identical"] = bool(int(flag.item()))
        result["parity_check"] = (f"every rank: first {m} {'pairs' if paired else 'reads'} of its own shard against the "
                                  f"CPU oracle, {threads} threads per rank; MIN over ranks")
        if not result["all_ranks_identical"]:
            result["parity_error"] = "GPU results differ from the oracle on some rank's sample"
    eng.close()
    if rank == 0 and tiers_on:
        # The other tiers, witnessed by whoever runs this file (never the headline `value`): pinned host arrays through
        # cs_trim_batch (tier T) and FASTQ files through the CLI in fresh child processes -- plain -> plain, plain -> gz,
        # multi-member gz -> gz, single-member gz -> gz (tier E) -- with the decompressed output checked against the
        # oracle's records (tools/tiers.py).
        from tools import tiers
        del d, sets
        torch.cuda.empty_cache()
        result["tiers"] = tiers.run_all(batch, min(args.tier_pairs, keep))
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()
    if result.get("parity_error"):  # (every rank holds the reduced verdict)
        sys.exit(3)


if __name__ == "__main__":
    main()
