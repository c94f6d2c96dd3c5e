#!/usr/bin/env python3
"""cutseq-compatible command line on top of the HIP trimming engine.

Rewritten counterpart of the reference's ``cutseq/run.py``: same flags (run.py:874-1020), same
preset resolution and upper-casing (1041-1056), same output naming (1058-1107), same single /
paired dispatch (842-863).  Where the reference assembles cutadapt modifiers and calls
``runner.run``, this compiles the op table (``plan.compile_*``) and streams record-aligned FASTQ
chunks through :class:`cutseq_amd.engine.TrimEngine` (one per visible GPU, chunks round-robin,
ordered write-back).  ``-t/--threads`` bounds the host thread pool that parses, formats and (de)compresses
(the reference passes it to ``make_runner(cores=N)``, run.py:436, 753); the per-read work runs on the GPU(s).
"""
from __future__ import annotations

import argparse
import logging
import os
import queue
import re
import sys
import threading
import time
from collections import deque
from typing import List, Optional

from . import __version__, abi, report, shard
from .common import BUILDIN_ADAPTERS, BarcodeConfig, print_builtin_adapters, remove_fq_suffix
from .plan import CutadaptConfig, TrimPlan, compile_paired, compile_single

logging.basicConfig(level=logging.INFO, format="%(asctime)s -  %(levelname)s - %(message)s")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(
        prog="cutseq",
        description="Trim sequencing adapters from NGS data automatically (MI355X-native engine, cutseq-compatible CLI).",
    )
    p.add_argument("input_file", type=str, nargs="*",
                   help="Input file path for NGS data, one or two files (for single or paired-end reads).")
    p.add_argument("-a", "--adapter-scheme", type=str,
                   help="Adapter sequence configuration string. Example: P5(INLINE5)UMI5XXXS>P7(INLINE3)UMI3XXXS. "
                        "Where P5/P7 are adapter sequences, (INLINE5/3) are optional inline barcodes, "
                        "UMI5/3 are N's for UMI bases, XXX are mask sequences, S is strand (>/< or -).")
    p.add_argument("-A", "--adapter-name", help="Built-in adapter name. choices:\n" + ",".join(BUILDIN_ADAPTERS.keys()))
    p.add_argument("-O", "--output-prefix", type=str,
                   help="Output file prefix for trimmed, short, and untrimmed data. "
                        "If not provided, output filenames are derived from input filenames.")
    p.add_argument("-o", "--output-file", type=str, nargs="+",
                   help="Output file path(s) for successfully trimmed reads. Must match number of input files.")
    p.add_argument("-s", "--short-file", type=str, nargs="+",
                   help="Output file path(s) for reads discarded due to being too short after trimming.")
    p.add_argument("-u", "--untrimmed-file", type=str, nargs="+",
                   help="Output file path(s) for reads discarded because expected inline barcodes were not found.")
    p.add_argument("--json-file", type=str, help="Output JSON file for trimming statistics.")
    p.add_argument("-q", "--min-quality", type=int, default=20,
                   help="Minimum quality score for trimming read tails. (Default: 20)")
    p.add_argument("-m", "--min-length", type=int, default=20,
                   help="Minimum length of reads to keep after trimming. (Default: 20)")
    p.add_argument("--with-rname-suffix", action="store_true",
                   help="Indicate if read names have MGI-style suffixes like '/1', '/2', '.1', or '.2' to be stripped.")
    p.add_argument("--ensure-inline-barcode", action="store_true",
                   help="If set, reads without the specified inline barcode(s) will be written to the untrimmed files.")
    p.add_argument("--trim-polyA", action="store_true", help="Enable trimming of polyA/T tails.")
    p.add_argument("--trim-polyA-wo-direction", action="store_true",
                   help="Trim polyA/T tails regardless of strand information.")
    p.add_argument("--conditional-cutter", action=argparse.BooleanOptionalAction, default=True,
                   help="Enable/disable conditional cutting for UMIs/masks.")
    p.add_argument("--force-trim-min-length", type=int, default=50,
                   help="Minimum read length to enforce UMI/mask trimming even if no adapter is found. (Default: 50)")
    p.add_argument("--force-anywhere", action="store_true",
                   help="Force adapter trimming to match anywhere in the read, not just at the ends.")
    p.add_argument("--auto-rc", action="store_true",
                   help="Automatically reverse complement reads if the library strand is '-'. "
                        "For paired-end, R1 and R2 will be swapped.")
    p.add_argument("-t", "--threads", type=int, default=None,
                   help="Number of host threads for reading, (de)compression and writing. The reference's -t "
                        "(default 1) is the number of cutadapt worker PROCESSES that do the per-read work; here that "
                        "work runs on the GPU(s), so -t only bounds the host I/O pool and the output does not depend "
                        "on it. -t 1 is accepted and means one host thread. (Default: all usable cores)")
    p.add_argument("-n", "--dry-run", action="store_true",
                   help="Print the sequence of modifier steps instead of running the pipeline.")
    p.add_argument("-V", "--version", action="version", version=f"%(prog)s {__version__}")
    p.add_argument("--list-adapters", action="store_true",
                   help="List all built-in adapter names and their schemes, then exit.")
    p.add_argument("--cutadapt-selection", choices=["4", "3"], default="4",
                   help="Aligner candidate selection to follow: cutadapt >= 4 (default) or 3.x (DESIGN.md section 0).")
    p.add_argument("--ranks", type=int, default=1, metavar="N",
                   help="Extension: one process per GPU (N processes, rank r on GPU r or on the r-th entry of "
                        "CUTSEQ_DEVICES); the input is split at record boundaries, the output parts are concatenated "
                        "in rank order. Output is identical to the one-process run.")
    p.add_argument("--rank-spec", type=str, help=argparse.SUPPRESS)  # a child of --ranks: its share of the run
    p.add_argument("--demux-barcodes", type=str, metavar="FILE",
                   help="Extension: demultiplex on the 5' inline barcode. FILE lists 'name<TAB>sequence' per line (all as "
                        "long as the scheme's inline barcode); trimmed reads go to <prefix>_<name>_trimmed_R1/2.fastq.gz, "
                        "reads without any of the barcodes to the untrimmed files. Equals one --ensure-inline-barcode "
                        "run per barcode.")
    return p


# Output naming (user-visible contract of the reference CLI, run.py:1058-1107): for every output class the
# files are, in this order of precedence, the ones given on the command line (one per input file), the
# -O prefix, or the input name without its FASTQ suffix, followed by "_<class>_R<mate>.fastq.gz".
OUTPUT_CLASSES = (  # (argparse attribute, word in the file name)
    ("output_file", "trimmed"),
    ("short_file", "short"),
    ("untrimmed_file", "untrimmed"),
)
_SCHEME_HAS_INLINE = re.compile(r".*\([ATGCatgc]+\).*")


def _fail(message: str):
    logging.error(message)
    sys.exit(1)


def output_paths(given, inputs, prefix, word):
    """File names of one output class, one per input file."""
    if given:
        if len(given) != len(inputs):
            _fail(f"Number of {word} output files ({len(given)}) must match number of input files ({len(inputs)}).")
        return list(given)
    stems = [prefix] * len(inputs) if prefix is not None else [remove_fq_suffix(name) for name in inputs]
    return [f"{stem}_{word}_R{mate}.fastq.gz" for mate, stem in enumerate(stems, 1)]


def resolve_scheme(args) -> str:
    """-a wins over -A; -A looks the preset up case-insensitively and, like the reference, falls back to
    reading an unknown name as a scheme (run.py:1041-1056)."""
    scheme = args.adapter_scheme
    if args.adapter_name is not None and scheme is not None:
        logging.info("Adapter scheme is provided, ignoring adapter name.")
    elif args.adapter_name is not None:
        scheme = BUILDIN_ADAPTERS.get(args.adapter_name.upper())
        if scheme is None:
            logging.error(f"Adapter name '{args.adapter_name} not found in built-in adapters.")
            scheme = args.adapter_name
    if scheme is None:
        _fail("Adapter scheme or name is required. Use -a or -A.")
    return scheme.replace(" ", "").upper()


def resolve_args(args):
    """What the reference's main() settles between parse_args and run_cutseq (run.py:1029-1107)."""
    if args.list_adapters:
        print_builtin_adapters()
        sys.exit(0)
    inputs = args.input_file or []
    if len(inputs) > 2:
        _fail("Input file can not be more than two.")
    args.adapter_scheme = resolve_scheme(args)
    if not inputs:
        # the reference dies with an IndexError here (run.py:1079-1083); same exit class, clearer message
        _fail("Input file is required.")
    args.demux = None
    if args.demux_barcodes:
        from .demux import read_barcode_file
        if args.output_file:
            _fail("--demux-barcodes names the trimmed files itself (<prefix>_<name>_trimmed_R1.fastq.gz): drop -o.")
        try:
            args.demux = read_barcode_file(args.demux_barcodes)  # ([names], [sequences])
        except (OSError, ValueError) as exc:
            _fail(str(exc))
    wants_untrimmed = bool(args.untrimmed_file) or args.demux is not None or (
        args.ensure_inline_barcode and _SCHEME_HAS_INLINE.match(args.adapter_scheme) is not None)
    for attr, word in OUTPUT_CLASSES:
        if attr == "untrimmed_file" and not wants_untrimmed:
            args.untrimmed_file = [None] * len(inputs)
        else:
            setattr(args, attr, output_paths(getattr(args, attr), inputs, args.output_prefix, word))
    if args.demux is not None:  # one pair of trimmed files per barcode instead of the common one
        args.demux_files = [output_paths(None, inputs, args.output_prefix, f"{name}_trimmed") for name in args.demux[0]]
    return args


def settings_from_args(args) -> CutadaptConfig:
    st = CutadaptConfig()
    st.rname_suffix = args.with_rname_suffix  # parsed, stored, never consulted (as in the reference)
    st.ensure_inline_barcode = args.ensure_inline_barcode
    st.trim_polyA = args.trim_polyA
    st.trim_polyA_wo_direction = args.trim_polyA_wo_direction
    st.conditional_cutter = args.conditional_cutter
    st.threads = args.threads if args.threads is not None else 0
    st.min_length = args.min_length
    st.min_quality = args.min_quality
    st.dry_run = args.dry_run
    st.auto_rc = args.auto_rc
    st.json_file = args.json_file
    st.force_trim_min_length = args.force_trim_min_length
    st.force_anywhere = args.force_anywhere
    st.select_rule = abi.CS_SELECT_LEFTMOST if args.cutadapt_selection == "4" else abi.CS_SELECT_SCORE
    if getattr(args, "demux", None) is not None:
        st.demux_barcodes = list(args.demux[1])
    return st


def compile_plan(args, barcode: BarcodeConfig, settings: CutadaptConfig) -> TrimPlan:
    if len(args.input_file) == 1:
        return compile_single(barcode, settings, untrimmed_requested=args.untrimmed_file[0] is not None)
    return compile_paired(
        barcode, settings,
        untrimmed_requested=args.untrimmed_file[0] is not None and args.untrimmed_file[1] is not None)


def dry_run_steps(tp: TrimPlan) -> List[str]:
    """The reference's modifier list in ITS order (cutseq/run.py:326-426, 533-731): the two SuffixRemovers, the
    adapter and UMI steps, the (PairedEnd)Renamer right behind the UMI step (run.py:377-380, 642-645), then masks,
    poly-A, quality trimming (and the single-end reverse complement).  The text of a step is this build's own
    (cutadapt's ``repr``s are not available here); the ORDER is the reference's."""
    if tp.paired:
        steps = [f"({a}, {b})" for a, b in zip(tp.r1.describe(), tp.r2.describe())]
        renamer = f"PairedEndRenamer({'{id}_{r1.cut_prefix}{r2.cut_prefix}' if tp.has_umi else '{id}'!r})"
    else:
        steps = list(tp.r1.describe())
        renamer = f"Renamer({'{id}_{cut_prefix}{cut_suffix}' if tp.has_umi else '{id}'!r})"
    at = len(tp.r1.ops) if tp.r1.rename_at is None else tp.r1.rename_at
    steps.insert(len(tp.r1.name_suffixes) + at, renamer)
    if not tp.paired and tp.reverse_complement:
        steps.append("ReverseComplementConverter()")
    return steps


def dry_run(tp: TrimPlan, barcode: BarcodeConfig):
    if tp.paired:  # run.py:734-749: the parts through print, the steps through logging
        for b in ["p5", "p7", "inline5", "inline3", "umi5", "umi3", "mask5", "mask3", "strand"]:
            print(f"{b}: {getattr(barcode, b)}")
        for i, step in enumerate(dry_run_steps(tp), 1):
            logging.info(f"Step {i}: {step}")
    else:  # run.py:429-432
        for i, step in enumerate(dry_run_steps(tp), 1):
            print(f"Step {i}: {step}")


class _DeviceWorker(threading.Thread):
    """One GPU: owns the device's TrimEngine, keeps ``SLOTS`` chunks in flight on it (the H2D copy of one
    overlaps the kernels and the D2H copy of the other) and reports every finished chunk to ``done``.
    Counterpart of one worker process of ``make_runner(inpaths, cores=N)`` (cutseq/run.py:436, 753)."""

    SLOTS = 2

    def __init__(self, tp: TrimPlan, device: int, done: "queue.Queue", chunk_reads: int):
        super().__init__(daemon=True, name=f"cutseq-gpu{device}")
        self.tp, self.device, self.done, self.chunk_reads = tp, device, done, chunk_reads
        self.inbox: "queue.Queue" = queue.Queue(maxsize=self.SLOTS)
        self.engine = None
        self.stride_cap = 0
        self.stats = None  # summed cs_stats of every engine this device has had
        self.error: Optional[BaseException] = None

    def _collect_stats(self):
        part = [s.as_dict() for s in self.engine.stats()]
        self.stats = part if self.stats is None else [shard.merge_stats([a, b]) for a, b in zip(self.stats, part)]

    def _engine_for(self, stride: int, inflight: deque):
        """The engine, rebuilt with longer rows when a chunk needs them -- after everything the old one still
        has in flight came back and its counters were saved."""
        if self.engine is not None and stride <= self.stride_cap:
            return
        from .engine import TrimEngine
        while inflight:
            self._finish_oldest(inflight)
        if self.engine is not None:
            self._collect_stats()
            self.engine.close()
        self.stride_cap = max(stride, 152)
        self.engine = TrimEngine(self.tp, device=self.device, slots=self.SLOTS, max_reads=self.chunk_reads,
                                 max_stride=self.stride_cap)

    def _finish_oldest(self, inflight: deque):
        k, slot, chunk, res = inflight.popleft()
        self.engine.wait(slot)
        self.done.put((k, chunk, res))

    def run(self):
        from . import fastq
        inflight: deque = deque()  # (chunk index, slot, chunk, result arrays), oldest first
        n = 0
        try:
            while True:
                item = self.inbox.get()
                if item is None:
                    break
                k, chunk = item
                self._engine_for(chunk.stride, inflight)
                if len(inflight) == self.SLOTS:
                    self._finish_oldest(inflight)  # frees exactly the slot this chunk takes
                paired = chunk.paired
                # result arrays in pinned memory, like the inputs: the D2H copy is a DMA transfer too
                owned = [fastq.PINNED.take(chunk.n * 8)]
                out1 = owned[0][: chunk.n * 8].view(abi.RESULT_DTYPE)
                cap2 = out2 = None
                if self.tp.needs_cap2:
                    owned.append(fastq.PINNED.take(chunk.n * 4))
                    cap2 = owned[-1][: chunk.n * 4].view(abi.CAP2_DTYPE)
                if paired:
                    owned.append(fastq.PINNED.take(chunk.n * 8))
                    out2 = owned[-1][: chunk.n * 8].view(abi.RESULT_DTYPE)
                if self.tp.demux is not None:
                    owned.append(fastq.PINNED.take(chunk.n))
                    chunk.bc = owned[-1][: chunk.n]
                chunk._owned = chunk._owned + tuple((fastq.PINNED, b) for b in owned)
                slot = n % self.SLOTS
                res = self.engine.submit(slot, chunk.seq1, chunk.qual1, chunk.len1, chunk.seq2, chunk.qual2,
                                         chunk.len2, out=(out1, cap2, out2), bc=chunk.bc)
                inflight.append((k, slot, chunk, res))
                n += 1
            while inflight:
                self._finish_oldest(inflight)
            if self.engine is not None:
                self._collect_stats()
        except BaseException as exc:
            self.error = exc
        finally:
            if self.engine is not None:
                self.engine.close()
            self.done.put(self)  # "this device is finished" (or failed)


def run_pipeline(args, tp: TrimPlan, shares=None) -> dict:
    """Stream the input through the GPU engine(s); returns the run statistics.

    Counterpart of ``runner.run(pipeline, Progress(), outfiles)`` (cutseq/run.py:473, 794): chunks of
    ``fastq.CHUNK_READS`` records go round-robin to one worker thread per GPU, come back in any order and are
    formatted / compressed / written strictly in input order."""
    from . import capi, fastq

    n_dev = capi.device_count()
    _phase("devices counted")
    if n_dev <= 0:
        raise capi.HipUnavailable("no HIP device visible; cutseq_amd has no CPU trimming path")
    if getattr(args, "threads", None) is not None:
        if args.threads < 1:
            _fail("-t/--threads must be at least 1.")
        try:
            fastq.set_threads(args.threads)
        except RuntimeError:  # a second run in one process (tests): the pool of the first run stays
            logging.warning("-t/--threads ignored: the host thread pool of this process is already running.")
    want = os.environ.get("CUTSEQ_DEVICES")  # e.g. "0,1,2,3"; a device may be listed twice (two engines on it)
    devices = [int(x) for x in want.split(",")] if want else list(range(n_dev))
    if any(d < 0 or d >= n_dev for d in devices):
        _fail(f"GPU {max(devices)} does not exist: this process sees {n_dev} (--ranks N puts rank r on GPU r; "
              "CUTSEQ_DEVICES=0,0,... lists the GPUs to use, one entry per rank or worker, repeats allowed).")
    # Text path (default): the device parses the records and formats the output (textio.py / cs_text_*); the host
    # path below (native parser / formatter in a thread pool) stays for demultiplexing runs and as CUTSEQ_TEXT_PATH=0.
    if os.environ.get("CUTSEQ_TEXT_PATH", "1") != "0":
        from . import textio
        return textio.run_text_pipeline(args, tp, devices, _block_records(args, len(devices)),
                                        shares=shares)
    if shares is not None:
        raise ValueError("--ranks needs the text path (no demultiplexing, CUTSEQ_TEXT_PATH unset)")
    from . import codec
    for name in args.input_file:  # the host parser reads plain and gzip FASTQ files, nothing else
        container, first, _ = codec.sniff_input(name) if name != "-" else ("stdin", b"", None)
        if container not in ("plain", "gzip") or first in (b">", b"#"):
            _fail(f"{name}: FASTA input, standard input and bzip2 / xz / zstandard files need the text path (CUTSEQ_TEXT_PATH unset).")
    chunk_reads = int(os.environ.get("CUTSEQ_CHUNK_READS", fastq.CHUNK_READS))
    paired = tp.paired
    in1 = args.input_file[0]
    in2 = args.input_file[1] if paired else None

    # output files: trimmed / short / untrimmed, per mate; paired --auto-rc on a '-' library swaps
    # the trimmed pair (run.py:787-791)
    def mk(names):
        return [fastq.OutputFile(n) if n else None for n in names]

    n_bins = len(tp.demux.barcodes) if tp.demux is not None else 0
    trimmed = mk(args.output_file) if not n_bins else [None] * len(args.output_file)
    if paired and tp.swap_outputs:
        trimmed = trimmed[::-1]
    outs = [trimmed, mk(args.short_file), mk(args.untrimmed_file)]
    for names in (args.demux_files if n_bins else ()):  # streams 3 .. : the trimmed reads of one barcode each
        files = mk(names)
        outs.append(files[::-1] if paired and tp.swap_outputs else files)
    # which (route, mate) streams go to disk, and whether they are gzip
    gz = [[(fh.gz if fh is not None else None) for fh in (group + [None])[:2]] for group in outs]

    totals = report.new_totals()
    progress = report.Progress()
    t0 = time.perf_counter()
    pool = fastq._pool()
    max_finishing = fastq.pool_size() + 2  # chunks being formatted / compressed (about 90 MB each)
    done: "queue.Queue" = queue.Queue()
    if tp.demux is not None:
        from . import demux  # look-up tables once, here: every device worker would otherwise build its own
        demux.ensure_tables(tp, devices[0])
    workers = [_DeviceWorker(tp, dev, done, chunk_reads) for dev in devices]
    failure: List[BaseException] = []

    def finish(chunk, r1, cap2, r2):
        """Pool job: records -> bytes on their way to disk; also this chunk's share of the report."""
        try:
            blobs, counts = fastq.finish_chunk(chunk, tp, r1, cap2, r2, gz, n_bins=n_bins)
            part = report.new_totals()
            report.account_chunk(part, tp, chunk.len1, r1, chunk.len2 if paired else None, r2 if paired else None)
            part["routes"] = counts
            return blobs, part
        finally:
            chunk.release()

    def collect():
        """Re-establishes the input order behind the GPUs and feeds the writers."""
        waiting, next_k, alive = {}, 0, len(workers)
        finishing: deque = deque()
        try:
            while alive or waiting:
                item = done.get()
                if isinstance(item, _DeviceWorker):
                    alive -= 1
                    if item.error is not None:
                        raise item.error
                    continue
                waiting[item[0]] = item
                while next_k in waiting:
                    _, chunk, (r1, cap2, r2) = waiting.pop(next_k)
                    next_k += 1
                    progress.update(chunk.n)
                    fut = pool.submit(finish, chunk, r1, cap2, r2)
                    for route in range(len(outs)):
                        for m in range(2 if paired else 1):
                            if outs[route][m] is not None:
                                outs[route][m].write_job(fut, route, m)
                    finishing.append(fut)
                    while finishing and (len(finishing) >= max_finishing or finishing[0].done()):
                        report.merge_totals(totals, finishing.popleft().result()[1])
                if not alive and waiting and next_k not in waiting:
                    raise RuntimeError("a chunk went missing between the GPU workers and the writer")
            while finishing:
                report.merge_totals(totals, finishing.popleft().result()[1])
        except BaseException as exc:
            failure.append(exc)

    collector = threading.Thread(target=collect, daemon=True, name="cutseq-collect")
    for w in workers:
        w.start()
    collector.start()
    first_error: Optional[BaseException] = None
    try:
        for k, chunk in enumerate(fastq.read_chunks(in1, in2, chunk_reads, pinned=True)):
            if failure:
                break
            if chunk.stride > abi.CS_MAX_STRIDE:
                raise ReadTooLong(f"reads longer than {abi.CS_MAX_STRIDE} nt are not supported by the GPU tile "
                                  f"(record {k * chunk_reads + int(chunk.len1.argmax()) + 1} or its mate)")
            w = workers[k % len(workers)]
            while not failure:
                try:
                    w.inbox.put((k, chunk), timeout=0.2)
                    break
                except queue.Full:
                    continue
    except BaseException as exc:
        first_error = exc
    finally:
        for w in workers:
            while True:  # the sentinel must get in even when the worker died with a full inbox
                try:
                    w.inbox.put(None, timeout=0.2)
                    break
                except queue.Full:
                    if not w.is_alive():
                        break
        for w in workers:
            w.join()
        collector.join()
        for group in outs:
            for fh in group:
                if fh is None:
                    continue
                try:
                    fh.close()
                except BaseException as exc:  # keep closing the others; report the first failure
                    first_error = first_error or exc
    if first_error is None and failure:
        first_error = failure[0]
    if first_error is None:
        first_error = next((w.error for w in workers if w.error is not None), None)
    if first_error is not None:
        raise first_error  # (the pinned arena is left alone: parse / format jobs of the pool may still hold its buffers)
    fastq.PINNED.free_pinned()  # every chunk has been written and released
    stats = [w.stats for w in workers if w.stats is not None]
    totals["seconds"] = time.perf_counter() - t0
    progress.close()
    totals["bin_names"] = list(args.demux[0]) if n_bins else None
    totals["stats"] = stats
    totals["devices"] = devices
    return totals


class ReadTooLong(ValueError):
    pass


def run_cutseq(args, argv=None):
    barcode = BarcodeConfig(args.adapter_scheme)
    settings = settings_from_args(args)
    try:
        tp = compile_plan(args, barcode, settings)
    except ValueError as exc:  # e.g. a barcode list the demultiplexer cannot take: the CLI's exit style, no traceback
        _fail(str(exc))
    if settings.dry_run:
        dry_run(tp, barcode)
        return None
    totals = None
    if getattr(args, "rank_spec", None):  # a child of --ranks: its share, its part files, its totals; the parent reports
        from . import ranks
        spec = ranks.load_spec(args.rank_spec)
        for key, names in spec["outputs"].items():
            if key.startswith("demux_files:"):
                args.demux_files[int(key.split(":")[1])] = names
            else:
                setattr(args, key, names)
        try:
            totals = run_pipeline(args, tp, shares=spec["inputs"])
        except (ValueError, OSError) as exc:
            from . import fastq
            # records out of step in THIS share (malformed record, mate ids that differ, unequal record counts): the one
            # failure that tells the parent its split by gzip members did not hold (ranks.RANK_EXIT_RECORDS)
            if isinstance(exc, fastq.FastqFormatError) or "IDs not identical" in str(exc) or "improperly paired" in str(exc):
                logging.error(str(exc))
                sys.exit(ranks.RANK_EXIT_RECORDS)
            raise
        ranks.dump_totals(spec, totals)
        return totals
    if getattr(args, "ranks", 1) > 1:
        if os.environ.get("CUTSEQ_TEXT_PATH", "1") == "0":
            logging.warning("--ranks ignored: it needs the text path.")
        else:
            from . import ranks
            totals = ranks.run_parent(list(sys.argv[1:] if argv is None else argv), args, tp)
    if totals is None:
        _phase("plan compiled")
        totals = run_pipeline(args, tp)
        _phase("pipeline done")
    if args.json_file:
        paired = tp.paired
        rep = report.json_report(
            tp, totals, barcode, args.input_file[0], args.input_file[1] if paired else None,
            args.output_file[0], args.output_file[1] if paired else None,
            args.short_file[0], args.short_file[1] if paired else None,
            args.untrimmed_file[0], args.untrimmed_file[1] if paired else None)
        report.write_json(args.json_file, rep)
    print(report.minimal_report(tp, totals), file=sys.stderr)
    return totals


def _block_records(args, n_devices: int) -> int:
    """Records per block of the text path: ``CUTSEQ_CHUNK_READS`` if set, else 262 144 -- less for small inputs, so
    that a short run does not allocate (and page-lock) for blocks it never fills: about two dozen blocks per device.
    (With the block buffers on huge pages the size matters little for a fresh 8 M-pair process -- 0.83 to 0.90 s for
    plain text with blocks of 65 536 to 262 144 records, and gzip input likes the big ones: 1.33 against 1.51 s;
    profiles/r03_cold_runs.log.)"""
    from . import textio
    if os.environ.get("CUTSEQ_CHUNK_READS"):
        return int(os.environ["CUTSEQ_CHUNK_READS"])
    try:
        size = os.path.getsize(args.input_file[0])
    except OSError:
        return textio.CHUNK_READS
    if args.input_file[0].endswith(".gz"):
        size *= 4  # (a guess at the text behind it)
    records = size // 300  # (a short-read record is ~330 bytes)
    want = records // (24 * max(n_devices, 1))
    block = textio.CHUNK_READS
    while block > 16384 and block > want:
        block //= 2
    return block


def _phase(tag: str) -> None:
    """Diagnostic (CUTSEQ_PROFILE=1): seconds since the interpreter started, per phase of the run."""
    if os.environ.get("CUTSEQ_PROFILE") == "1":
        print(f'{{"cutseq_phase": "{tag}", "process_seconds": {time.perf_counter() - _T_IMPORT + _IMPORT_OFFSET:.3f}}}', file=sys.stderr)


_T_IMPORT = time.perf_counter()
try:  # seconds between process start and this module's import (Linux: from /proc)
    with open("/proc/self/stat") as _fh:
        _start_ticks = int(_fh.read().rsplit(")", 1)[1].split()[19])
    with open("/proc/uptime") as _fh:
        _IMPORT_OFFSET = float(_fh.read().split()[0]) - _start_ticks / os.sysconf("SC_CLK_TCK")
except Exception:  # pragma: no cover
    _IMPORT_OFFSET = 0.0


def main(argv: Optional[List[str]] = None):
    _phase("main")
    parser = build_parser()
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) == 0:
        parser.print_help(sys.stdout)
        sys.exit(0)
    args = resolve_args(parser.parse_args(argv))
    try:
        run_cutseq(args, argv)
    except (ReadTooLong, FileNotFoundError) as exc:  # user errors: the reference's exit style (run.py:1035-1039)
        _fail(str(exc))
    except RuntimeError as exc:
        if type(exc).__name__ != "RankFailure":
            raise
        _fail(f"{exc} -- its own message is above")  # (--ranks: the rank has already said what went wrong)


def console_main():
    """Entry point of the ``cutseq`` console script and of ``python -m cutseq_amd.run``.  When the run is over the
    process ends at once: unwinding the HIP runtime, its streams and gigabytes of page-locked memory takes longer than
    a short run itself (profiles/HISTORY.md), and the operating system reclaims all of it anyway.  Tools that
    collect their data when the process exits normally (rocprofv3 and friends preload a library) get the normal
    exit; so does anybody who sets CUTSEQ_FAST_EXIT=0."""
    traced = any("rocprof" in os.environ.get(k, "").lower() or "roctracer" in os.environ.get(k, "").lower()
                 for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD"))
    if traced or os.environ.get("CUTSEQ_FAST_EXIT", "1") == "0":
        return main()
    code = 0
    try:
        main()
    except SystemExit as exc:
        code = exc.code if isinstance(exc.code, int) else (0 if exc.code is None else 1)
    except BaseException:
        import traceback
        traceback.print_exc()
        code = 1
    sys.stdout.flush()
    sys.stderr.flush()
    logging.shutdown()
    os._exit(code)


if __name__ == "__main__":
    console_main()
