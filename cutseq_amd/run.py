#!/usr/bin/env python3
"""cutseq-compatible command line on top of the HIP trimming engine.

Rewritten counterpart of the reference's ``cutseq/run.py``: same flags (run.py:874-1020), same
preset resolution and upper-casing (1041-1056), same output naming (1058-1107), same single /
paired dispatch (842-863).  Where the reference assembles cutadapt modifiers and calls
``runner.run``, this compiles the op table (``plan.compile_*``) and streams record-aligned FASTQ
chunks through :class:`cutseq_amd.engine.TrimEngine` (one per visible GPU, chunks round-robin,
ordered write-back).  ``-t/--threads`` is accepted for compatibility; the per-read work runs on
the GPU(s).
"""
from __future__ import annotations

import argparse
import logging
import os
import re
import sys
import time
from collections import deque
from typing import List, Optional

from . import __version__, abi, report
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
    p.add_argument("-t", "--threads", type=int, default=1, help="Accepted for compatibility. (Default: 1)")
    p.add_argument("-n", "--dry-run", action="store_true",
                   help="Print the sequence of modifier steps instead of running the pipeline.")
    p.add_argument("-V", "--version", action="version", version=f"%(prog)s {__version__}")
    p.add_argument("--list-adapters", action="store_true",
                   help="List all built-in adapter names and their schemes, then exit.")
    p.add_argument("--cutadapt-selection", choices=["4", "3"], default="4",
                   help="Aligner candidate selection to follow: cutadapt >= 4 (default) or 3.x (DESIGN.md section 0).")
    return p


def _output_names(output_files, input_files, output_prefix, output_suffix):
    """reference: validate_output_file (run.py:1058-1086)."""
    default_format = ".fastq.gz"
    r1_suffix = "_" + output_suffix + "_R1" + default_format
    r2_suffix = "_" + output_suffix + "_R2" + default_format
    if output_files:
        if len(output_files) != len(input_files):
            logging.error(
                f"Number of {output_suffix} output files ({len(output_files)}) must match number of input files ({len(input_files)})."
            )
            sys.exit(1)
        return output_files
    if output_prefix is not None:
        if len(input_files) == 1:
            return [output_prefix + r1_suffix]
        return [output_prefix + r1_suffix, output_prefix + r2_suffix]
    if len(input_files) == 1:
        return [remove_fq_suffix(input_files[0]) + r1_suffix]
    return [remove_fq_suffix(input_files[0]) + r1_suffix, remove_fq_suffix(input_files[1]) + r2_suffix]


def resolve_args(args):
    """Everything main() does between parse_args and run_cutseq (run.py:1029-1107)."""
    if args.list_adapters:
        print_builtin_adapters()
        sys.exit(0)
    if args.input_file is None:
        logging.error("Input file is required.")
        sys.exit(1)
    elif len(args.input_file) > 2:
        logging.error("Input file can not be more than two.")
        sys.exit(1)
    if args.adapter_name is not None:
        if args.adapter_scheme is not None:
            logging.info("Adapter scheme is provided, ignoring adapter name.")
        else:
            args.adapter_scheme = BUILDIN_ADAPTERS.get(args.adapter_name.upper())
            if args.adapter_scheme is None:
                logging.error(f"Adapter name '{args.adapter_name} not found in built-in adapters.")
                args.adapter_scheme = args.adapter_name  # the reference falls back to using the name as scheme
    elif args.adapter_scheme is None:
        logging.error("Adapter scheme or name is required. Use -a or -A.")
        sys.exit(1)
    args.adapter_scheme = args.adapter_scheme.replace(" ", "").upper()
    if len(args.input_file) == 0:
        # the reference raises IndexError here (run.py:1079-1083); same exit class, clearer message
        logging.error("Input file is required.")
        sys.exit(1)
    args.output_file = _output_names(args.output_file, args.input_file, args.output_prefix, "trimmed")
    args.short_file = _output_names(args.short_file, args.input_file, args.output_prefix, "short")
    has_inline = re.match(r".*\([ATGCatgc]+\).*", args.adapter_scheme) is not None
    if args.untrimmed_file or (args.ensure_inline_barcode and has_inline):
        args.untrimmed_file = _output_names(args.untrimmed_file, args.input_file, args.output_prefix, "untrimmed")
    else:
        args.untrimmed_file = [None] * len(args.input_file)
    return args


def settings_from_args(args) -> CutadaptConfig:
    st = CutadaptConfig()
    st.rname_suffix = args.with_rname_suffix  # parsed, stored, never consulted (as in the reference)
    st.ensure_inline_barcode = args.ensure_inline_barcode
    st.trim_polyA = args.trim_polyA
    st.trim_polyA_wo_direction = args.trim_polyA_wo_direction
    st.conditional_cutter = args.conditional_cutter
    st.threads = args.threads
    st.min_length = args.min_length
    st.min_quality = args.min_quality
    st.dry_run = args.dry_run
    st.auto_rc = args.auto_rc
    st.json_file = args.json_file
    st.force_trim_min_length = args.force_trim_min_length
    st.force_anywhere = args.force_anywhere
    st.select_rule = abi.CS_SELECT_LEFTMOST if args.cutadapt_selection == "4" else abi.CS_SELECT_SCORE
    return st


def compile_plan(args, barcode: BarcodeConfig, settings: CutadaptConfig) -> TrimPlan:
    if len(args.input_file) == 1:
        return compile_single(barcode, settings, untrimmed_requested=args.untrimmed_file[0] is not None)
    return compile_paired(
        barcode, settings,
        untrimmed_requested=args.untrimmed_file[0] is not None and args.untrimmed_file[1] is not None)


def dry_run(tp: TrimPlan, barcode: BarcodeConfig):
    if tp.paired:
        for b in ["p5", "p7", "inline5", "inline3", "umi5", "umi3", "mask5", "mask3", "strand"]:
            print(f"{b}: {getattr(barcode, b)}")
        steps1, steps2 = tp.r1.describe(), tp.r2.describe()
        i = 0
        for i, (a, b) in enumerate(zip(steps1, steps2), 1):
            logging.info(f"Step {i}: ({a}, {b})")
        logging.info(f"Step {i + 1}: PairedEndRenamer({'{id}_{r1.cut_prefix}{r2.cut_prefix}' if tp.has_umi else '{id}'!r})")
    else:
        for i, a in enumerate(tp.r1.describe(), 1):
            print(f"Step {i}: {a}")
        print(f"Step {i + 1}: Renamer({'{id}_{cut_prefix}{cut_suffix}' if tp.has_umi else '{id}'!r})")
        if tp.reverse_complement:
            print(f"Step {i + 2}: ReverseComplementConverter()")


def run_pipeline(args, tp: TrimPlan) -> dict:
    """Stream the input through the GPU engine(s); returns the run statistics."""
    from . import capi, fastq
    from .engine import TrimEngine

    n_dev = capi.device_count()
    if n_dev <= 0:
        raise capi.HipUnavailable("no HIP device visible; cutseq_amd has no CPU trimming path")
    want = os.environ.get("CUTSEQ_DEVICES")
    devices = [int(x) for x in want.split(",")] if want else list(range(n_dev))
    paired = tp.paired
    in1 = args.input_file[0]
    in2 = args.input_file[1] if paired else None

    # output files: trimmed / short / untrimmed, per mate; paired --auto-rc on a '-' library swaps
    # the trimmed pair (run.py:787-791)
    def mk(names):
        return [fastq.OutputFile(n) if n else None for n in names]

    trimmed = mk(args.output_file)
    if paired and tp.swap_outputs:
        trimmed = trimmed[::-1]
    outs = [trimmed, mk(args.short_file), mk(args.untrimmed_file)]

    engines = {}
    stride_cap = {}
    slots_per_engine = 2
    inflight = deque()  # (engine, slot, chunk, result arrays): on the GPU
    finishing = deque()  # futures of fastq.finish_chunk, in chunk order
    totals = report.new_totals()
    t0 = time.perf_counter()
    # which (route, mate) streams go to disk, and whether they are gzip
    gz = [[(fh.gz if fh is not None else None) for fh in (group + [None])[:2]] for group in outs]
    pool = fastq._pool()
    max_finishing = fastq.pool_size() + 2  # chunks being formatted / compressed (about 90 MB each)

    def engine_for(dev, stride):
        eng = engines.get(dev)
        if eng is None or stride_cap[dev] < stride:
            if eng is not None:
                eng.close()
            cap = max(stride, 152)
            eng = TrimEngine(tp, device=dev, slots=slots_per_engine, max_reads=fastq.CHUNK_READS, max_stride=cap)
            engines[dev], stride_cap[dev] = eng, cap
        return eng

    def finish(chunk, r1, cap2, r2):
        """Worker thread: records -> bytes on their way to disk; also this chunk's share of the report."""
        try:
            blobs, counts = fastq.finish_chunk(chunk, tp, r1, cap2, r2, gz)
            part = report.new_totals()
            report.account_chunk(part, tp, chunk.len1, r1, chunk.len2 if paired else None, r2 if paired else None)
            part["routes"] = counts
            return blobs, part
        finally:
            chunk.release()

    def reap(block: bool):
        while finishing and (block or finishing[0].done()):
            _, part = finishing.popleft().result()
            report.merge_totals(totals, part)
            block = False

    def drain_one():
        eng, slot, chunk, res = inflight.popleft()
        eng.wait(slot)
        r1, cap2, r2 = res
        fut = pool.submit(finish, chunk, r1, cap2, r2)
        for route in range(3):
            for m in range(2 if paired else 1):
                fh = outs[route][m]
                if fh is not None:
                    fh.write_job(fut, route, m)
        finishing.append(fut)
        reap(len(finishing) >= max_finishing)

    try:
        k = 0
        for chunk in fastq.read_chunks(in1, in2):
            if chunk.stride > abi.CS_MAX_STRIDE:
                raise ValueError(f"reads longer than {abi.CS_MAX_STRIDE} nt are not supported by the GPU tile")
            dev = devices[k % len(devices)]
            slot = (k // len(devices)) % slots_per_engine
            while len(inflight) >= len(devices) * slots_per_engine:
                drain_one()
            # a slot that is still in flight on this engine must be drained before reuse
            while any(e is engines.get(dev) and sl == slot for e, sl, _, _ in inflight):
                drain_one()
            eng = engine_for(dev, chunk.stride)
            res = eng.submit(slot, chunk.seq1, chunk.qual1, chunk.len1, chunk.seq2, chunk.qual2, chunk.len2)
            inflight.append((eng, slot, chunk, res))
            k += 1
        while inflight:
            drain_one()
        while finishing:
            reap(True)
        stats = [e.stats() for e in engines.values()]
    finally:
        for e in engines.values():
            e.close()
        for group in outs:
            for fh in group:
                if fh is not None:
                    fh.close()
    totals["seconds"] = time.perf_counter() - t0
    totals["stats"] = stats
    totals["devices"] = devices
    return totals


def run_cutseq(args):
    barcode = BarcodeConfig(args.adapter_scheme)
    settings = settings_from_args(args)
    tp = compile_plan(args, barcode, settings)
    if settings.dry_run:
        dry_run(tp, barcode)
        return None
    totals = run_pipeline(args, tp)
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


def main(argv: Optional[List[str]] = None):
    parser = build_parser()
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) == 0:
        parser.print_help(sys.stdout)
        sys.exit(0)
    args = resolve_args(parser.parse_args(argv))
    run_cutseq(args)


if __name__ == "__main__":
    main()
