"""Text path (``cs_text_*``): FASTQ text in, finished FASTQ text out, all on the device -- against the CPU oracle's
results formatted by the Python record logic (``hostfmt``), byte for byte and route by route.

Reference surface replaced: dnaio's record reader / writer and the name modifiers around cutadapt's modifier loop
(cutseq/run.py:330, 378, 434-441, 537-542, 642-645, 751-758, 785-794).  Edge cases are the ones ``tests/test_host_io.py``
holds for the host parser: CRLF, no final newline, mismatched ids, malformed and truncated records.
"""
import numpy as np
import pytest

from cutseq_amd import abi, plan as planmod, synth, textpath
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig
from cutseq_amd.engine import TrimEngine

import util
from test_oracle import CHAIN_CASES

pytestmark = pytest.mark.gpu


def fastq_text(names, seq, qual, lens, eol=b"\n", final_newline=True):
    recs = []
    for i, name in enumerate(names):
        n = int(lens[i])
        recs.append(b"@" + name + eol + seq[i, :n].tobytes() + eol + b"+" + eol + qual[i, :n].tobytes() + eol)
    body = b"".join(recs)
    return body if final_newline else body[: -len(eol)]


def expected_streams(tp, batch, names1, names2):
    (o1, cap2, _), m2 = util.oracle_run(tp, batch, threads=8)
    recs = util.format_batch(tp, batch, names1, names2, o1, cap2, m2[0] if m2 else None)
    streams = [[b"", b""] for _ in range(3)]
    counts = [0, 0, 0]
    for route, r1, r2 in recs:
        streams[route][0] += r1
        if r2 is not None:
            streams[route][1] += r2
        counts[route] += 1
    return streams, counts


def run_text(tp, text1, text2, n, stride, max_records=None):
    with TrimEngine(tp, device=0, slots=0) as eng:
        with textpath.TextEngine(eng, slots=2, max_text_bytes=max(len(text1), len(text2 or b""), 1) + 1024,
                                 max_records=max_records or max(n, 1), stride=stride) as te:
            return te.run(text1, n, text2)


@pytest.mark.parametrize("name,flags,paired", [c for c in CHAIN_CASES if "shortcut" not in c[1]])
def test_text_path_equals_oracle_plus_record_logic(name, flags, paired):
    scheme = BUILDIN_ADAPTERS.get(name, name)
    st = planmod.CutadaptConfig()
    for k, v in flags.items():
        setattr(st, k, v)
    n = 3000
    batch = synth.generate_pairs(n, 150, scheme, seed=5, chunk_index=len(name), single_end=not paired,
                                 poly_fraction=0.1, art5_fraction=0.02)
    rng = np.random.default_rng(7)
    cut = rng.random(n) < 0.1  # ragged lengths, some empty
    batch.len1[cut] = rng.integers(0, 150, size=int(cut.sum())).astype(np.uint16)
    if paired:
        batch.len2[cut] = rng.integers(0, 150, size=int(cut.sum())).astype(np.uint16)
    tp = util.compile_plan(scheme, st, paired, untrimmed_requested="INLINE" in name)
    names1 = [f"SIM:{i}/1 1:N:0:X".encode() if i % 3 else f"SIM:{i}.1".encode() for i in range(n)]
    names2 = [f"SIM:{i}/2 2:N:0:X".encode() if i % 3 else f"SIM:{i}.2".encode() for i in range(n)] if paired else None
    want, want_counts = expected_streams(tp, batch, names1, names2)
    text1 = fastq_text(names1, batch.seq1, batch.qual1, batch.len1)
    text2 = fastq_text(names2, batch.seq2, batch.qual2, batch.len2) if paired else None
    got, counts = run_text(tp, text1, text2, n, batch.stride)
    assert counts == want_counts
    for route in range(3):
        for m in range(2 if paired else 1):
            assert got[route][m] == want[route][m], (textpath.ROUTES[route], m)


@pytest.mark.parametrize("eol,final_newline", [(b"\n", False), (b"\r\n", True), (b"\r\n", False)])
def test_text_path_line_endings(eol, final_newline):
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    scheme = BUILDIN_ADAPTERS["TAKARAV3"]
    tp = util.compile_plan(scheme, st, True)
    n = 700
    batch = synth.generate_pairs(n, 150, scheme, seed=9)
    names1 = [s.encode() for s in synth.headers(n, 1)]
    names2 = [s.encode() for s in synth.headers(n, 2)]
    want, want_counts = expected_streams(tp, batch, names1, names2)
    text1 = fastq_text(names1, batch.seq1, batch.qual1, batch.len1, eol, final_newline)
    text2 = fastq_text(names2, batch.seq2, batch.qual2, batch.len2, eol, True)
    got, counts = run_text(tp, text1, text2, n, batch.stride)
    assert counts == want_counts and got == want


def test_text_path_fixture_and_several_batches_in_flight():
    """The reference's own reads (fixture10k) in batches of 2 500 through three slots; different batch sizes."""
    rec1 = util.read_fastq_gz(util.GOLDEN / "fixture10k_R1.fq.gz")
    rec2 = util.read_fastq_gz(util.GOLDEN / "fixture10k_R2.fq.gz")
    batch = util.batch_from_records(rec1, rec2)
    st = planmod.CutadaptConfig()
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    names1, names2 = [r[0] for r in rec1], [r[0] for r in rec2]
    want, want_counts = expected_streams(tp, batch, names1, names2)
    sizes = [2500, 2500, 1, 3999, 1000]
    texts, lo = [], 0
    for size in sizes:
        hi = lo + size
        texts.append((fastq_text(names1[lo:hi], batch.seq1[lo:hi], batch.qual1[lo:hi], batch.len1[lo:hi]),
                      fastq_text(names2[lo:hi], batch.seq2[lo:hi], batch.qual2[lo:hi], batch.len2[lo:hi]), size))
        lo = hi
    assert lo == len(rec1)
    got = [[b"", b""] for _ in range(3)]
    counts = [0, 0, 0]
    with TrimEngine(tp, device=0, slots=0) as eng:
        with textpath.TextEngine(eng, slots=3, max_text_bytes=max(len(t[0]) for t in texts) + 4096, max_records=4000,
                                 stride=batch.stride) as te:
            pending = []

            def finish(slot):
                res = te.wait(slot)
                out = [np.empty(max(int(res.out_bytes[m]), 1), dtype=np.uint8) for m in range(2)]
                te.fetch(slot, out[0], out[1])
                part = textpath.split_routes(res, out, True)
                for route in range(3):
                    counts[route] += int(res.route_count[route])
                    for m in range(2):
                        got[route][m] += part[route][m]

            for k, (t1, t2, size) in enumerate(texts):
                if len(pending) == 3:
                    finish(pending.pop(0))
                te.submit(k % 3, t1, len(t1), t2, len(t2), size)
                pending.append(k % 3)
            while pending:
                finish(pending.pop(0))
            st1, st2 = eng.stats()
    assert counts == want_counts and sum(counts) == len(rec1)
    assert got == want
    assert int(st1.n_reads) == len(rec1) == int(st2.n_reads)


def test_text_path_errors_are_the_readers_errors():
    st = planmod.CutadaptConfig()
    tp = util.compile_plan(BUILDIN_ADAPTERS["TAKARAV3"], st, True)
    good1 = b"@r1 1\nACGTACGTAC\n+\nIIIIIIIIII\n@r2 1\nACGTACGTAC\n+\nIIIIIIIIII\n"
    good2 = b"@r1 2\nACGTACGTAC\n+\nIIIIIIIIII\n@r2 2\nACGTACGTAC\n+\nIIIIIIIIII\n"
    with TrimEngine(tp, device=0, slots=0) as eng:
        with textpath.TextEngine(eng, slots=1, max_text_bytes=4096, max_records=16, stride=12) as te:
            streams, counts = te.run(good1, 2, good2)
            assert sum(counts) == 2
            # mate ids differ in the second record (PairedEndRenamer: "Input read IDs not identical")
            with pytest.raises(textpath.TextFormatError) as e:
                te.run(good1, 2, good2.replace(b"@r2 2", b"@rX 2"))
            assert e.value.code == abi.CS_TEXT_ERR_IDS_DIFFER and e.value.record == 1
            # ... but a trailing /1 vs /2 (or 1 vs 2) is the same id
            te.run(good1.replace(b"@r2 1", b"@r2/1 x"), 2, good2.replace(b"@r2 2", b"@r2/2 y"))
            # no '@', no '+', lengths differ
            for bad in (good1.replace(b"@r2", b"r2"), good1.replace(b"+\nIIIIIIIIII\n@r2", b"-\nIIIIIIIIII\n@r2"),
                        good1[:-3] + b"\n"):
                with pytest.raises(textpath.TextFormatError) as e:
                    te.run(bad, 2, good2)
                assert e.value.code == abi.CS_TEXT_ERR_MALFORMED
            # a truncated block: fewer lines than announced
            with pytest.raises(textpath.TextFormatError) as e:
                te.run(good1[: len(good1) // 2 + 3], 2, good2)
            assert e.value.code in (abi.CS_TEXT_ERR_LINE_COUNT, abi.CS_TEXT_ERR_MALFORMED)
            # a read longer than the rows (12 here) is no error: it takes the long-read kernel
            long1 = b"@r1 1\n" + b"ACGTTGCA" * 5 + b"\n+\n" + b"I" * 40 + b"\n"
            s_long, c_long = te.run(long1, 1, b"@r1 2\nACGT\n+\nIIII\n")
            assert sum(c_long) == 1 and b"ACGTTGCA" in b"".join(x[0] for x in s_long)
            # the engine is still usable afterwards
            streams2, counts2 = te.run(good1, 2, good2)
            assert streams2 == streams and counts2 == counts
            # empty batch
            streams3, counts3 = te.run(b"", 0, b"")
            assert counts3 == [0, 0, 0] and streams3 == [[b"", b""]] * 3


@pytest.mark.parametrize("lz", [True, False])
@pytest.mark.parametrize("n", [1, 40, 3000, 60_000])
def test_device_compressed_output_is_gzip_of_the_same_text(n, lz, monkeypatch):
    """cs_text_params.compress: every route's output as ONE gzip member written on the device (32 KB deflate blocks with
    their own Huffman codes -- runs and the previous record's name as LZ77 matches, CUTSEQ_GPU_LZ=0: literals only --,
    stored blocks where that does not pay: the one-record case) -- any inflater must give back exactly the text the
    uncompressed form returns, CRC-32 and ISIZE included (Python's gzip checks both).  The ratio is pinned: the
    counterpart of xopen's level-1 writer must not quietly get worse (zlib level 1 on the same text is within a few
    per cent of the LZ form, profiles/r04_gzip_ratio.json)."""
    import gzip
    import zlib
    if not lz:
        monkeypatch.setenv("CUTSEQ_GPU_LZ", "0")
    scheme = BUILDIN_ADAPTERS["TAKARAV3"]
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    st.min_length = 60 if n > 100 else 20  # a well-filled "short" route as well
    tp = util.compile_plan(scheme, st, True)
    batch = synth.generate_pairs(n, 150, scheme, seed=3, adapter_fraction=0.6)
    names1 = [s.encode() for s in synth.headers(n, 1)]
    names2 = [s.encode() for s in synth.headers(n, 2)]
    text1 = fastq_text(names1, batch.seq1, batch.qual1, batch.len1)
    text2 = fastq_text(names2, batch.seq2, batch.qual2, batch.len2)
    plain, counts = run_text(tp, text1, text2, n, batch.stride)
    with TrimEngine(tp, device=0, slots=0) as eng:
        with textpath.TextEngine(eng, slots=2, max_text_bytes=len(text1) + 1024, max_records=n, stride=batch.stride,
                                 compress=True) as te:
            for _ in range(2):  # (slot state from the first batch must not leak into the second)
                packed, counts2 = te.run(text1, n, text2)
                assert counts2 == counts
                for route in range(3):
                    for m in range(2):
                        if plain[route][m]:
                            assert packed[route][m][:4] == b"\x1f\x8b\x08\x00"
                            assert gzip.decompress(packed[route][m]) == plain[route][m], (route, m)
                        else:
                            assert packed[route][m] == b""
    if n >= 3000:
        raw = sum(len(x) for row in plain for x in row)
        ratio = raw / sum(len(x) for row in packed for x in row)
        level1 = raw / sum(len(zlib.compress(x, 1)) + 12 for row in plain for x in row if x)
        if lz:
            assert ratio > 0.93 * level1, (ratio, level1)  # within a few per cent of `gzip -1`
        else:
            assert ratio > 2.5, ratio  # Huffman-only: ~3 bits per base, 2 per quality value, names a little over 4 per byte


def test_device_gzip_of_tiny_and_repetitive_records():
    """The device's LZ77 stage at its edges: records of a few bytes (thousands of line ends per 32 KB chunk: more than the
    chunk's line index holds), empty reads, long homopolymer / single-quality runs that cross the 128-byte slices and the
    32 KB chunks, names that share nothing and names that are identical.  Whatever the matcher finds, zlib must give
    back the uncompressed form's bytes."""
    import gzip
    import random
    rng = random.Random(19)
    scheme = "ACACGACGCTCTTCCGATCT>AGATCGGAAGAGCACACGTC"  # no UMI, no masks: names and reads pass through
    st = planmod.CutadaptConfig()
    st.min_length = 0
    st.min_quality = 0
    tp = util.compile_plan(scheme, st, False)
    for style in ("tiny", "runs", "same-names"):
        reads, names = [], []
        for i in range(12_000 if style == "tiny" else 3_000):
            if style == "tiny":
                n = rng.choice([0, 0, 1, 2, 3])
                reads.append((util.random_dna(rng, n), "I" * n))
                names.append(b"r%d" % i)
            elif style == "runs":
                n = rng.choice([150, 150, 149, 37])
                base = rng.choice("ACGT")
                reads.append((base * n, rng.choice("I9-") * n))
                names.append(b"%x" % rng.getrandbits(64))
            else:
                reads.append((util.random_dna(rng, 150), "I" * 150))
                names.append(b"INSTRUMENT:123:FLOWCELL:1:1101:10000:20000 1:N:0:ACGT")
        batch = util.batch_from_reads(reads)
        text = fastq_text(names, batch.seq1, batch.qual1, batch.len1)
        n = len(reads)
        plain, counts = run_text(tp, text, None, n, batch.stride)
        with TrimEngine(tp, device=0, slots=0) as eng:
            with textpath.TextEngine(eng, slots=1, max_text_bytes=len(text) + 1024, max_records=n, stride=batch.stride,
                                     compress=True) as te:
                packed, counts2 = te.run(text, n, None)
        assert counts2 == counts
        for route in range(3):
            if plain[route][0]:
                assert gzip.decompress(packed[route][0]) == plain[route][0], (style, route)
        if style != "tiny":
            assert sum(len(x[0]) for x in packed) * 4 < sum(len(x[0]) for x in plain), style  # runs and names do compress


@pytest.mark.parametrize("compress,paired,length,short_rows", [(False, True, 8, False), (True, True, 8, False), (False, False, 8, False),
                                                              (False, True, 12, False), (True, True, 8, True), (False, True, 12, True)])
def test_text_path_demultiplexes_into_one_route_per_barcode(compress, paired, length, short_rows):
    """cs_text_params.n_bins: the trimmed records of barcode b leave the device as route 3 + b (plain text or one gzip
    member per route), in input order; short and untrimmed pairs keep routes 1 and 2.  Expected: the array API's
    results and barcode indices for the same reads (held to the oracle by tests/test_gpu_demux.py), formatted by the
    record logic.  12-base barcodes: the op's other form (every read through the resolve kernel).  ``short_rows``: rows
    of 64 bases for 150-base reads, so every read takes the long-read kernel, which knows both forms too."""
    import gzip
    import random
    import hostfmt
    from test_gpu_demux import barcode_set, plant_barcodes, scheme_with
    rng = random.Random(77)
    codes = barcode_set(rng, 24, length, 4)
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    st.min_length = 40
    st.demux_barcodes = codes
    n = 3_000 if short_rows else 20_000
    batch = synth.generate_pairs(n, 150, scheme_with(codes[0]), seed=8, adapter_fraction=0.5, single_end=not paired)
    plant_barcodes(rng, batch, codes, length)
    tp = (planmod.compile_paired if paired else planmod.compile_single)(BarcodeConfig(scheme_with(codes[0])), st)
    names1 = [s.encode() for s in synth.headers(n, 1)]
    names2 = [s.encode() for s in synth.headers(n, 2)]
    text1 = fastq_text(names1, batch.seq1, batch.qual1, batch.len1)
    text2 = fastq_text(names2, batch.seq2, batch.qual2, batch.len2) if paired else None
    bc = np.empty(n, dtype=np.uint8)
    with TrimEngine(tp, device=0, slots=1, max_reads=n, max_stride=batch.stride) as eng:
        r1, cap2, r2 = eng.submit(0, batch.seq1, batch.qual1, batch.len1, batch.seq2, batch.qual2, batch.len2, bc=bc)
        eng.wait(0)
    want = [[b"", b""] for _ in range(3 + len(codes))]
    want_counts = [0] * (3 + len(codes))
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
            assert bc[i] != abi.CS_DEMUX_NONE
            route = 3 + int(bc[i])
        want[route][0] += rec1
        want[route][1] += rec2
        want_counts[route] += 1
    assert sum(1 for c in want_counts[3:] if c) == len(codes) and want_counts[1] > n // 200 and want_counts[2] > n // 200
    with TrimEngine(tp, device=0, slots=0) as eng:
        with textpath.TextEngine(eng, slots=2, max_text_bytes=len(text1) + 1024, max_records=n,
                                 stride=64 if short_rows else batch.stride, compress=compress, bins=len(codes)) as te:
            for _ in range(2):  # (a slot's second batch: the route block is reset)
                got, counts = te.run(text1, n, text2)
                assert counts == want_counts
                for route in range(3 + len(codes)):
                    for m in range(2):
                        data = got[route][m]
                        if compress and data:
                            data = gzip.decompress(data)
                        assert data == want[route][m], (route, m)
            res = None
            te.submit(1, text1, len(text1), text2, len(text2) if paired else 0, n)
            res = te.wait(1)
            assert int(res.route_count[0]) == sum(want_counts[3:]) and int(res.route_count[1]) == want_counts[1]
            out = [np.empty(max(int(res.out_bytes[m]), 1), dtype=np.uint8) for m in range(2)]
            te.fetch(1, out[0], out[1] if paired else None)


@pytest.mark.parametrize("style", ["every byte value", "fibonacci frequencies", "two symbols"])
def test_device_gzip_code_construction_at_its_edges(style):
    """The device's Huffman construction (deflate_kernels.hip.inc, huff_build: the whole block builds the code -- compaction
    by a scan, rank sort, two-queue merge, length limit on the histogram, canonical codes from per-wave ballots) away from
    FASTQ's few dozen symbols: quality lines that use every byte value but the line ends (more than 256 used literal /
    length symbols: both symbols of a thread), frequencies that grow like Fibonacci numbers (an unlimited code would be
    twenty bits deep: the 15-bit limit is enforced), and a text of two symbols (codes of one bit).  Any inflater must
    give back the uncompressed form's bytes."""
    import gzip
    import random
    rng = random.Random(31)
    scheme = "ACACGACGCTCTTCCGATCT>AGATCGGAAGAGCACACGTC"
    st = planmod.CutadaptConfig()
    st.min_length = 0
    st.min_quality = -1000  # nothing is quality-trimmed, whatever the bytes say
    tp = util.compile_plan(scheme, st, False)
    n, L = 4000, 148
    if style == "every byte value":
        alphabet = [b for b in range(1, 256) if b not in (10, 13)]
        weights = [1] * len(alphabet)
    elif style == "fibonacci frequencies":
        alphabet = list(range(40, 40 + 22))
        weights = [1, 1]
        while len(weights) < len(alphabet):
            weights.append(weights[-1] + weights[-2])
    else:
        alphabet, weights = [ord("I")], [1]
    seq = np.full((n, 152), ord("A") if style == "two symbols" else 0, dtype=np.uint8)
    qual = np.zeros((n, 152), dtype=np.uint8)
    for i in range(n):
        if style != "two symbols":
            seq[i, :L] = np.frombuffer(bytes(rng.choices(b"ACGT", k=L)), dtype=np.uint8)
        qual[i, :L] = np.frombuffer(bytes(rng.choices(alphabet, weights=weights, k=L)), dtype=np.uint8)
    lens = np.full(n, L, dtype=np.uint16)
    names = [b"A" if style == "two symbols" else b"q%d" % i for i in range(n)]
    text = fastq_text(names, seq, qual, lens)
    plain, counts = run_text(tp, text, None, n, 152)
    with TrimEngine(tp, device=0, slots=0) as eng:
        with textpath.TextEngine(eng, slots=1, max_text_bytes=len(text) + 1024, max_records=n, stride=152, compress=True) as te:
            packed, counts2 = te.run(text, n, None)
    assert counts2 == counts and sum(counts) == n
    for route in range(3):
        if plain[route][0]:
            assert gzip.decompress(packed[route][0]) == plain[route][0], (style, route)
    if style == "fibonacci frequencies":
        assert len(packed[0][0]) < 0.8 * len(plain[0][0])  # (a limited code still compresses a skewed source)
