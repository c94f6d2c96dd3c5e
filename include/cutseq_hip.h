/*
 * cutseq_hip.h -- C ABI of the MI355X-native trimming engine (libcutseq_hip.so).
 *
 * The reference (y9c/cutseq) has no FFI: its hot path sits behind cutadapt's Python
 * operator API.  This header is the boundary a maintainer binds
 * with ctypes instead; every entry point names the reference interface it replaces
 * (paths relative to the reference tree).  Plain pointers and sizes only, no
 * exceptions, no Python/torch types.  Return value: 0 = ok, negative = cs_status;
 * cs_last_error() gives the thread-local message.
 *
 * Data contract
 *   A batch is a structure-of-arrays of reads: `seq` and `qual` are row-major byte
 *   matrices [n_reads][stride] (ASCII as in the FASTQ record, stride a multiple of 4,
 *   bytes past len[i] are ignored for the results), `len` holds the read lengths.  Speed note: the scan kernel re-codes
 *   a tile of 64 rows with a fast form that vouches for the bytes A, C, G, T and N and falls back to the exact form,
 *   for good, in a wave that meets any other byte ANYWHERE in its rows (IUPAC codes, lower case, the padding): rows
 *   padded with 'N' behind the read -- what cutseq_amd's own producers write -- run about 2 % faster than rows padded
 *   with zeros.  Results are the same for any padding.  len[i] <= stride is the
 *   caller's contract; the kernel clamps a longer value to stride instead of reading past the row.
 *   Headers never cross the boundary: the device returns, per read, the surviving
 *   interval of the ORIGINAL record plus the location of the captured UMI, and the
 *   host renames/formats (cutadapt's Renamer is string work).
 */
#ifndef CUTSEQ_HIP_H
#define CUTSEQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CS_ABI_VERSION 6
#define CS_MAX_ADAPTER 128 /* longest adapter sequence an op can carry          */
#define CS_MAX_OPS 24      /* longest per-mate op chain                           */
#define CS_MAX_STRIDE 1536 /* longest row the LDS tile can stage (64 rows/block) */
#define CS_LEN_SKIP 0xFFFF /* cs_reads.len[i]: the kernels pass over this read (no result, not counted); the text path
                              marks reads longer than the rows this way and walks their chain in a kernel of their own */
#define CS_MAX_READ (1u << 24) /* longest read the text path takes (cs_text_*): positions are 32-bit there */

typedef enum cs_status {
  CS_OK = 0,
  CS_ERR_ARG = -1,     /* bad argument (null pointer, stride, op table)      */
  CS_ERR_HIP = -2,     /* HIP runtime error, message in cs_last_error()       */
  CS_ERR_NO_GPU = -3,  /* no usable gfx950 device: there is NO CPU fallback   */
  CS_ERR_NOMEM = -4,
  CS_ERR_STATE = -5
} cs_status;

/* ---- op table: what cutseq/run.py:305-433 (single) and :493-731 (paired) compile -- */

enum { CS_OP_ADAPTER = 1, CS_OP_CUT = 2, CS_OP_QTRIM = 3, CS_OP_DEMUX = 4 };

/* CS_OP_DEMUX -- extension, not a reference capability (BASELINE.json config 5): ONE pass
 * decides which of up to 255 equally long inline barcodes starts the read, where the reference would
 * need one `--ensure-inline-barcode` run per barcode, each with AdapterCutter([PrefixAdapter(barcode, 0.2)])
 * (run.py:357-362, 592-597).  A PrefixAdapter match depends on nothing but the first m + k bases of the
 * interval, so the op is a table look-up: entry = outcome of all those PrefixAdapter ops on that prefix,
 * built by running them (cutseq_amd/demux.py drives the device's own adapter op over every possible
 * prefix) and attached with cs_plan_set_demux.  Alphabet {A, C, G, T, other}; prefixes of every length
 * 0 .. m + k (reads shorter than m + k) are covered.  The matched barcode's index goes to cs_reads.bc. */
#define CS_DEMUX_NONE 0xFF      /* cs_reads.bc: no barcode matched                                   */
#define CS_DEMUX_MAX_PREFIX 11  /* m + k <= 11: at most 5^11 + ... table entries (2 bytes each)       */
#define CS_DEMUX_MAX_LONG 24    /* m + k <= 24 with cs_plan_set_demux_ops (no table of every prefix)  */
#define CS_DEMUX_BY_OPS 1       /* cs_op.shortcut of a CS_OP_DEMUX op: take cs_plan_set_demux_ops whatever m + k is
                                   (how cutseq_amd/demux.py builds the TABLES: one pass over every prefix)  */
/* table entry (uint16): [7:0] barcode index or CS_DEMUX_NONE, [11:8] bases the match removes,
 * [14] more than one barcode matched (reported: an exact copy if there is one, else the lowest index) */
#define CS_DEMUX_ENTRY(id, rstop, ambiguous) ((uint16_t)((id) | ((rstop) << 8) | ((ambiguous) ? 0x4000 : 0)))

/* cutadapt aligner flags (cutadapt.align.EndSkip) */
enum { CS_REF_START = 1, CS_QUERY_START = 2, CS_REF_END = 4, CS_QUERY_STOP = 8 };
enum {
  CS_WHERE_BACK = 14,               /* BackAdapter (run.py:346,569,580)              */
  CS_WHERE_FRONT = 11,
  CS_WHERE_PREFIX = 8,              /* PrefixAdapter (run.py:358,593,605)            */
  CS_WHERE_SUFFIX = 2,              /* SuffixAdapter (run.py:365)                    */
  CS_WHERE_FRONT_NOT_INTERNAL = 9,  /* NonInternalFrontAdapter (run.py:400,686,695)  */
  CS_WHERE_BACK_NOT_INTERNAL = 6,   /* NonInternalBackAdapter (run.py:393,679,702)   */
  CS_WHERE_ANYWHERE = 15            /* BackAdapter(force_anywhere=True)              */
};
enum { CS_REMOVE_BEFORE = 0, CS_REMOVE_AFTER = 1 };
/* Exact-substring short cut taken before the aligner.  CS_SHORTCUT_NONE (the default of every
 * op cutseq_amd/plan.py compiles) follows cutadapt >= 3: <Adapter>.match_to goes straight to
 * Aligner.locate (behind a result-neutral k-mer filter).  CS_SHORTCUT_FIND restates cutadapt <= 2.x
 * (str.find / str.rfind first) and is kept as an opt-in so its exposure can be measured
 * (tools/parity_exposure.py). */
enum { CS_SHORTCUT_NONE = 0, CS_SHORTCUT_FIND = 1 };
/* which candidate wins inside Aligner.locate */
enum {
  CS_SELECT_LEFTMOST = 0, /* cutadapt >= 4.0: first hit, replaced only by an overlapping/longer hit of higher score */
  CS_SELECT_SCORE = 1     /* cutadapt 3.x / SURVEY.md appendix B.2: highest score, then fewest errors           */
};

/* read bases as the aligner sees them: <Adapter>.match_to aligns sequence.upper() while the
 * output keeps the original bytes */
enum { CS_CASE_FOLD = 0, CS_CASE_SENSITIVE = 1 };
/* Aligner.locate on a mismatching cell whose insertion (same column, row above) and deletion (same
 * row, previous column) cost the same and less than the diagonal: which one supplies origin/score */
enum {
  CS_TIE_INSERTION = 0, /* `elif cost_insertion <= cost_deletion` first */
  CS_TIE_DELETION = 1   /* the other order */
};

/* per-read result flag bits */
enum {
  CS_F_ADAPTER5 = 0x01,  /* step-2 5' adapter matched (run.py:332-341, 544-563)   */
  CS_F_ADAPTER3 = 0x02,  /* step-3 3' adapter matched (run.py:343-355, 565-590)   */
  CS_F_INLINE = 0x04,    /* inline-barcode adapter matched (run.py:357-370)       */
  CS_F_POLY = 0x08,      /* a poly-A/T adapter matched (run.py:388-413, 673-716)  */
  CS_F_QTRIMMED = 0x10,  /* QualityTrimmer removed >= 1 base                      */
  CS_F_TOO_SHORT = 0x20, /* TooShort(min_length) is true for this mate (run.py:446-451) */
  CS_F_UNTRIMMED = 0x40, /* IsUntrimmedAny is true for this mate (run.py:97-110)  */
  CS_F_AMBIGUOUS = 0x80  /* CS_OP_DEMUX: more than one barcode matched this read   */
};

typedef struct cs_op {
  uint8_t kind;          /* CS_OP_*                                                   */
  uint8_t align_flags;   /* ADAPTER: CS_WHERE_*                                       */
  uint8_t reversed;      /* ADAPTER: 1 = RightmostFrontAdapter: `seq` holds the reversed
                            adapter, the aligner walks the read right-to-left;
                            DEMUX: 1 = the barcodes end the read (SuffixAdapter ops through
                            cs_plan_set_demux_ops: CS_WHERE_SUFFIX, CS_REMOVE_AFTER)     */
  uint8_t remove;        /* ADAPTER: CS_REMOVE_BEFORE (read[rstop:]) / _AFTER (read[:rstart]) */
  uint8_t shortcut;      /* ADAPTER: CS_SHORTCUT_*; DEMUX: CS_DEMUX_BY_OPS = the barcodes' own ops even where a table would fit */
  uint8_t match_flag;    /* ADAPTER: CS_F_* bit OR-ed into the result when it matched */
  uint8_t required;      /* ADAPTER: listed in IsUntrimmedAny -> CS_F_UNTRIMMED if it did not match */
  uint8_t conditional;   /* CUT: ConditionalCutter semantics (run.py:145-161)         */
  uint8_t capture;       /* CUT: 0 none, 1 = cs_result.cap_*, 2 = cs_cap2             */
  uint8_t homopolymer;   /* ADAPTER: filled by cs_plan_create (all bases equal)       */
  uint8_t q_base;        /* QTRIM: quality base (33)                                  */
  uint8_t stat_slot;     /* index into cs_stats.op_matched                            */
  uint16_t m;            /* ADAPTER: sequence length; DEMUX: barcode length           */
  uint16_t k;            /* ADAPTER / DEMUX: int(max_error_rate * m), computed in double on the host */
  uint16_t min_overlap;  /* ADAPTER: already capped to m                              */
  int16_t cut_len;       /* CUT: >0 from the 5' end, <0 from the 3' end               */
  uint16_t force_min_len;/* CUT (conditional): force_trim_min_length                  */
  int16_t q_cutoff;      /* QTRIM: cutoff_back (cutoff_front is 0 in cutseq)          */
  uint8_t seq[CS_MAX_ADAPTER];     /* ADAPTER: upper-cased bases                      */
  uint8_t thr[CS_MAX_ADAPTER + 1]; /* ADAPTER: thr[L] = floor(L * max_error_rate) in IEEE double */
  uint8_t _pad[3];
} cs_op;

typedef struct cs_params {
  uint32_t abi_version;  /* CS_ABI_VERSION                                   */
  uint16_t min_length;   /* TooShort threshold (-m, run.py:943-948)           */
  uint8_t select_rule;   /* CS_SELECT_*                                       */
  uint8_t use_filter;    /* 1 = bit-parallel pre-filter + windowed DP (result-neutral), 0 = full DP */
  uint8_t case_rule;     /* CS_CASE_*  (0 = fold, the cutadapt behaviour)         */
  uint8_t indel_tie;     /* CS_TIE_*                                          */
  uint8_t reserved8[2];
  uint32_t reserved[5];
} cs_params;

/* 8 bytes per read: [start, stop) of the ORIGINAL record survives; cap_* locates the
 * bases a capturing cut removed (the UMI that cutseq's Renamer appends, run.py:378,643). */
typedef struct cs_result {
  uint16_t start;
  uint16_t stop;
  uint16_t cap_off;
  uint8_t cap_len;
  uint8_t flags;
} cs_result;

/* second capture, single-end schemes with a 5' AND a 3' UMI only (run.py:373-378) */
typedef struct cs_cap2 {
  uint16_t off;
  uint8_t len;
  uint8_t _pad;
} cs_cap2;

typedef struct cs_stats {
  uint64_t n_reads;
  uint64_t in_bp;
  uint64_t out_bp;       /* sum of (stop - start) over all reads of the mate   */
  uint64_t qualtrim_bp;  /* QualityTrimmer.trimmed_bases                       */
  uint64_t n_too_short;
  uint64_t n_untrimmed;
  uint64_t n_exact_dp;   /* reads that needed the exact DP (diagnostic)        */
  uint64_t n_refiltered; /* reads the existence-only scan of a rare adapter op could not clear (diagnostic)   */
  uint64_t op_matched[CS_MAX_OPS]; /* AdapterCutter.with_adapters per op         */
} cs_stats;

/* one mate's device- or host-resident arrays */
typedef struct cs_reads {
  const uint8_t *seq;
  const uint8_t *qual;
  const uint16_t *len;
  cs_result *out;
  cs_cap2 *cap2; /* may be NULL */
  uint8_t *bc;   /* may be NULL; CS_OP_DEMUX: index of the barcode that matched, CS_DEMUX_NONE otherwise */
} cs_reads;

typedef struct cs_plan cs_plan;
typedef struct cs_engine cs_engine;

int cs_abi_version(void);
const char *cs_last_error(void);
int cs_device_count(void); /* number of visible HIP devices, <0 on error */

/* Replaces the modifier-list assembly of pipeline_single / pipeline_paired
 * (cutseq/run.py:326-426, 533-731): the caller hands over the compiled chain per mate.
 * n2 == 0 -> single-end.  The plan is immutable and device independent. */
int cs_plan_create(const cs_op *ops_r1, int n1, const cs_op *ops_r2, int n2, const cs_params *params,
                   cs_plan **out);
void cs_plan_destroy(cs_plan *plan);

/* Attach the look-up table of a CS_OP_DEMUX op (mate 1 or 2, op index).  `table` holds one uint16 entry per
 * prefix: first the single entry of the empty prefix, then the 5 prefixes of length 1, ... up to length
 * m + k, each block indexed by sum(digit[t] * 5^t) with digit = 0, 1, 2, 3 for A, C, T, G (bits 2:1 of the
 * ASCII code) and 4 for anything else; `entries` must be (5^(m+k+1) - 1) / 4.  The plan copies the table. */
int cs_plan_set_demux(cs_plan *plan, int mate, int op_index, const uint16_t *table, size_t entries);

/* CS_OP_DEMUX with longer barcodes (CS_DEMUX_MAX_PREFIX < m + k <= CS_DEMUX_MAX_LONG; 10- to 20-base barcodes at the
 * reference's rate of 0.2): a table of every prefix no longer fits, so the op carries the barcodes' own ops instead --
 * (also for shorter ones when the op says CS_DEMUX_BY_OPS)
 * `ops[b]` is the CS_OP_ADAPTER op of barcode b exactly as a single-barcode plan would hold it (CS_WHERE_PREFIX,
 * CS_REMOVE_BEFORE, min_overlap = m, its thr[] table; A/C/G/T only).  The library derives a look-up table over the
 * first min(m + k, 9) bases that names the barcodes which can still match (a superset, from plain edit distance), and
 * the device runs the ops of those candidates on the read and merges the outcomes by the rule of the table entries
 * above.  Reads of such a plan all pass through the resolve kernel: slower than the table form, same results. */
int cs_plan_set_demux_ops(cs_plan *plan, int mate, int op_index, const cs_op *ops, int n_ops);

/* One engine per GPU (one per process in the multi-GPU layout; replaces
 * make_runner(inpaths, cores=threads), run.py:436,753). Uploads the op tables, owns a
 * stream, the device statistics block and `n_slots` staging slots of `max_reads` reads
 * x `max_stride` bytes for cs_trim_batch (0 slots: device-pointer API only). */
int cs_engine_create(const cs_plan *plan, int device, uint32_t n_slots, uint32_t max_reads,
                     uint32_t max_stride, cs_engine **out);
void cs_engine_destroy(cs_engine *eng);

/* The hot path, inputs already resident in HBM: two kernel launches over both mates -- the scan kernel
 * (staging, bit-parallel filters, closed forms, cuts, quality trimming; reads that need the exact DP go
 * to a queue in HBM) and the resolve kernel (strip DP on that queue, then the rest of those reads'
 * chains) -- replacing the per-read modifier loop inside runner.run(pipeline, ...), run.py:473,794.
 * `stream` is a hipStream_t (NULL = the engine's own stream); asynchronous: the results are complete
 * in `stream` order.  r2 == NULL for single-end.  Calls may come on different streams: each call takes
 * one of the engine's lanes (hand-out counters, queue) and first waits for that lane's previous user. */
int cs_trim_device(cs_engine *eng, void *stream, const cs_reads *r1, const cs_reads *r2,
                   uint32_t n_reads, uint32_t stride);

/* The same two launches, pipelined across calls: the scan kernel runs on `stream`, the resolve kernel on
 * the engine's resolve stream behind it, so the NEXT call's scan kernel overlaps THIS call's resolve
 * kernel (a few latency-bound waves that fit beside the scan kernel's).  The results of a pipelined call
 * are complete in `stream` order only after cs_join(eng, stream); keep the result arrays of up to three
 * calls apart (a call waits for the call three before it).  This is how a runner keeps batches in flight
 * (make_runner(cores=N), run.py:436,753) and what bench.py times. */
int cs_trim_device_pipelined(cs_engine *eng, void *stream, const cs_reads *r1, const cs_reads *r2,
                             uint32_t n_reads, uint32_t stride);
/* makes `stream` (NULL = the engine's own) wait for every resolve kernel issued so far */
int cs_join(cs_engine *eng, void *stream);

/* Same path for host buffers (pinned via cs_alloc_pinned for true overlap), pipelined the same way:
 * hipMemcpyAsync H2D -> scan kernel on the engine stream, resolve kernel -> hipMemcpyAsync D2H on the
 * resolve stream, using staging slot `slot`; returns immediately, cs_sync(eng, slot) waits for the
 * results.  With two slots in flight one slot's upload and scan overlap the other's resolve and download. */
int cs_trim_batch(cs_engine *eng, uint32_t slot, const cs_reads *r1, const cs_reads *r2,
                  uint32_t n_reads, uint32_t stride);
int cs_sync(cs_engine *eng, uint32_t slot);

/* Device-side counters (counterpart of cutadapt's Statistics, run.py:473,794).
 * stats[0] = mate 1, stats[1] = mate 2.  Ordered behind every launch the engine has issued so far (on
 * any stream) and synchronous. */
int cs_stats_fetch(cs_engine *eng, cs_stats stats[2], int reset);

/* Timing of the last cs_trim_device / cs_trim_device_pipelined call, measured with HIP events recorded
 * around each kernel on the stream it ran on (ms, the two kernels added).  Synchronises on the stop event. */
int cs_last_kernel_ms(cs_engine *eng, float *ms);
/* the same launch split into its two kernels: ms[0] = scan kernel, ms[1] = resolve kernel */
int cs_last_kernel_split_ms(cs_engine *eng, float ms[2]);
/* Sums of those event-measured durations over every cs_trim_device / cs_trim_device_pipelined call since
 * the last reset (*calls of them): the per-kernel averages bench.py reports.  Waits for the calls in flight. */
int cs_kernel_time_totals(cs_engine *eng, uint32_t *calls, float ms[2], int reset);

/* ---- text path: raw FASTQ text in, finished FASTQ text out -------------------------------
 * Replaces what dnaio / cutadapt.files do around the modifier loop for the reference (record parsing in
 * runner.run's reader, cutseq/run.py:434-441, 751-758; SuffixRemover / Renamer / PairedEndRenamer string work,
 * run.py:330, 378, 537-542, 642-645; TooShort -> IsUntrimmedAny -> sink routing and record formatting,
 * run.py:446-471, 760-793): the host hands over record-aligned text blocks exactly as they come out of the file
 * or the inflate threads, the device finds the records, runs the trimming kernels on them and writes the output
 * records of the three routes; the host only reads / inflates and deflates / writes.
 *
 * A batch is `n_records` complete 4-line records per mate ('\n' or '\r\n' line ends; the last line may lack
 * its line end).  Output per mate: the records of route 0 (trimmed), 1 (too short), 2 (untrimmed) back to back,
 * each route in input order, each record as  @<id>[_<captured bases>]\n<seq[start:stop]>\n+\n<qual[start:stop]>\n.
 * Errors are reported the way the reference's reader would raise them (cs_text_result.error), never ignored. */
enum {
  CS_TEXT_OK = 0,
  CS_TEXT_ERR_MALFORMED = 1,   /* no '@' / '+' line, sequence and quality lengths differ: record `error_record` */
  CS_TEXT_ERR_TOO_LONG = 2,    /* a read is longer than CS_MAX_READ                                            */
  CS_TEXT_ERR_IDS_DIFFER = 3,  /* PairedEndRenamer: "Input read IDs not identical" at record `error_record`    */
  CS_TEXT_ERR_LINE_COUNT = 4   /* the text does not hold 4 * n_records lines                                   */
};

typedef struct cs_text_params {
  uint8_t has_umi;            /* Renamer template carries the captures: {id}_{cut_prefix}{cut_suffix}           */
  uint8_t untrimmed_filter;   /* IsUntrimmedAny filter installed (run.py:453-467, 771-784)                      */
  uint8_t reverse_complement; /* single-end --auto-rc on a '-' library (run.py:420-426)                         */
  uint8_t compress;           /* 1: every route's output leaves the device as ONE gzip member (32 KB deflate blocks with
                                 their own dynamic Huffman codes behind an LZ77 stage -- byte runs and the previous
                                 record's name as matches, bases as literals; the counterpart of xopen's level-1 writer
                                 the reference's OutputFiles use): route_bytes / out_bytes then count compressed bytes  */
  uint32_t max_tag;           /* most bytes a record name can gain: 1 + the plan's capture lengths (0 = no UMI) */
  const char *suffix1[2];     /* SuffixRemover literals of mate 1, applied in order (NULL = none)               */
  const char *suffix2[2];
  uint32_t n_bins;            /* plans with a CS_OP_DEMUX op (table form): the number of barcodes.  The trimmed
                                 records of barcode b then form route 3 + b (route 0 stays empty): 3 + n_bins streams
                                 per mate, back to back in route order; sizes through cs_text_routes.  0: no bins */
  uint8_t fasta_out;          /* records leave as FASTA (">id\nsequence\n"): the input had no qualities -- what
                                 runner.input_file_format().has_qualities() tells OutputFiles, cutseq/run.py:437-441,
                                 754-758 -- or the output files are named .fasta / .fa                              */
  uint8_t _reserved[3];
} cs_text_params;

typedef struct cs_text_result {
  int32_t error;              /* CS_TEXT_*                                                                      */
  uint32_t error_record;      /* first offending record of the batch                                            */
  uint32_t max_len;           /* longest read of the batch                                                      */
  uint32_t n_records;
  uint32_t route_count[3];    /* records (pairs) per route (n_bins > 0: [0] = all barcodes together; likewise below) */
  uint32_t _pad;
  uint64_t route_bytes[3][2]; /* [route][mate]                                                                  */
  uint64_t out_bytes[2];      /* per mate: sum over the routes = what cs_text_fetch copies                      */
  uint64_t written_bp[2];     /* per mate: bases of the records of route 0 (cutadapt's written_bp)              */
  uint32_t n_lines[2];        /* per mate: line ends found in the text (diagnostic for CS_TEXT_ERR_LINE_COUNT)  */
  uint32_t n_long[2];         /* per mate: reads longer than the rows (they took the slow, exact kernel)         */
  uint64_t text_bytes[3][2];  /* compress = 1: uncompressed size of every stream (route_bytes holds the gzip sizes) */
} cs_text_result;

typedef struct cs_text cs_text;

/* `n_slots` batches in flight, each up to `max_text_bytes` of text per mate and `max_records` records, rows of
 * `stride` bytes (multiple of 4, <= CS_MAX_STRIDE).  Uses the engine's plan, streams and statistics block. */
int cs_text_create(cs_engine *eng, const cs_text_params *params, uint32_t n_slots, uint64_t max_text_bytes,
                   uint32_t max_records, uint32_t stride, cs_text **out);
void cs_text_destroy(cs_text *t);
/* Asynchronous: upload (text1 / text2: host memory, pinned for true overlap; they must stay untouched until
 * cs_text_wait returns), record index, trimming kernels, output formatting.  text2 == NULL for single-end. */
int cs_text_submit(cs_text *t, uint32_t slot, const void *text1, uint64_t bytes1, const void *text2, uint64_t bytes2,
                   uint32_t n_records);
/* Blocks until the slot's batch is formatted on the device; sizes and errors in *res. */
int cs_text_wait(cs_text *t, uint32_t slot, cs_text_result *res);
/* Behind cs_text_wait: the sizes of every route stream, 3 + n_bins of them: bytes[route][mate] as cs_text_fetch
 * delivers them (gzip members if compress), text_bytes[route][mate] uncompressed, count[route] records.  The streams of
 * a mate lie in route order in the fetched buffer.  Any of the three pointers may be NULL. */
int cs_text_routes(cs_text *t, uint32_t slot, uint64_t *bytes, uint64_t *text_bytes, uint32_t *count);
/* Copies the output text (res->out_bytes[m] bytes per mate) into the caller's buffers and blocks until it is
 * there; the slot is free for the next cs_text_submit afterwards.  dst2 == NULL for single-end. */
int cs_text_fetch(cs_text *t, uint32_t slot, void *dst1, void *dst2);

void *cs_alloc_pinned(size_t bytes);
/* The same on transparent huge pages (an anonymous mapping, touched, hipHostRegister-ed): 2-3x quicker to get --
 * what a short run feels --, same copy bandwidth; falls back to cs_alloc_pinned.  Freed by cs_free_pinned too. */
void *cs_alloc_pinned_huge(size_t bytes);
void cs_free_pinned(void *p);
void *cs_alloc_device(int device, size_t bytes);
void cs_free_device(int device, void *p);
int cs_copy_to_device(int device, void *dst, const void *src, size_t bytes);
int cs_copy_to_host(int device, void *dst, const void *src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* CUTSEQ_HIP_H */
