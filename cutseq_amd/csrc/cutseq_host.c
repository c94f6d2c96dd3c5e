/*
 * cutseq_host.c -- host-side native helpers (plain C, no GPU): the seeded synthetic read
 * generator used by bench.py and the tests.  Built into libcutseq_host.so by
 * cutseq_amd/build.py.  Nothing here trims reads.
 *
 * Generator model.  Library molecule, top strand:
 *     P5 | inline5 umi5 mask5 | insert | mask3 umi3 inline3 | P7
 *     R1 = [5' artefact P5] head insert tail p7.fw filler...
 *     R2 = [5' artefact rc(P7)] rc(tail) rc(insert) rc(head) p5.rc filler...
 * Every pair draws from its own counter-based stream (splitmix64 keyed by seed and the
 * global pair index), so any chunking / threading / rank split yields the same bytes.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/cutseq_synth.h" /* csh_synth_params: one definition for the host and the device form */

typedef struct {
  uint64_t s;
} rng_t;

static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint64_t rng_next(rng_t *r) {
  r->s += 0x9E3779B97F4A7C15ull;
  return mix64(r->s);
}
static inline uint32_t rng_below(rng_t *r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }
static inline double rng_unit(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

static const char BASES[4] = {'A', 'C', 'G', 'T'};
static inline uint8_t comp(uint8_t c) {
  switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return c;
  }
}

#define MAX_TPL 4096

/* append helpers on a bounded template buffer */
static inline int put(uint8_t *t, int n, const uint8_t *src, int len) {
  if (len > MAX_TPL - n) len = MAX_TPL - n;
  if (len > 0) memcpy(t + n, src, (size_t)len);
  return n + (len > 0 ? len : 0);
}
static inline int put_rc(uint8_t *t, int n, const uint8_t *src, int len) {
  for (int i = len - 1; i >= 0 && n < MAX_TPL; i--) t[n++] = comp(src[i]);
  return n;
}

static void finish_read(rng_t *r, const csh_synth_params *p, uint8_t *tpl, int tlen, int ad_lo, int ad_hi,
                        uint8_t *seq, uint8_t *qual, uint16_t *len) {
  const int L = (int)p->read_len;
  /* one single-base indel inside the adapter region of a few adapter-bearing reads */
  if (ad_lo < L && ad_hi > ad_lo && rng_unit(r) < p->indel_frac) {
    int hi = ad_hi < L ? ad_hi : L;
    int d = ad_lo + (int)rng_below(r, (uint32_t)(hi - ad_lo));
    if (rng_next(r) & 1) { /* deletion */
      memmove(tpl + d, tpl + d + 1, (size_t)(tlen - d - 1));
      tlen--;
    } else if (tlen < MAX_TPL) { /* insertion */
      memmove(tpl + d + 1, tpl + d, (size_t)(tlen - d));
      tpl[d] = (uint8_t)BASES[rng_below(r, 4)];
      tlen++;
    }
  }
  /* degrading quality tail on 20 % of the reads */
  int tail_from = L + 1;
  if (rng_unit(r) < 0.20) tail_from = (int)(L * 0.6) + (int)rng_below(r, (uint32_t)(L - (int)(L * 0.6)) + 1u);
  for (int i = 0; i < L; i++) {
    uint8_t c = i < tlen ? tpl[i] : (uint8_t)BASES[rng_below(r, 4)];
    uint64_t x = rng_next(r);
    double u = (double)(x >> 40) * (1.0 / 16777216.0);           /* 24 bits */
    double w = (double)((x >> 16) & 0xFFFFFF) * (1.0 / 16777216.0); /* 24 bits */
    if (u < p->sub_rate) c = (uint8_t)BASES[x & 3];
    uint8_t q = 'I';
    if (w < 0.08)
      q = '-';
    else if (w < 0.16)
      q = '9';
    if (i >= tail_from) {
      double t = (double)(rng_next(r) >> 40) * (1.0 / 16777216.0);
      if (t < 0.55)
        q = '#';
      else if (t < 0.80)
        q = '-';
    }
    if (u >= p->sub_rate && u < p->sub_rate + p->n_rate) {
      c = 'N';
      q = '#';
    }
    seq[i] = c;
    qual[i] = q;
  }
  /* padding behind the read: 'N' in the sequence row (the scan kernel's fast re-coding vouches for A, C, G, T, N only:
     trim_kernel.hip.inc, encode4_pairs_fast; any padding is correct, this one is the fast one), 0 in the quality row */
  for (uint32_t i = (uint32_t)L; i < p->stride; i++) {
    seq[i] = 'N';
    qual[i] = 0;
  }
  *len = (uint16_t)L;
}

static void synth_one(const csh_synth_params *p, uint64_t index, uint8_t *seq1, uint8_t *qual1, uint16_t *len1,
                      uint8_t *seq2, uint8_t *qual2, uint16_t *len2) {
  rng_t r;
  r.s = mix64(p->seed ^ mix64(index + 0x632BE59BD9B4E019ull));
  const int L = (int)p->read_len;
  const int i5 = (int)strlen(p->inline5), i3 = (int)strlen(p->inline3);
  const int head_len = i5 + p->umi5 + p->mask5, tail_len = p->mask3 + p->umi3 + i3;
  const int max_ins = L + 40;
  /* insert length: adapter-bearing reads skew towards the read end */
  const int full = L - head_len; /* insert length at which the 3' structure just leaves R1 */
  int ins_len;
  double u = rng_unit(&r), v = rng_unit(&r);
  if (u < p->adapter_fraction) { /* whole 3' adapter inside R1, skewed towards the read end */
    int vis = full - tail_len - (int)strlen(p->p7_fw);
    int span = vis - 25;
    if (span < 1) span = 1;
    ins_len = vis - (int)(v * v * span);
  } else if (u < p->adapter_fraction + p->partial_fraction) {
    ins_len = full - tail_len - 3 - (int)rng_below(&r, 17);
  } else {
    ins_len = full + (int)rng_below(&r, 40);
  }
  if (ins_len < 1) ins_len = 1;
  if (ins_len > max_ins) ins_len = max_ins;

  uint8_t head[1024], tail[1024], insert[MAX_TPL];
  int h = 0, t = 0;
  if (head_len > 1000 || tail_len > 1000 || ins_len > MAX_TPL - 64) return;
  h = put(head, h, (const uint8_t *)p->inline5, i5);
  for (int i = 0; i < p->umi5 + p->mask5; i++) head[h++] = (uint8_t)BASES[rng_below(&r, 4)];
  for (int i = 0; i < p->mask3 + p->umi3; i++) tail[t++] = (uint8_t)BASES[rng_below(&r, 4)];
  memcpy(tail + t, p->inline3, (size_t)i3);
  t += i3;
  for (int i = 0; i < ins_len; i++) insert[i] = (uint8_t)BASES[rng_below(&r, 4)];
  if (p->strand != 0 && rng_unit(&r) < p->poly_fraction) {
    int pl = 10 + (int)rng_below(&r, 31);
    if (pl > ins_len) pl = ins_len;
    if (p->strand > 0)
      memset(insert + ins_len - pl, 'A', (size_t)pl);
    else
      memset(insert, 'T', (size_t)pl);
  }

  uint8_t tpl[MAX_TPL];
  int n = 0, ad_lo, ad_hi;
  /* mate 1 */
  if (rng_unit(&r) < p->art5_fraction) n = put(tpl, n, (const uint8_t *)p->p5_fw, (int)strlen(p->p5_fw));
  n = put(tpl, n, head, h);
  n = put(tpl, n, insert, ins_len);
  n = put(tpl, n, tail, t);
  ad_lo = n;
  n = put(tpl, n, (const uint8_t *)p->p7_fw, (int)strlen(p->p7_fw));
  ad_hi = n;
  finish_read(&r, p, tpl, n, ad_lo, ad_hi, seq1, qual1, len1);
  if (p->single_end) return;
  /* mate 2 reads the bottom strand */
  n = 0;
  if (rng_unit(&r) < p->art5_fraction) n = put(tpl, n, (const uint8_t *)p->p7_rc, (int)strlen(p->p7_rc));
  n = put_rc(tpl, n, tail, t);
  n = put_rc(tpl, n, insert, ins_len);
  n = put_rc(tpl, n, head, h);
  ad_lo = n;
  n = put(tpl, n, (const uint8_t *)p->p5_rc, (int)strlen(p->p5_rc));
  ad_hi = n;
  finish_read(&r, p, tpl, n, ad_lo, ad_hi, seq2, qual2, len2);
}

typedef struct {
  const csh_synth_params *p;
  uint32_t lo, hi;
  uint8_t *seq1, *qual1, *seq2, *qual2;
  uint16_t *len1, *len2;
} synth_job;

static void *synth_worker(void *arg) {
  synth_job *j = (synth_job *)arg;
  const size_t st = j->p->stride;
  for (uint32_t i = j->lo; i < j->hi; i++)
    synth_one(j->p, j->p->first_index + i, j->seq1 + i * st, j->qual1 + i * st, j->len1 + i,
              j->seq2 ? j->seq2 + i * st : NULL, j->qual2 ? j->qual2 + i * st : NULL, j->len2 ? j->len2 + i : NULL);
  return NULL;
}

int csh_synth_pairs(const csh_synth_params *p, uint32_t n, uint8_t *seq1, uint8_t *qual1, uint16_t *len1,
                    uint8_t *seq2, uint8_t *qual2, uint16_t *len2, int n_threads) {
  if (!p || !seq1 || !qual1 || !len1) return -1;
  if (p->read_len == 0 || p->read_len > p->stride || p->read_len > 2000) return -1;
  if (!p->single_end && (!seq2 || !qual2 || !len2)) return -1;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 128) n_threads = 128;
  synth_job jobs[128];
  pthread_t tids[128];
  uint32_t chunk = (n + (uint32_t)n_threads - 1) / (uint32_t)n_threads;
  for (int t = 0; t < n_threads; t++) {
    uint32_t lo = (uint32_t)t * chunk, hi = lo + chunk;
    if (lo > n) lo = n;
    if (hi > n) hi = n;
    jobs[t] = (synth_job){p, lo, hi, seq1, qual1, seq2, qual2, len1, len2};
    pthread_create(&tids[t], NULL, synth_worker, &jobs[t]);
  }
  for (int t = 0; t < n_threads; t++) pthread_join(tids[t], NULL);
  return 0;
}

int csh_abi_version(void) { return 1; }

/* ===========================================================================================
 * FASTQ chunk parser / formatter (host side of tier E; counterpart of dnaio's reader/writer
 * that cutadapt drives for the reference, cutseq/run.py:434-441, 751-758).  String work only.
 * =========================================================================================== */

/* Count complete 4-line records at the start of buf (at most max_records).
 * Returns the number of records; *consumed = bytes they occupy, *max_len = longest sequence line.
 * A record is complete when its 4th line ends with '\n' or (at_eof) at the end of the buffer. */
int64_t csh_fastq_count(const uint8_t *buf, int64_t n, int64_t max_records, int at_eof, int64_t *consumed,
                        int32_t *max_len) {
  int64_t pos = 0, rec = 0, rec_start = 0;
  int32_t longest = 0;
  while (rec < max_records && pos < n) {
    int64_t p = pos;
    int32_t seq_len = 0;
    int ok = 1;
    for (int line = 0; line < 4; line++) {
      const uint8_t *nl = (const uint8_t *)memchr(buf + p, '\n', (size_t)(n - p));
      int64_t end;
      if (nl) {
        end = nl - buf;
      } else if (at_eof && line == 3 && p < n) {
        end = n; /* last line without trailing newline */
      } else {
        ok = 0;
        break;
      }
      if (line == 1) {
        int64_t l = end - p;
        if (l > 0 && buf[end - 1] == '\r') l--;
        seq_len = (int32_t)l;
      }
      p = nl ? end + 1 : end;
    }
    if (!ok) break;
    if (seq_len > longest) longest = seq_len;
    rec++;
    pos = p;
    rec_start = pos;
  }
  *consumed = rec_start;
  *max_len = longest;
  return rec;
}

/* Parse exactly n_records records into SoA rows.  name_off/name_len locate the header (without
 * '@', without line end) inside buf.  Returns bytes consumed, or -(record index + 1) on a
 * malformed record (missing '@' / '+', sequence and quality lengths differ, longer than stride). */
int64_t csh_fastq_parse(const uint8_t *buf, int64_t n, int64_t n_records, uint32_t stride, uint8_t *seq,
                        uint8_t *qual, uint16_t *len, int64_t *name_off, int32_t *name_len) {
  int64_t pos = 0;
  for (int64_t r = 0; r < n_records; r++) {
    int64_t starts[4], ends[4];
    for (int line = 0; line < 4; line++) {
      if (pos > n) return -(r + 1);
      const uint8_t *nl = pos < n ? (const uint8_t *)memchr(buf + pos, '\n', (size_t)(n - pos)) : NULL;
      int64_t end = nl ? (nl - buf) : n;
      starts[line] = pos;
      ends[line] = end;
      if (ends[line] > starts[line] && buf[ends[line] - 1] == '\r') ends[line]--;
      pos = nl ? end + 1 : end;
    }
    if (ends[0] <= starts[0] || buf[starts[0]] != '@') return -(r + 1);
    if (ends[2] <= starts[2] || buf[starts[2]] != '+') return -(r + 1);
    int64_t sl = ends[1] - starts[1], ql = ends[3] - starts[3];
    if (sl != ql || sl > (int64_t)stride || sl > 65535) return -(r + 1);
    uint8_t *srow = seq + (size_t)r * stride, *qrow = qual + (size_t)r * stride;
    memcpy(srow, buf + starts[1], (size_t)sl);
    memcpy(qrow, buf + starts[3], (size_t)sl);
    if ((uint32_t)sl < stride) {
      memset(srow + sl, 0, stride - (size_t)sl);
      memset(qrow + sl, 0, stride - (size_t)sl);
    }
    len[r] = (uint16_t)sl;
    name_off[r] = starts[0] + 1;
    name_len[r] = (int32_t)(ends[0] - starts[0] - 1);
  }
  return pos;
}

typedef struct csh_result {
  uint16_t start, stop, cap_off;
  uint8_t cap_len, flags;
} csh_result;
typedef struct csh_cap2 {
  uint16_t off;
  uint8_t len, pad;
} csh_cap2;

typedef struct csh_format_params {
  int32_t paired;
  int32_t has_umi;          /* Renamer template carries the captured bases */
  int32_t untrimmed_filter; /* IsUntrimmedAny filter installed */
  int32_t reverse_complement; /* single-end --auto-rc on a '-' library */
  uint8_t flag_too_short, flag_untrimmed, pad[2];
  const char *suffix1[2]; /* SuffixRemover literals of mate 1, applied in order */
  const char *suffix2[2];
} csh_format_params;

static int is_space(uint8_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

/* SuffixRemover chain on the whole header */
static int32_t strip_suffixes(const uint8_t *name, int32_t n, const char *const suf[2]) {
  for (int i = 0; i < 2; i++) {
    if (!suf[i]) continue;
    int32_t sl = (int32_t)strlen(suf[i]);
    if (sl && n >= sl && memcmp(name + n - sl, suf[i], (size_t)sl) == 0) n -= sl;
  }
  return n;
}
/* Renamer.parse_name: name.split(maxsplit=1)[0] if there are two fields, else the whole name */
static void read_id(const uint8_t *name, int32_t n, int32_t *off, int32_t *len) {
  int32_t a = 0;
  while (a < n && is_space(name[a])) a++;
  int32_t b = a;
  while (b < n && !is_space(name[b])) b++;
  int32_t c = b;
  while (c < n && is_space(name[c])) c++;
  if (b > a && c < n) { /* two fields */
    *off = a;
    *len = b - a;
  } else {
    *off = 0;
    *len = n;
  }
}
/* dnaio.record_names_match: ids up to the first space/tab, a trailing 1/2/3 ignored */
static int ids_match(const uint8_t *n1, int32_t l1, const uint8_t *n2, int32_t l2) {
  int32_t a = 0, b = 0;
  while (a < l1 && n1[a] != ' ' && n1[a] != '\t') a++;
  while (b < l2 && n2[b] != ' ' && n2[b] != '\t') b++;
  if (a && b && n1[a - 1] >= '1' && n1[a - 1] <= '3' && n2[b - 1] >= '1' && n2[b - 1] <= '3') {
    a--;
    b--;
  }
  return a == b && memcmp(n1, n2, (size_t)a) == 0;
}

static uint8_t rc_table[256];
static int rc_ready = 0;
static void rc_init(void) {
  for (int i = 0; i < 256; i++) rc_table[i] = (uint8_t)i;
  const char *from = "ACGTUMRWSYKVHDBNacgtumrwsykvhdbn", *to = "TGCAAKYWSRMBDHVNtgcaakywsrmbdhvn";
  for (int i = 0; from[i]; i++) rc_table[(uint8_t)from[i]] = (uint8_t)to[i];
  rc_ready = 1;
}

static uint8_t *emit(uint8_t *o, const uint8_t *id, int32_t idl, const uint8_t *tag1, int32_t t1, const uint8_t *tag2,
                     int32_t t2, int has_umi, const uint8_t *seq, const uint8_t *qual, int32_t s, int32_t e, int rc) {
  *o++ = '@';
  memcpy(o, id, (size_t)idl);
  o += idl;
  if (has_umi) {
    *o++ = '_';
    memcpy(o, tag1, (size_t)t1);
    o += t1;
    memcpy(o, tag2, (size_t)t2);
    o += t2;
  }
  *o++ = '\n';
  int32_t L = e - s;
  if (!rc) {
    memcpy(o, seq + s, (size_t)L);
    o += L;
    *o++ = '\n';
    *o++ = '+';
    *o++ = '\n';
    memcpy(o, qual + s, (size_t)L);
    o += L;
  } else {
    for (int32_t i = 0; i < L; i++) *o++ = rc_table[seq[e - 1 - i]];
    *o++ = '\n';
    *o++ = '+';
    *o++ = '\n';
    for (int32_t i = 0; i < L; i++) *o++ = qual[e - 1 - i];
  }
  *o++ = '\n';
  return o;
}

/* Format one chunk.  out[route][mate] are caller-allocated buffers (capacity: raw chunk bytes +
 * 260 per record is always enough); out_len[route][mate] receives the bytes written and
 * counts[route] the records (pairs).  Routes: 0 trimmed, 1 short, 2 untrimmed.
 * Returns 0, or -(record index + 1) when the mates' ids differ. */
int64_t csh_format_chunk(const csh_format_params *fp, int64_t n, uint32_t stride, const uint8_t *raw1,
                         const int64_t *name_off1, const int32_t *name_len1, const uint8_t *seq1,
                         const uint8_t *qual1, const csh_result *res1, const csh_cap2 *cap2, const uint8_t *raw2,
                         const int64_t *name_off2, const int32_t *name_len2, const uint8_t *seq2,
                         const uint8_t *qual2, const csh_result *res2, uint8_t *out[3][2], int64_t out_len[3][2],
                         int64_t counts[3]) {
  if (!rc_ready) rc_init();
  uint8_t *w[3][2];
  for (int r = 0; r < 3; r++) {
    counts[r] = 0;
    for (int m = 0; m < 2; m++) w[r][m] = out[r][m];
  }
  for (int64_t i = 0; i < n; i++) {
    const uint8_t *nm1 = raw1 + name_off1[i];
    int32_t nl1 = strip_suffixes(nm1, name_len1[i], fp->suffix1);
    int32_t id1o, id1l;
    read_id(nm1, nl1, &id1o, &id1l);
    const uint8_t *s1 = seq1 + (size_t)i * stride, *q1 = qual1 + (size_t)i * stride;
    unsigned flags = res1[i].flags;
    if (!fp->paired) {
      int route = (flags & fp->flag_too_short) ? 1 : ((fp->untrimmed_filter && (flags & fp->flag_untrimmed)) ? 2 : 0);
      const uint8_t *t2 = cap2 ? s1 + cap2[i].off : NULL;
      w[route][0] = emit(w[route][0], nm1 + id1o, id1l, s1 + res1[i].cap_off, res1[i].cap_len, t2,
                         cap2 ? cap2[i].len : 0, fp->has_umi, s1, q1, res1[i].start, res1[i].stop,
                         fp->reverse_complement);
      counts[route]++;
      continue;
    }
    const uint8_t *nm2 = raw2 + name_off2[i];
    int32_t nl2 = strip_suffixes(nm2, name_len2[i], fp->suffix2);
    if (!ids_match(nm1, nl1, nm2, nl2)) return -(i + 1);
    int32_t id2o, id2l;
    read_id(nm2, nl2, &id2o, &id2l);
    const uint8_t *s2 = seq2 + (size_t)i * stride, *q2 = qual2 + (size_t)i * stride;
    flags |= res2[i].flags;
    int route = (flags & fp->flag_too_short) ? 1 : ((fp->untrimmed_filter && (flags & fp->flag_untrimmed)) ? 2 : 0);
    w[route][0] = emit(w[route][0], nm1 + id1o, id1l, s1 + res1[i].cap_off, res1[i].cap_len, s2 + res2[i].cap_off,
                       res2[i].cap_len, fp->has_umi, s1, q1, res1[i].start, res1[i].stop, 0);
    w[route][1] = emit(w[route][1], nm2 + id2o, id2l, s1 + res1[i].cap_off, res1[i].cap_len, s2 + res2[i].cap_off,
                       res2[i].cap_len, fp->has_umi, s2, q2, res2[i].start, res2[i].stop, 0);
    counts[route]++;
  }
  for (int r = 0; r < 3; r++)
    for (int m = 0; m < 2; m++) out_len[r][m] = w[r][m] - out[r][m];
  return 0;
}


/* ---- demultiplexed output (extension, BASELINE.json config 5): the trimmed route split by barcode -------------
 * Like csh_format_chunk, but every record of route 0 goes to the bin bin[i] (< n_bins) names: the bins of one
 * mate lie back to back in out_binned[mate] (capacity as for a route buffer), bin b at
 * [bin_off[mate * (n_bins + 1) + b], bin_off[.. + b + 1]).  Two passes: sizes, then bytes.  Records whose
 * bin is out of range (no barcode) are written to route 2 whatever their flags say.
 * bin_counts[b] receives the records (pairs) of bin b, counts[1], counts[2] those of the other routes. */
typedef struct {
  int route;
  int32_t id1o, id1l, id2o, id2l, nl1, nl2;
} csh_prep;

static int prep_record(const csh_format_params *fp, int64_t i, const uint8_t *raw1, const int64_t *name_off1,
                       const int32_t *name_len1, const csh_result *res1, const uint8_t *raw2, const int64_t *name_off2,
                       const int32_t *name_len2, const csh_result *res2, csh_prep *p) {
  const uint8_t *nm1 = raw1 + name_off1[i];
  p->nl1 = strip_suffixes(nm1, name_len1[i], fp->suffix1);
  read_id(nm1, p->nl1, &p->id1o, &p->id1l);
  unsigned flags = res1[i].flags;
  if (fp->paired) {
    const uint8_t *nm2 = raw2 + name_off2[i];
    p->nl2 = strip_suffixes(nm2, name_len2[i], fp->suffix2);
    if (!ids_match(nm1, p->nl1, nm2, p->nl2)) return -1;
    read_id(nm2, p->nl2, &p->id2o, &p->id2l);
    flags |= res2[i].flags;
  }
  p->route = (flags & fp->flag_too_short) ? 1 : ((fp->untrimmed_filter && (flags & fp->flag_untrimmed)) ? 2 : 0);
  return 0;
}

int64_t csh_format_chunk_bins(const csh_format_params *fp, int64_t n, uint32_t stride, const uint8_t *raw1,
                              const int64_t *name_off1, const int32_t *name_len1, const uint8_t *seq1,
                              const uint8_t *qual1, const csh_result *res1, const csh_cap2 *cap2, const uint8_t *raw2,
                              const int64_t *name_off2, const int32_t *name_len2, const uint8_t *seq2,
                              const uint8_t *qual2, const csh_result *res2, const uint8_t *bin, int32_t n_bins,
                              uint8_t *out_binned[2], int64_t *bin_off, int64_t *bin_counts, uint8_t *out[3][2],
                              int64_t out_len[3][2], int64_t counts[3]) {
  if (!rc_ready) rc_init();
  if (!bin || n_bins < 1 || n_bins > 255) return -(n + 1);
  const int mates = fp->paired ? 2 : 1;
  const int64_t stride_off = n_bins + 1;
  for (int m = 0; m < 2; m++)
    for (int b = 0; b <= n_bins; b++) bin_off[m * stride_off + b] = 0;
  for (int b = 0; b < n_bins; b++) bin_counts[b] = 0;
  /* pass 1: bytes per bin and mate (accumulated at bin_off[.. + b + 1]) */
  for (int64_t i = 0; i < n; i++) {
    csh_prep p;
    if (prep_record(fp, i, raw1, name_off1, name_len1, res1, raw2, name_off2, name_len2, res2, &p)) return -(i + 1);
    if (p.route != 0 || bin[i] >= n_bins) continue;
    int32_t tag = 0;
    if (fp->has_umi) tag = 1 + res1[i].cap_len + (fp->paired ? res2[i].cap_len : (cap2 ? cap2[i].len : 0));
    bin_off[bin[i] + 1] += p.id1l + 2 * (int64_t)(res1[i].stop - res1[i].start) + 6 + tag;
    if (fp->paired) bin_off[stride_off + bin[i] + 1] += p.id2l + 2 * (int64_t)(res2[i].stop - res2[i].start) + 6 + tag;
    bin_counts[bin[i]]++;
  }
  for (int m = 0; m < mates; m++)
    for (int b = 0; b < n_bins; b++) bin_off[m * stride_off + b + 1] += bin_off[m * stride_off + b];
  /* pass 2: bytes */
  uint8_t *w[3][2];
  for (int r = 0; r < 3; r++) {
    counts[r] = 0;
    for (int m = 0; m < 2; m++) w[r][m] = out[r][m];
  }
  int64_t *cur = (int64_t *)malloc((size_t)(2 * stride_off) * sizeof(int64_t));
  if (!cur) return -(n + 1);
  for (int64_t j = 0; j < 2 * stride_off; j++) cur[j] = bin_off[j];
  for (int64_t i = 0; i < n; i++) {
    csh_prep p;
    (void)prep_record(fp, i, raw1, name_off1, name_len1, res1, raw2, name_off2, name_len2, res2, &p);
    int route = p.route;
    if (route == 0 && bin[i] >= n_bins) route = 2;
    const uint8_t *nm1 = raw1 + name_off1[i];
    const uint8_t *s1 = seq1 + (size_t)i * stride, *q1 = qual1 + (size_t)i * stride;
    const uint8_t *t2;
    int32_t t2l;
    const uint8_t *s2 = NULL, *q2 = NULL;
    if (fp->paired) {
      s2 = seq2 + (size_t)i * stride;
      q2 = qual2 + (size_t)i * stride;
      t2 = s2 + res2[i].cap_off;
      t2l = res2[i].cap_len;
    } else {
      t2 = cap2 ? s1 + cap2[i].off : NULL;
      t2l = cap2 ? cap2[i].len : 0;
    }
    uint8_t *o1 = route == 0 ? out_binned[0] + cur[bin[i]] : w[route][0];
    uint8_t *e1 = emit(o1, nm1 + p.id1o, p.id1l, s1 + res1[i].cap_off, res1[i].cap_len, t2, t2l, fp->has_umi, s1, q1,
                       res1[i].start, res1[i].stop, fp->paired ? 0 : fp->reverse_complement);
    if (route == 0)
      cur[bin[i]] += e1 - o1;
    else
      w[route][0] = e1;
    if (fp->paired) {
      const uint8_t *nm2 = raw2 + name_off2[i];
      uint8_t *o2 = route == 0 ? out_binned[1] + cur[stride_off + bin[i]] : w[route][1];
      uint8_t *e2 = emit(o2, nm2 + p.id2o, p.id2l, s1 + res1[i].cap_off, res1[i].cap_len, t2, t2l, fp->has_umi, s2, q2,
                         res2[i].start, res2[i].stop, 0);
      if (route == 0)
        cur[stride_off + bin[i]] += e2 - o2;
      else
        w[route][1] = e2;
    }
    if (route != 0) counts[route]++;
  }
  free(cur);
  counts[0] = 0;
  for (int b = 0; b < n_bins; b++) counts[0] += bin_counts[b];
  for (int r = 0; r < 3; r++)
    for (int m = 0; m < 2; m++) out_len[r][m] = w[r][m] - out[r][m];
  return 0;
}

/* ===========================================================================================
 * Text path (cutseq_amd/textpath.py): the host only has to cut the input into blocks of whole
 * records; the device finds the records inside a block.  Two vectorised passes over the text.
 * =========================================================================================== */

/* number of '\n' in buf[0, n) */
#include <immintrin.h>
__attribute__((target("avx2"))) static int64_t count_newlines_avx2(const uint8_t *buf, int64_t n) {
  const __m256i nl = _mm256_set1_epi8('\n');
  int64_t total = 0, i = 0;
  while (i + 32 <= n) {
    /* byte counters: at most 255 rounds before they are folded into 64-bit sums */
    __m256i acc = _mm256_setzero_si256();
    int64_t rounds = (n - i) / 32;
    if (rounds > 255) rounds = 255;
    for (int64_t r = 0; r < rounds; r++, i += 32)
      acc = _mm256_sub_epi8(acc, _mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(buf + i)), nl));
    const __m256i sad = _mm256_sad_epu8(acc, _mm256_setzero_si256());
    total += _mm256_extract_epi64(sad, 0) + _mm256_extract_epi64(sad, 1) + _mm256_extract_epi64(sad, 2) +
             _mm256_extract_epi64(sad, 3);
  }
  for (; i < n; i++) total += buf[i] == '\n';
  return total;
}
int64_t csh_count_newlines(const uint8_t *buf, int64_t n) {
  static int have_avx2 = -1;
  if (have_avx2 < 0) have_avx2 = __builtin_cpu_supports("avx2") ? 1 : 0;
  if (have_avx2) return count_newlines_avx2(buf, n);
  int64_t total = 0;
  const uint8_t *p = buf, *end = buf + n;
  while (p < end && (p = (const uint8_t *)memchr(p, '\n', (size_t)(end - p))) != NULL) {
    total++;
    p++;
  }
  return total;
}

/* offset just behind the k-th '\n' (k >= 1) of buf[0, n), or -1 when there are fewer */
int64_t csh_after_kth_newline(const uint8_t *buf, int64_t n, int64_t k) {
  if (k < 1) return 0;
  int64_t i = 0;
  while (i < n) { /* skip whole blocks, then walk the one that holds the k-th newline */
    int64_t end = i + 65536 < n ? i + 65536 : n;
    int64_t c = csh_count_newlines(buf + i, end - i);
    if (c < k) {
      k -= c;
      i = end;
      continue;
    }
    for (; i < end; i++)
      if (buf[i] == '\n' && --k == 0) return i + 1;
  }
  return -1;
}

/* ===========================================================================================
 * gzip members around device-compressed deflate data (cs_text_* with compression): the device returns
 * one CRC-32 per 32 KB chunk; the member's CRC-32 is their combination (zlib's crc32_combine,
 * restated: multiplication by x^(8 len) modulo the reflected polynomial).
 * =========================================================================================== */
#define CSH_CRC_POLY 0xedb88320u

static uint32_t crc_multmodp(uint32_t a, uint32_t b) {
  uint32_t m = 1u << 31, p = 0;
  for (;;) {
    if (a & m) {
      p ^= b;
      if ((a & (m - 1)) == 0) break;
    }
    m >>= 1;
    b = (b & 1) ? (b >> 1) ^ CSH_CRC_POLY : b >> 1;
  }
  return p;
}

/* x^(n * 2^k) modulo p */
static uint32_t crc_x2nmodp(uint64_t n, unsigned k) {
  static uint32_t table[32];
  static int ready = 0;
  if (!ready) {
    uint32_t p = 1u << 30; /* x^1 */
    table[0] = p;
    for (int i = 1; i < 32; i++) table[i] = p = crc_multmodp(p, p);
    ready = 1;
  }
  uint32_t p = 1u << 31; /* x^0 */
  while (n) {
    if (n & 1) p = crc_multmodp(table[k & 31], p);
    n >>= 1;
    k++;
  }
  return p;
}

/* CRC-32 of the concatenation of n pieces with CRCs crc[i] and lengths len[i] (bytes) */
uint32_t csh_crc32_combine_many(const uint32_t *crc, const uint32_t *len, int64_t n) {
  uint32_t total = 0;
  for (int64_t i = 0; i < n; i++) total = crc_multmodp(crc_x2nmodp(len[i], 3), total) ^ crc[i];
  return total;
}

/* the 256 constants x^(8 * bytes) for bytes = piece * step, piece = 0 .. 255: what a device thread multiplies the
 * CRC of its slice by to move it `bytes` towards the front (cutseq_hip.hip uploads the table once per engine) */
void csh_crc32_shift_table(uint32_t *out, uint32_t step) {
  for (uint32_t i = 0; i < 256; i++) out[i] = crc_x2nmodp((uint64_t)i * step, 3);
}

/* ---------------------------------------------------------------------------------------------------------------
 * FASTA input (the reference reads whatever dnaio detects and asks input_file_format().has_qualities(),
 * cutseq/run.py:437-441, 754-758).  The device parses four-line records, so FASTA text is re-shaped on its way in:
 *     >name          ->  @name
 *     ACGT                ACGTACGT        (sequence lines joined)
 *     ACGT                +
 *                         ~~~~~~~~        (a quality no cutoff trims: the QualityTrimmer step is a no-op without
 *                                          qualities; the output side writes FASTA records, so it never shows)
 * Reader rules as dnaio's FastaReader: lines are stripped of white space at both ends, empty lines and lines that start
 * with '#' are skipped, a record starts at a line that starts with '>', anything else before the first record is an
 * error.  Only WHOLE records are converted: `*consumed` stops at the start of the last record unless `final` is set
 * (the caller carries the rest into the next call).  Returns the bytes written, -1 when `cap` is too small (3 * n + 16
 * always suffices), -2 on a format error (*err_line = 1-based line of the call's text).
 */
static int fa_space(uint8_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

int64_t csh_fasta_to_fastq(const uint8_t *src, int64_t n, uint8_t *dst, int64_t cap, int final, int in_record,
                           int64_t *consumed, int64_t *records, int64_t *err_line) {
  int64_t o = 0, pos = 0, line = 0, recs = 0;
  int64_t rec_src = -1, rec_dst = 0; /* the open record: where it starts in src / dst */
  int64_t seq_len = 0;
  int open = 0;
  (void)in_record;
  *consumed = 0;
  *records = 0;
  *err_line = 0;
  while (pos < n) {
    /* one line: [pos, eol) */
    const uint8_t *nl = memchr(src + pos, '\n', (size_t)(n - pos));
    int64_t eol = nl ? (int64_t)(nl - src) : n;
    if (!nl && !final) break; /* an incomplete line: the caller brings the rest */
    int64_t a = pos, b = eol;
    while (a < b && fa_space(src[a])) ++a;
    while (b > a && fa_space(src[b - 1])) --b;
    ++line;
    const int64_t line_start = pos;
    pos = nl ? eol + 1 : n;
    if (a == b) continue;
    if (src[a] == '>') {
      if (open) { /* the record in front is complete: its '+' and quality lines */
        if (o + 4 + seq_len + 1 > cap) return -1;
        dst[o++] = '\n';
        dst[o++] = '+';
        dst[o++] = '\n';
        memset(dst + o, '~', (size_t)seq_len);
        o += seq_len;
        dst[o++] = '\n';
        ++recs;
        *consumed = line_start;
        *records = recs;
      }
      open = 1;
      rec_src = line_start;
      rec_dst = o;
      seq_len = 0;
      if (o + (b - a) + 1 > cap) return -1;
      dst[o++] = '@';
      memcpy(dst + o, src + a + 1, (size_t)(b - a - 1));
      o += b - a - 1;
      dst[o++] = '\n';
    } else if (src[a] == '#') {
      continue;
    } else if (open) {
      if (o + (b - a) > cap) return -1;
      memcpy(dst + o, src + a, (size_t)(b - a));
      o += b - a;
      seq_len += b - a;
    } else {
      *err_line = line;
      return -2;
    }
  }
  if (open) {
    if (final) {
      if (o + 4 + seq_len + 1 > cap) return -1;
      dst[o++] = '\n';
      dst[o++] = '+';
      dst[o++] = '\n';
      memset(dst + o, '~', (size_t)seq_len);
      o += seq_len;
      dst[o++] = '\n';
      ++recs;
      *consumed = n;
      *records = recs;
    } else {
      o = rec_dst; /* the open record waits for its end */
      *consumed = rec_src;
    }
  } else {
    *consumed = final ? n : pos; /* blank lines / comments only */
  }
  *records = recs;
  return o;
}
