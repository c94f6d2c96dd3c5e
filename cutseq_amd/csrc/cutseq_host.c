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

typedef struct csh_synth_params {
  uint32_t read_len;
  uint32_t stride;
  uint64_t seed;
  uint64_t first_index;
  const char *p5_fw, *p7_fw, *p5_rc, *p7_rc; /* NUL-terminated */
  const char *inline5, *inline3;
  int32_t umi5, umi3, mask5, mask3;
  int32_t strand; /* +1, -1, 0 */
  int32_t single_end;
  double adapter_fraction; /* inserts shorter than the read: 3' adapter visible             */
  double partial_fraction; /* inserts that leave only a 3..19 nt adapter prefix in the read */
  double poly_fraction;    /* poly-A/T stretch of 10..40 nt at the insert end                */
  double art5_fraction;    /* 5' adapter artefact in front of the read                       */
  double sub_rate;         /* per-base substitution                                          */
  double indel_frac;       /* reads with an adapter that get one indel inside it             */
  double n_rate;           /* per-base N                                                     */
} csh_synth_params;

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
  for (uint32_t i = (uint32_t)L; i < p->stride; i++) seq[i] = qual[i] = 0;
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
