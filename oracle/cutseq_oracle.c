/*
 * cutseq_oracle.c -- CPU restatement of the trimming hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (cutseq_amd + libcutseq_hip.so) never does.
 *
 * PARITY UNPINNED.  The per-read arithmetic of y9c/cutseq lives in the third-party
 * dependency cutadapt (pyproject.toml:17 pins `cutadapt~=5.0`; no lock file), which is
 * not in /root/reference, not installed and not fetchable.  The reference ships no tests
 * and no expected outputs.  What follows restates cutadapt 5.x's
 * published algorithm (src/cutadapt/_align.pyx Aligner.locate, adapters.py *.match_to,
 * qualtrim.pyx quality_trim_index, modifiers.py) from knowledge of its source, anchored
 * on the reference's own call sites:
 *   chain composition          cutseq/run.py:326-426 (single), 533-731 (paired)
 *   ConditionalCutter          cutseq/run.py:145-161
 *   IsUntrimmedAny             cutseq/run.py:97-110
 *   TooShort / filters         cutseq/run.py:446-451, 763-769
 * Every cutadapt rule that is a recollection sits in ONE function below, named after
 * the cutadapt function it restates, so it can be corrected in one edit.
 *
 * Plain C, scalar, no dependencies besides libc/pthread.  It deliberately keeps
 * cutadapt's own formulation (Ukkonen cut-off `last`, three-int cells) rather than the
 * GPU kernel's (saturated packed cells, bit-parallel filter, windowed DP), so a parity
 * run compares two different programs.
 */
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/cutseq_hip.h"

typedef struct {
  int cost, score, origin;
} entry_t;

typedef struct {
  int origin, cost, score, ref_stop, query_stop;
} match_t;

#define MATCH_SCORE 1
#define MISMATCH_SCORE (-1)
#define INSERTION_SCORE (-2)
#define DELETION_SCORE (-2)

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* cutadapt _align.pyx: the "update best" predicate inside Aligner.locate.
 * CS_SELECT_LEFTMOST: cutadapt >= 4.0 (changelog 4.0: "more accurately pick the leftmost
 *   adapter occurrence"): first acceptable hit; later hits replace it only if they overlap
 *   it (start within m/2 of it) or are longer, AND score higher.
 * CS_SELECT_SCORE: cutadapt 3.x, the rule SURVEY.md appendix B.2 wrote down. */
static int better_candidate(int select_rule, int m, int n, const match_t *best, int best_length, int cost,
                            int score, int origin, int length) {
  if (best->cost == m + n + 1) return 1; /* nothing recorded yet */
  if (select_rule == CS_SELECT_SCORE)
    return score > best->score || (score == best->score && cost < best->cost);
  if (origin <= best->origin + m / 2 && score > best->score) return 1;
  if (length > best_length && score > best->score) return 1;
  return 0;
}

/*
 * cutadapt _align.pyx Aligner.locate(query) for an ACGT-only reference (compare_ascii
 * branch: plain byte equality, no wildcards, indel cost 1).
 * thr[L] = floor(L * max_error_rate) stands for `cost <= L * max_error_rate`.
 * out = (ref_start, ref_stop, query_start, query_stop, score, errors); returns 1 on a hit.
 */
static uint8_t ascii_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

/* `query` is read in place: character j - 1 of the aligner's query is query[(j - 1) * qstep], upper-cased when
 * `fold` is set -- RightmostFrontAdapter walks the read backwards (qstep = -1 from its last base) and
 * match_to aligns sequence.upper(), neither needs a copy of the read. */
static int locate_strided(const uint8_t *ref, int m, const uint8_t *query, int qstep, int fold, int n,
                          const uint8_t *thr, int k, int flags, int min_overlap, int select_rule, int indel_tie,
                          int out[6]) {
  const int start_in_ref = flags & CS_REF_START, start_in_query = flags & CS_QUERY_START;
  const int stop_in_ref = flags & CS_REF_END, stop_in_query = flags & CS_QUERY_STOP;
  entry_t column[CS_MAX_ADAPTER + 1];
  int i, j;
  int max_n = n, min_n = 0;
  if (!start_in_query) max_n = imin(n, m + k);
  if (!stop_in_query) min_n = imax(0, n - m - k);

  for (i = 0; i <= m; i++) {
    column[i].score = 0;
    if (!start_in_ref && !start_in_query) {
      column[i].cost = imax(i, min_n);
      column[i].origin = 0;
    } else if (start_in_ref && !start_in_query) {
      column[i].cost = min_n;
      column[i].origin = imin(0, min_n - i);
    } else if (!start_in_ref && start_in_query) {
      column[i].cost = i;
      column[i].origin = imax(0, min_n - i);
    } else {
      column[i].cost = imin(i, min_n);
      column[i].origin = min_n - i;
    }
  }

  match_t best;
  best.ref_stop = m;
  best.query_stop = n;
  best.cost = m + n + 1;
  best.origin = 0;
  best.score = 0;

  /* Ukkonen's trick: index of the last cell that is at most k */
  int last = imin(m, k + 1);
  if (start_in_ref) last = m;

  for (j = min_n + 1; j <= max_n; j++) {
    entry_t diag = column[0];
    if (start_in_query)
      column[0].origin = j;
    else
      column[0].cost = j;
    const uint8_t qraw = query[(ptrdiff_t)(j - 1) * qstep];
    const uint8_t qc = fold ? ascii_upper(qraw) : qraw;
    for (i = 1; i <= last; i++) {
      int cost, origin, score;
      if (ref[i - 1] == qc) {
        cost = diag.cost;
        origin = diag.origin;
        score = diag.score + MATCH_SCORE;
      } else {
        int cost_diag = diag.cost + 1;
        int cost_deletion = column[i].cost + 1;
        int cost_insertion = column[i - 1].cost + 1;
        if (cost_diag <= cost_deletion && cost_diag <= cost_insertion) {
          cost = cost_diag;
          origin = diag.origin;
          score = diag.score + MISMATCH_SCORE;
        } else if (indel_tie == CS_TIE_INSERTION ? cost_insertion <= cost_deletion
                                                  : cost_insertion < cost_deletion) {
          /* SURVEY.md appendix B.2 order: insertion wins the tie (CS_TIE_DELETION: it loses) */
          cost = cost_insertion;
          origin = column[i - 1].origin;
          score = column[i - 1].score + INSERTION_SCORE;
        } else {
          cost = cost_deletion;
          origin = column[i].origin;
          score = column[i].score + DELETION_SCORE;
        }
      }
      diag = column[i];
      column[i].cost = cost;
      column[i].origin = origin;
      column[i].score = score;
    }
    while (last >= 0 && column[last].cost > k) last--;
    if (last < m) {
      last++;
    } else if (stop_in_query) {
      /* full reference reached: candidate ending at query position j */
      int cost = column[m].cost, score = column[m].score, origin = column[m].origin;
      int length = m + imin(origin, 0);
      int ok = length >= min_overlap && cost <= thr[length];
      int best_length = m + imin(best.origin, 0);
      if (ok && better_candidate(select_rule, m, n, &best, best_length, cost, score, origin, length)) {
        best.score = score;
        best.cost = cost;
        best.origin = origin;
        best.ref_stop = m;
        best.query_stop = j;
        if (cost == 0 && origin >= 0) break; /* exact full-length hit: stop early */
      }
    }
  }

  if (max_n == n) {
    /* candidates that end at the end of the query: partial reference prefix allowed
     * only with REF_END */
    int first_i = stop_in_ref ? 0 : m;
    for (i = m; i >= first_i; i--) {
      int cost = column[i].cost, score = column[i].score, origin = column[i].origin;
      int length = i + imin(origin, 0);
      int ok = length >= min_overlap && length >= 0 && cost <= thr[imax(length, 0)];
      int best_length = best.ref_stop + imin(best.origin, 0);
      if (ok && better_candidate(select_rule, m, n, &best, best_length, cost, score, origin, length)) {
        best.score = score;
        best.cost = cost;
        best.origin = origin;
        best.ref_stop = i;
        best.query_stop = n;
      }
    }
  }

  if (best.cost == m + n + 1) return 0;
  if (best.origin >= 0) {
    out[0] = 0;
    out[2] = best.origin;
  } else {
    out[0] = -best.origin;
    out[2] = 0;
  }
  out[1] = best.ref_stop;
  out[3] = best.query_stop;
  out[4] = best.score;
  out[5] = best.cost;
  return 1;
}

int cs_oracle_locate2(const uint8_t *ref, int m, const uint8_t *query, int n, const uint8_t *thr, int k,
                      int flags, int min_overlap, int select_rule, int indel_tie, int out[6]) {
  return locate_strided(ref, m, query, 1, 0, n, thr, k, flags, min_overlap, select_rule, indel_tie, out);
}

/* the SURVEY appendix B.2 tie order, kept under the old name for callers that do not care */
int cs_oracle_locate(const uint8_t *ref, int m, const uint8_t *query, int n, const uint8_t *thr, int k,
                     int flags, int min_overlap, int select_rule, int out[6]) {
  return cs_oracle_locate2(ref, m, query, n, thr, k, flags, min_overlap, select_rule, CS_TIE_INSERTION, out);
}

/* str.find: leftmost exact occurrence of pat in text, -1 if none */
static int find_exact(const uint8_t *text, int n, const uint8_t *pat, int m) {
  for (int p = 0; p + m <= n; p++)
    if (memcmp(text + p, pat, (size_t)m) == 0) return p;
  return -1;
}

/*
 * cutadapt adapters.py  <Adapter>.match_to(sequence) for the six classes cutseq uses:
 *   every class           : Aligner.locate(sequence.upper())  (the k-mer finder in front of it is
 *                           result-neutral); RightmostFrontAdapter locates the reversed adapter in
 *                           the reversed read and maps the coordinates back
 *   op->shortcut == FIND  : cutadapt <= 2.x, str.find / str.rfind first (opt-in, not the default)
 *   params->case_rule     : CS_CASE_SENSITIVE skips the upper() (opt-in, not the default)
 * `op->seq` is already reversed for the rightmost variant, so find-on-reversed == rfind.
 * Returns 1 and (rstart, rstop) in forward read coordinates.
 */
int cs_oracle_match2(const cs_op *op, const cs_params *params, const uint8_t *read, int n, int *rstart,
                     int *rstop) {
  const int select_rule = params->select_rule, fold = params->case_rule == CS_CASE_FOLD;
  int m = op->m, qs, qe, hit = 0;
  if (op->shortcut == CS_SHORTCUT_FIND) { /* cutadapt <= 2.x str.find / str.rfind: works on a copy (opt-in path) */
    uint8_t buf[65536];
    for (int i = 0; i < n; i++) {
      uint8_t c = read[op->reversed ? n - 1 - i : i];
      buf[i] = fold ? ascii_upper(c) : c;
    }
    int pos = find_exact(buf, n, op->seq, m);
    if (pos >= 0) {
      qs = pos;
      qe = pos + m;
      hit = 1;
    }
  }
  if (!hit) {
    int out[6];
    const uint8_t *q0 = op->reversed ? read + (n > 0 ? n - 1 : 0) : read;
    if (!locate_strided(op->seq, m, q0, op->reversed ? -1 : 1, fold, n, op->thr, op->k, op->align_flags,
                        op->min_overlap, select_rule, params->indel_tie, out))
      return 0;
    qs = out[2];
    qe = out[3];
  }
  if (op->reversed) {
    *rstart = n - qe;
    *rstop = n - qs;
  } else {
    *rstart = qs;
    *rstop = qe;
  }
  return 1;
}

int cs_oracle_match(const cs_op *op, int select_rule, const uint8_t *read, int n, int *rstart, int *rstop) {
  cs_params p;
  memset(&p, 0, sizeof p);
  p.select_rule = (uint8_t)select_rule;
  return cs_oracle_match2(op, &p, read, n, rstart, rstop);
}

/* cutadapt qualtrim.pyx quality_trim_index(qualities, cutoff_front=0, cutoff_back, base):
 * BWA-style running sum from the 3' end.  With cutoff_front == 0 the 5' loop never moves
 * `start` (first base with q > 0 makes the sum negative), so only `stop` is computed. */
int cs_oracle_quality_trim_index(const uint8_t *qual, int n, int cutoff_back, int base) {
  int s = 0, max_qual = 0, stop = n;
  for (int i = n - 1; i >= 0; i--) {
    s += cutoff_back - ((int)qual[i] - base);
    if (s < 0) break;
    if (s > max_qual) {
      max_qual = s;
      stop = i;
    }
  }
  return stop; /* start(=0) >= stop collapses to the empty read, same interval */
}

/* One read through one mate's op chain: the body of cutadapt's per-read modifier loop as
 * cutseq configures it.  The read is never copied: every modifier only removes a prefix
 * or suffix, so the state is the half-open interval [s, e) of the original record. */
void cs_oracle_trim_read(const cs_op *ops, int n_ops, const cs_params *params, const uint8_t *seq,
                         const uint8_t *qual, int len, cs_result *res, cs_cap2 *cap2, cs_stats *st) {
  int s = 0, e = len, n_matches = 0;
  unsigned flags = 0;
  int cap_off = 0, cap_len = 0, cap2_off = 0, cap2_len = 0;
  for (int t = 0; t < n_ops; t++) {
    const cs_op *op = &ops[t];
    int n = e - s;
    if (op->kind == CS_OP_ADAPTER) {
      int rstart, rstop;
      if (cs_oracle_match2(op, params, seq + s, n, &rstart, &rstop)) {
        n_matches++; /* info.matches.append(match) */
        flags |= op->match_flag;
        if (st) st->op_matched[op->stat_slot]++;
        if (op->remove == CS_REMOVE_BEFORE)
          s += rstop; /* RemoveBeforeMatch.trimmed: read[rstop:] */
        else
          e = s + rstart; /* RemoveAfterMatch.trimmed: read[:rstart] */
      } else if (op->required) {
        flags |= CS_F_UNTRIMMED; /* IsUntrimmedAny: adapter not in info.matches */
      }
    } else if (op->kind == CS_OP_CUT) {
      /* ConditionalCutter.__call__ (cutseq/run.py:154-155) */
      if (op->conditional && n_matches == 0 && n < (int)op->force_min_len) continue;
      int c, off;
      if (op->cut_len > 0) { /* info.cut_prefix = seq[:L]; read[L:] */
        c = imin(op->cut_len, n);
        off = s;
        s += c;
      } else if (op->cut_len < 0) { /* info.cut_suffix = seq[L:]; read[:L] */
        c = imin(-op->cut_len, n);
        off = e - c;
        e -= c;
      } else {
        continue;
      }
      if (op->capture == 1) {
        cap_off = off;
        cap_len = c;
      } else if (op->capture == 2) {
        cap2_off = off;
        cap2_len = c;
      }
    } else if (op->kind == CS_OP_QTRIM) {
      int stop = cs_oracle_quality_trim_index(qual + s, n, op->q_cutoff, op->q_base);
      if (stop < n) flags |= CS_F_QTRIMMED;
      if (st) st->qualtrim_bp += (uint64_t)(n - stop);
      e = s + stop;
    }
  }
  if (e - s < (int)params->min_length) flags |= CS_F_TOO_SHORT;
  res->start = (uint16_t)s;
  res->stop = (uint16_t)e;
  res->cap_off = (uint16_t)cap_off;
  res->cap_len = (uint8_t)cap_len;
  res->flags = (uint8_t)flags;
  if (cap2) {
    cap2->off = (uint16_t)cap2_off;
    cap2->len = (uint8_t)cap2_len;
    cap2->_pad = 0;
  }
  if (st) {
    st->n_reads++;
    st->in_bp += (uint64_t)len;
    st->out_bp += (uint64_t)(e - s);
    if (flags & CS_F_TOO_SHORT) st->n_too_short++;
    if (flags & CS_F_UNTRIMMED) st->n_untrimmed++;
  }
}

/* One mate of a batch, same array layout as cs_trim_device (include/cutseq_hip.h). */
int cs_oracle_trim(const cs_op *ops, int n_ops, const cs_params *params, const uint8_t *seq,
                   const uint8_t *qual, const uint16_t *len, uint32_t n_reads, uint32_t stride,
                   cs_result *out, cs_cap2 *cap2, cs_stats *stats) {
  if (!ops || n_ops < 0 || n_ops > CS_MAX_OPS || !params || (n_reads && (!seq || !qual || !len || !out)))
    return CS_ERR_ARG;
  for (uint32_t r = 0; r < n_reads; r++) {
    if (len[r] > stride) return CS_ERR_ARG;
    cs_oracle_trim_read(ops, n_ops, params, seq + (size_t)r * stride, qual + (size_t)r * stride, len[r],
                        &out[r], cap2 ? &cap2[r] : NULL, stats);
  }
  return CS_OK;
}

/* ---- multi-threaded driver, used as the timed CPU baseline (bench.py cpu_baseline) ---- */
typedef struct {
  const cs_op *ops;
  int n_ops;
  const cs_params *params;
  const uint8_t *seq, *qual;
  const uint16_t *len;
  uint32_t lo, hi, stride;
  cs_result *out;
  cs_cap2 *cap2;
  cs_stats stats;
  int rc;
} job_t;

static void *worker(void *arg) {
  job_t *j = (job_t *)arg;
  j->rc = cs_oracle_trim(j->ops, j->n_ops, j->params, j->seq + (size_t)j->lo * j->stride,
                         j->qual + (size_t)j->lo * j->stride, j->len + j->lo, j->hi - j->lo, j->stride,
                         j->out + j->lo, j->cap2 ? j->cap2 + j->lo : NULL, &j->stats);
  return NULL;
}

int cs_oracle_trim_mt(const cs_op *ops, int n_ops, const cs_params *params, const uint8_t *seq,
                      const uint8_t *qual, const uint16_t *len, uint32_t n_reads, uint32_t stride,
                      cs_result *out, cs_cap2 *cap2, cs_stats *stats, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  job_t *jobs = (job_t *)calloc((size_t)n_threads, sizeof(job_t));
  pthread_t *tids = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  if (!jobs || !tids) return CS_ERR_NOMEM;
  uint32_t chunk = (n_reads + (uint32_t)n_threads - 1) / (uint32_t)n_threads;
  int rc = CS_OK;
  for (int t = 0; t < n_threads; t++) {
    uint32_t lo = (uint32_t)t * chunk, hi = lo + chunk;
    if (lo > n_reads) lo = n_reads;
    if (hi > n_reads) hi = n_reads;
    jobs[t] = (job_t){ops, n_ops, params, seq, qual, len, lo, hi, stride, out, cap2, {0}, 0};
    pthread_create(&tids[t], NULL, worker, &jobs[t]);
  }
  for (int t = 0; t < n_threads; t++) {
    pthread_join(tids[t], NULL);
    if (jobs[t].rc != CS_OK) rc = jobs[t].rc;
    if (stats) {
      const uint64_t *src = (const uint64_t *)&jobs[t].stats;
      uint64_t *dst = (uint64_t *)stats;
      for (size_t w = 0; w < sizeof(cs_stats) / sizeof(uint64_t); w++) dst[w] += src[w];
    }
  }
  free(jobs);
  free(tids);
  return rc;
}

int cs_oracle_abi_version(void) { return CS_ABI_VERSION; }
