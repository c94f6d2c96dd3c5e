/* pinflate_loop.h -- the decoder's block loop, included by pinflate.c once per output form:
 *   PI_NAME coded_block_fast,  PI_T uint16_t, PI_MARKERS 1   chunks of a stream entered in the middle (symbols, markers)
 *   PI_NAME coded_block_bytes, PI_T uint8_t,  PI_MARKERS 0   a member from its start (plain bytes; a reference in front
 *                                                            of the output is corrupt data)
 */
static int PI_NAME(bits_t *b, const huff_t *lit, const huff_t *dist, PI_T *out, size_t cap, size_t *o) {
  size_t at = *o;
  const uint32_t *lt = lit->wide, *dt = dist->wide;
  const uint32_t mask = (1u << WIDE_BITS) - 1u;
  /* the bit reader's state in locals for the length of the block */
  const uint8_t *in = b->in;
  const size_t n = b->n;
  size_t pos = b->pos;
  uint64_t buf = b->buf;
  int cnt = b->cnt;
  int rc = PI_OK;
#define PI_REFILL()                              \
  do {                                           \
    if (pos + 8 <= n) {                          \
      uint64_t w_;                               \
      memcpy(&w_, in + pos, 8);                  \
      buf |= w_ << cnt;                          \
      pos += (size_t)((63 - cnt) >> 3);          \
      cnt |= 56;                                 \
    } else {                                     \
      while (cnt <= 56) {                        \
        buf |= (uint64_t)(pos < n ? in[pos] : 0) << cnt; \
        pos++;                                   \
        cnt += 8;                                \
      }                                          \
    }                                            \
  } while (0)
#define PI_SYNC() (b->pos = pos, b->buf = buf, b->cnt = cnt)
/* consume the entry's bits; `saved_` keeps the buffer for the extra bits */
#define PI_TAKE(e_) (saved = buf, buf >>= ((e_) & 63u), cnt -= (int)((e_) & 63u))
/* the extra bits of the entry just taken: the low (total) bits of the saved buffer above the code */
#define PI_EXTRA(e_) ((uint32_t)((saved & ((1ull << ((e_) & 63u)) - 1ull)) >> (((e_) >> 8) & 15u)))
  for (;;) {
    if (at + 320 > cap) {  /* four literals + the longest match + the overshoot of its last piece, or two short matches */
      rc = PI_ERR_SPACE;
      break;
    }
    PI_REFILL();
    if (pos > n + 16) {
      rc = PI_ERR_INPUT;
      break;
    }
    uint64_t saved;
    uint32_t e = lt[buf & mask];
    if ((e & 0x3000u) == 0 && e != 0) {  /* a whole match in one entry (huff_combine) */
    whole_match:
      /* TWO of them from one refill (2 x 24 bits of >= 56): the refill -- count, position, load, shift, or -- is a
       * dependent chain of its own, twice as long as look-up -> shift -> look-up */
      for (int twice = 0;; ++twice) {
        PI_TAKE(e);
        const int length = 3 + (int)((e >> 14) & 7u);
        const uint32_t tot = e & 63u, xb = (e >> 8) & 15u;
        const int d = (int)(e >> 17) + (int)((uint32_t)((saved & ((1ull << tot) - 1ull)) >> (tot - xb)));
        PI_T *dst = out + at;
        if ((size_t)d <= at && d >= 8) {
          memcpy(dst, dst - d, 8 * sizeof(PI_T));
          memcpy(dst + 8, dst - d + 8, 2 * sizeof(PI_T));  /* (lengths up to 10) */
        } else {
          if (!PI_MARKERS && (size_t)d > at) { /* nothing precedes a member */
            rc = PI_ERR_DATA;
            goto done;
          }
          for (int j = 0; j < length; ++j) {
            const int64_t from = (int64_t)at + j - d;
            dst[j] = from < 0 ? (PI_T)(0x8000u | (uint32_t)(WINDOW + from)) : out[from];
          }
        }
        at += (size_t)length;
        if (twice || cnt < 24) break;
        e = lt[buf & mask];
        if (!((e & 0x3000u) == 0 && e != 0)) break;  /* (looked up again behind the refill) */
      }
      continue;
    }
    if ((e & 0x3000u) == 0x1000u) {  /* literals: up to four from one refill (4 x 11 bits of >= 56) */
      PI_TAKE(e);
      out[at++] = (PI_T)(e >> 16);
      e = lt[buf & mask];
      if ((e & 0x3000u) == 0x1000u) {
        PI_TAKE(e);
        out[at++] = (PI_T)(e >> 16);
        e = lt[buf & mask];
        if ((e & 0x3000u) == 0x1000u) {
          PI_TAKE(e);
          out[at++] = (PI_T)(e >> 16);
          e = lt[buf & mask];
          if ((e & 0x3000u) == 0x1000u) {
            PI_TAKE(e);
            out[at++] = (PI_T)(e >> 16);
            continue;
          }
        }
      }
      /* not a literal; the look-up is still good behind a refill (that only adds bits on top) */
      PI_REFILL();
      if ((e & 0x3000u) == 0 && e != 0) goto whole_match;
    }
    int length;
    if (e == 0) { /* a code longer than WIDE_BITS (or none): the canonical walk */
      PI_SYNC();
      int sym = huff_decode(b, lit);
      pos = b->pos, buf = b->buf, cnt = b->cnt;
      if (sym < 0) {
        rc = PI_ERR_DATA;
        break;
      }
      if (sym < 256) {
        out[at++] = (PI_T)sym;
        continue;
      }
      if (sym == 256) break;
      sym -= 257;
      if (sym >= 29) {
        rc = PI_ERR_DATA;
        break;
      }
      PI_REFILL();
      length = LEN_BASE[sym] + (int)(buf & ((1u << LEN_EXTRA[sym]) - 1u));
      buf >>= LEN_EXTRA[sym];
      cnt -= LEN_EXTRA[sym];
    } else {
      PI_TAKE(e);
      if ((e & 0x3000u) == 0x3000u) break; /* end of block */
      length = (int)(e >> 16) + (int)PI_EXTRA(e);
      if ((e >> 16) == 0) { /* symbols 286 / 287 */
        rc = PI_ERR_DATA;
        break;
      }
    }
    /* >= 56 - 15 - 5 = 36 bits left: 15 + 13 for the distance */
    int d;
    {
      const uint32_t de = dt[buf & mask];
      if (de) {
        PI_TAKE(de);
        d = (int)(de >> 16) + (int)PI_EXTRA(de);
        if ((de >> 16) == 0) { /* codes 30 / 31 */
          rc = PI_ERR_DATA;
          break;
        }
      } else {
        PI_SYNC();
        const int ds = huff_decode(b, dist);
        pos = b->pos, buf = b->buf, cnt = b->cnt;
        if (ds < 0 || ds >= 30) {
          rc = PI_ERR_DATA;
          break;
        }
        d = DIST_BASE[ds] + (int)(buf & ((1u << DIST_EXTRA[ds]) - 1u));
        buf >>= DIST_EXTRA[ds];
        cnt -= DIST_EXTRA[ds];
      }
    }
    PI_T *dst = out + at;
    if ((size_t)d <= at) {
      const PI_T *src = dst - d;
      if (d >= 8) {  /* pieces of eight symbols, one behind the other: a piece may read what the one before wrote */
        memcpy(dst, src, 8 * sizeof(PI_T));
        memcpy(dst + 8, src + 8, 8 * sizeof(PI_T));
        if (length > 16)
          for (int j = 16; j < length; j += 8) memcpy(dst + j, src + j, 8 * sizeof(PI_T));
      } else {
        for (int j = 0; j < length; ++j) dst[j] = src[j];
      }
    } else {
      if (!PI_MARKERS) { /* nothing precedes a member */
        rc = PI_ERR_DATA;
        break;
      }
      /* reaches in front of the chunk: markers for that part, then the chunk's own symbols */
      for (int j = 0; j < length; ++j) {
        const int64_t from = (int64_t)at + j - d;
        dst[j] = from < 0 ? (PI_T)(0x8000u | (uint32_t)(WINDOW + from)) : out[from];
      }
    }
    at += (size_t)length;
  }
done:
  PI_SYNC();
#undef PI_REFILL
#undef PI_SYNC
#undef PI_TAKE
#undef PI_EXTRA
  *o = at;
  if (rc != PI_OK) return rc;
  return bits_pos(b) > (uint64_t)b->n * 8u ? PI_ERR_INPUT : PI_OK;
}

