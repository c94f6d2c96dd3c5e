/* pinflate.c -- the pieces of a parallel decoder for ONE deflate stream (host code, plain C, part of
 * libcutseq_host.so).
 *
 * The usual sequencer output is a .fastq.gz that is one single gzip member: a reader can only enter it at its start,
 * and zlib inflates it at ~0.3 GB/s of text on one core -- one million read pairs per second whatever the GPU does
 * (the reference reads its input through xopen the same way, cutseq/run.py:434-441, 751-758).  The known way round it
 * (pugz, rapidgzip): cut the COMPRESSED bytes into chunks, find a deflate block boundary near the start of every
 * chunk by trying bit positions, and decode every chunk on its own.  What a chunk cannot know is the 32 KB of text in
 * front of it; back-references into that window are written as MARKERS (16-bit symbols: 0..255 a byte, 0x8000 | i the
 * i-th byte of the unknown window), copies of markers copy markers, and once the chunk in front has been resolved the
 * markers are replaced in one linear pass.  A chunk's start is proven, not guessed: the chunk in front of it, decoded
 * from ITS proven start, must end on exactly that bit (cutseq_amd/codec.py walks the chain and re-decodes serially
 * where it does not).
 *
 *   csh_deflate_find_block   first bit position >= `from_bit` at which a dynamic-Huffman block with a valid header
 *                            starts, decodes to its end-of-block symbol without an invalid code and is followed by
 *                            a plausible block header
 *   csh_inflate_chunk        decode from a block boundary up to the first block boundary at or behind `stop_bit`
 *                            (or the end of the final block) into 16-bit symbols
 *   csh_resolve_markers      symbols + the 32 KB window in front of the chunk -> bytes
 *   csh_inflate_stream       a whole stream from its start (a gzip member), straight into bytes
 *
 * RFC 1951 throughout; the Huffman construction follows its section 3.2.2, the slow decoding loop is the canonical
 * count / first-code walk. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <emmintrin.h>

#define PI_OK 0
#define PI_ERR_DATA -1    /* not a valid deflate stream from this position */
#define PI_ERR_SPACE -2   /* the output buffer is too small: call again with a bigger one */
#define PI_ERR_INPUT -3   /* ran off the end of the compressed bytes */

#define FAST_BITS 10
#define MAXBITS 15
#define WINDOW 32768

typedef struct {
  const uint8_t *in;
  size_t n;      /* bytes of input */
  size_t pos;    /* next byte to load (may run past n: zero bytes, flagged when the position is checked) */
  uint64_t buf;  /* bits not consumed yet, LSB first */
  int cnt;
} bits_t;

static inline void bits_refill(bits_t *b) {
  if (b->pos + 8 <= b->n) {
    uint64_t w;
    memcpy(&w, b->in + b->pos, 8);
    b->buf |= w << b->cnt;
    b->pos += (size_t)((63 - b->cnt) >> 3);
    b->cnt |= 56;
  } else {
    while (b->cnt <= 56) {
      const uint64_t byte = b->pos < b->n ? b->in[b->pos] : 0;
      b->buf |= byte << b->cnt;
      b->pos++;
      b->cnt += 8;
    }
  }
}
static inline void bits_init(bits_t *b, const uint8_t *in, size_t n, uint64_t bitpos) {
  b->in = in;
  b->n = n;
  b->pos = (size_t)(bitpos >> 3);
  b->buf = 0;
  b->cnt = 0;
  bits_refill(b);
  b->buf >>= (bitpos & 7);
  b->cnt -= (int)(bitpos & 7);
}
static inline uint64_t bits_pos(const bits_t *b) { return ((uint64_t)b->pos << 3) - (uint64_t)b->cnt; }
static inline uint32_t bits_take(bits_t *b, int k) { /* k <= 32, the buffer holds enough (refilled by the caller) */
  const uint32_t v = (uint32_t)(b->buf & ((1ull << k) - 1ull));
  b->buf >>= k;
  b->cnt -= k;
  return v;
}

#define WIDE_BITS 11
typedef struct {
  uint16_t fast[1 << FAST_BITS]; /* (symbol << 4) | length for codes of up to FAST_BITS bits, 0 otherwise */
  uint16_t count[MAXBITS + 1];
  uint16_t symbol[288];
  uint8_t len[288];              /* the code lengths the tables were built from */
  int n;
  /* literal / length codes only, built by huff_widen for the real decoder: one look-up tells literal, length (with
   * its base and extra-bit count) and end of block apart.  [7:0] bits to consume in all, code + extra bits (0: a code
   * longer than WIDE_BITS or none) -- the decoder shifts by the entry itself, the dependent chain from one look-up to
   * the next is mask, load, shift --, [11:8] code length (where the extra bits start), [13:12] 1 literal / 2 length /
   * 3 end of block, [31:16] literal or length base (0: invalid).
   * Kind 0 with a non-zero entry is a WHOLE MATCH (huff_combine): a length code without extra bits (lengths 3..10)
   * and the distance code behind it both fit the index -- [5:0] bits to consume in all (both codes and the
   * distance's extra bits), [11:8] distance extra bits, [16:14] length - 3, [31:17] distance base. */
  uint32_t wide[1 << WIDE_BITS];
} huff_t;

/* -> 0 complete code, 1 incomplete, -1 over-subscribed */
static int huff_build(huff_t *h, const uint8_t *len, int n) {
  uint16_t offs[MAXBITS + 1], next[MAXBITS + 1];
  memset(h->count, 0, sizeof h->count);
  for (int s = 0; s < n; ++s) h->count[len[s]]++;
  memset(h->fast, 0, sizeof h->fast);
  memcpy(h->len, len, (size_t)n);
  h->n = n;
  if (h->count[0] == n) return 1; /* no codes at all */
  int left = 1;
  for (int l = 1; l <= MAXBITS; ++l) {
    left <<= 1;
    left -= h->count[l];
    if (left < 0) return -1;
  }
  offs[1] = 0;
  for (int l = 1; l < MAXBITS; ++l) offs[l + 1] = (uint16_t)(offs[l] + h->count[l]);
  for (int s = 0; s < n; ++s)
    if (len[s]) h->symbol[offs[len[s]]++] = (uint16_t)s;
  /* canonical codes, MSB first; the stream delivers them bit-reversed */
  uint32_t code = 0;
  for (int l = 1; l <= MAXBITS; ++l) {
    next[l] = (uint16_t)code;
    code = (code + h->count[l]) << 1;
  }
  for (int s = 0; s < n; ++s) {
    const int l = len[s];
    if (!l) continue;
    const uint32_t c = next[l]++;
    if (l > FAST_BITS) continue;
    uint32_t r = 0;
    for (int i = 0; i < l; ++i) r |= ((c >> i) & 1u) << (l - 1 - i);
    const uint16_t e = (uint16_t)((s << 4) | l);
    for (uint32_t i = r; i < (1u << FAST_BITS); i += 1u << l) h->fast[i] = e;
  }
  return left > 0 ? 1 : 0;
}

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};

static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

/* the wide table of a literal / length code (see huff_t) */
static void huff_widen(huff_t *h) {
  uint16_t next[MAXBITS + 1];
  uint32_t code = 0;
  for (int l = 1; l <= MAXBITS; ++l) {
    next[l] = (uint16_t)code;
    code = (code + h->count[l]) << 1;
  }
  memset(h->wide, 0, sizeof h->wide);
  for (int s = 0; s < h->n; ++s) {
    const int l = h->len[s];
    if (!l) continue;
    const uint32_t c = next[l]++;
    if (l > WIDE_BITS) continue;
    uint32_t r = 0;
    for (int i = 0; i < l; ++i) r |= ((c >> i) & 1u) << (l - 1 - i);
    uint32_t e;
    if (s < 256)
      e = ((uint32_t)s << 16) | 0x1000u | ((uint32_t)l << 8) | (uint32_t)l;
    else if (s == 256)
      e = 0x3000u | ((uint32_t)l << 8) | (uint32_t)l;
    else if (s - 257 < 29)
      e = ((uint32_t)LEN_BASE[s - 257] << 16) | 0x2000u | ((uint32_t)l << 8) | (uint32_t)(l + LEN_EXTRA[s - 257]);
    else
      e = 0x2000u | ((uint32_t)l << 8) | (uint32_t)l; /* 286, 287: codes without a meaning (base 0) */
    for (uint32_t i = r; i < (1u << WIDE_BITS); i += 1u << l) h->wide[i] = e;
  }
}

/* ... and of a distance code: [7:0] bits to consume in all, [11:8] code length, [31:16] distance base (0: codes 30, 31) */
static void huff_widen_dist(huff_t *h) {
  uint16_t next[MAXBITS + 1];
  uint32_t code = 0;
  for (int l = 1; l <= MAXBITS; ++l) {
    next[l] = (uint16_t)code;
    code = (code + h->count[l]) << 1;
  }
  memset(h->wide, 0, sizeof h->wide);
  for (int s = 0; s < h->n; ++s) {
    const int l = h->len[s];
    if (!l) continue;
    const uint32_t c = next[l]++;
    if (l > WIDE_BITS) continue;
    uint32_t r = 0;
    for (int i = 0; i < l; ++i) r |= ((c >> i) & 1u) << (l - 1 - i);
    const uint32_t e = s < 30 ? ((uint32_t)DIST_BASE[s] << 16) | ((uint32_t)l << 8) | (uint32_t)(l + DIST_EXTRA[s]) : ((uint32_t)l << 8) | (uint32_t)l;
    for (uint32_t i = r; i < (1u << WIDE_BITS); i += 1u << l) h->wide[i] = e;
  }
}

/* gzip -1 turns FASTQ into short matches, and a match is two codes: look-up, shift, look-up, shift.  Where a length
 * code without extra bits and the distance code behind it fit the index together, the literal / length table gets
 * an entry for the whole match (see huff_t): one look-up and one shift per match on the decoder's dependent chain. */
static void huff_combine(huff_t *lit, const huff_t *dist) {
  for (uint32_t i = 0; i < (1u << WIDE_BITS); ++i) {
    const uint32_t e = lit->wide[i];
    if ((e & 0x3000u) != 0x2000u) continue;          /* not a length */
    const uint32_t l = (e >> 8) & 15u;
    if ((e & 0xffu) != l || (e >> 16) < 3 || (e >> 16) > 10) continue; /* extra bits (or no meaning) */
    const uint32_t rem = WIDE_BITS - l;
    if (rem == 0) continue;
    const uint32_t de = dist->wide[(i >> l) & ((1u << rem) - 1u)];
    const uint32_t dl = (de >> 8) & 15u;
    if (de == 0 || dl > rem || (de >> 16) == 0) continue; /* the distance code is longer than what the index shows */
    const uint32_t xb = (de & 0xffu) - dl, total = l + (de & 0xffu);
    lit->wide[i] = total | (xb << 8) | (((e >> 16) - 3u) << 14) | ((de >> 16) << 17);
  }
}

/* one symbol; -1 = no such code.  The buffer holds at least MAXBITS bits. */
static inline int huff_decode(bits_t *b, const huff_t *h) {
  const uint16_t e = h->fast[b->buf & ((1u << FAST_BITS) - 1u)];
  if (e) {
    b->buf >>= (e & 15);
    b->cnt -= (e & 15);
    return e >> 4;
  }
  int code = 0, first = 0, index = 0;
  uint64_t v = b->buf;
  for (int l = 1; l <= MAXBITS; ++l) {
    code |= (int)(v & 1u);
    v >>= 1;
    const int count = h->count[l];
    if (code - count < first) {
      b->buf >>= l;
      b->cnt -= l;
      return h->symbol[index + (code - first)];
    }
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

static const uint8_t PRE_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* the header of a dynamic block (behind its three type bits) -> the two codes; PI_OK or PI_ERR_DATA.
 * The checks are zlib's: complete length-code code, no repeat without a predecessor, an end-of-block code, literal /
 * length code complete (or one single code), distance code complete, a single code or absent. */
static int dynamic_header(bits_t *b, huff_t *lit, huff_t *dist) {
  bits_refill(b);
  const int hlit = (int)bits_take(b, 5) + 257, hdist = (int)bits_take(b, 5) + 1, hclen = (int)bits_take(b, 4) + 4;
  if (hlit > 286 || hdist > 30) return PI_ERR_DATA;
  uint8_t pre[19];
  memset(pre, 0, sizeof pre);
  for (int i = 0; i < hclen; ++i) {
    if (b->cnt < 3) bits_refill(b);
    pre[PRE_ORDER[i]] = (uint8_t)bits_take(b, 3);
  }
  {
    /* complete or nothing (cheap, and what throws out nearly every wrong bit position in the block finder) */
    int left = 1 << 7;
    for (int i = 0; i < 19; ++i)
      if (pre[i]) left -= 1 << (7 - pre[i]);
    if (left != 0) return PI_ERR_DATA;
  }
  huff_t ph;
  if (huff_build(&ph, pre, 19) != 0) return PI_ERR_DATA;
  uint8_t len[286 + 30];
  int i = 0;
  while (i < hlit + hdist) {
    bits_refill(b);
    const int sym = huff_decode(b, &ph);
    if (sym < 0) return PI_ERR_DATA;
    if (sym < 16) {
      len[i++] = (uint8_t)sym;
      continue;
    }
    int rep, val = 0;
    if (sym == 16) {
      if (i == 0) return PI_ERR_DATA;
      val = len[i - 1];
      rep = 3 + (int)bits_take(b, 2);
    } else if (sym == 17) {
      rep = 3 + (int)bits_take(b, 3);
    } else {
      rep = 11 + (int)bits_take(b, 7);
    }
    if (i + rep > hlit + hdist) return PI_ERR_DATA;
    while (rep--) len[i++] = (uint8_t)val;
  }
  if (len[256] == 0) return PI_ERR_DATA;
  int rc = huff_build(lit, len, hlit);
  if (rc < 0 || (rc > 0 && hlit - lit->count[0] != 1)) return PI_ERR_DATA;
  rc = huff_build(dist, len + hlit, hdist);
  if (rc < 0 || (rc > 0 && hdist - dist->count[0] > 1)) return PI_ERR_DATA;
  return PI_OK;
}

static huff_t g_fixed_lit, g_fixed_dist;
static int g_fixed_ready = 0;
static void fixed_codes(void) {
  /* (idempotent: two threads that race here write the same bytes) */
  uint8_t len[288];
  int s = 0;
  for (; s < 144; ++s) len[s] = 8;
  for (; s < 256; ++s) len[s] = 9;
  for (; s < 280; ++s) len[s] = 7;
  for (; s < 288; ++s) len[s] = 8;
  huff_t l, d;
  huff_build(&l, len, 288);
  huff_widen(&l);
  for (s = 0; s < 30; ++s) len[s] = 5;
  huff_build(&d, len, 30);
  huff_widen_dist(&d);
  huff_combine(&l, &d);
  if (!g_fixed_ready) {
    g_fixed_lit = l;
    g_fixed_dist = d;
    __sync_synchronize();
    g_fixed_ready = 1;
  }
}

/* The symbols of one Huffman-coded block up to its end-of-block code.  out == NULL: dry run (block finder).
 * *o is the number of symbols the chunk has produced so far. */
static int coded_block(bits_t *b, const huff_t *lit, const huff_t *dist, uint16_t *out, size_t cap, size_t *o, size_t dry_limit) {
  size_t at = *o;
  for (;;) {
    bits_refill(b);
    if (b->pos > b->n + 16) return PI_ERR_INPUT; /* (zero bits behind the input decode to something for ever) */
    if (!out && at > dry_limit) return PI_ERR_DATA; /* (block finder: no real block is this long) */
    int sym = huff_decode(b, lit);
    if (sym < 0) return PI_ERR_DATA;
    if (sym < 256) {
      if (out) {
        if (at >= cap) return PI_ERR_SPACE;
        out[at] = (uint16_t)sym;
      }
      ++at;
      /* a second literal from the same refill: most of FASTQ is literals or short matches */
      if (b->cnt >= MAXBITS + 33) {
        sym = huff_decode(b, lit);
        if (sym < 0) return PI_ERR_DATA;
        if (sym < 256) {
          if (out) {
            if (at >= cap) return PI_ERR_SPACE;
            out[at] = (uint16_t)sym;
          }
          ++at;
          continue;
        }
      } else {
        continue;
      }
    }
    if (sym == 256) break;
    sym -= 257;
    if (sym >= 29) return PI_ERR_DATA;
    /* (after a refill the buffer holds >= 57 bits: 15 + 5 + 15 + 13 = 48 are the most one match takes) */
    const int length = LEN_BASE[sym] + (int)bits_take(b, LEN_EXTRA[sym]);
    const int ds = huff_decode(b, dist);
    if (ds < 0 || ds >= 30) return PI_ERR_DATA;
    const int d = DIST_BASE[ds] + (int)bits_take(b, DIST_EXTRA[ds]);
    if (out) {
      if (at + (size_t)length > cap) return PI_ERR_SPACE;
      if ((size_t)d <= at) {
        const uint16_t *src = out + at - d;
        uint16_t *dst = out + at;
        if (d >= length) {
          memcpy(dst, src, (size_t)length * 2);
        } else {
          for (int j = 0; j < length; ++j) dst[j] = src[j];
        }
      } else {
        /* reaches in front of the chunk: markers for that part, then the chunk's own symbols */
        for (int j = 0; j < length; ++j) {
          const int64_t from = (int64_t)at + j - d;
          out[at + j] = from < 0 ? (uint16_t)(0x8000u | (uint32_t)(WINDOW + from)) : out[from];
        }
      }
    }
    at += (size_t)length;
  }
  *o = at;
  return bits_pos(b) > (uint64_t)b->n * 8u ? PI_ERR_INPUT : PI_OK;
}

/* The same for the real decoder.  gzip -1 turns FASTQ into SHORT MATCHES (bases and binned qualities: five bytes per
 * match on average, hardly a literal), so what bounds the loop is the chain look-up -> shift -> look-up of a match's
 * two codes: the wide tables hold the bits to consume in ONE field that the shift uses as it is, the extra bits are
 * cut out of a copy of the bit buffer beside the chain, one refill serves a whole match (56 bits >= 15 + 5 + 15 + 13),
 * and a match of up to 16 symbols is two unconditional 16-byte copies.  (It wants 320 symbols of room in front of
 * every step: PI_ERR_SPACE earlier than strictly needed.) */
#define PI_NAME coded_block_fast
#define PI_T uint16_t
#define PI_MARKERS 1
#include "pinflate_loop.h"
#undef PI_NAME
#undef PI_T
#undef PI_MARKERS
#define PI_NAME coded_block_bytes
#define PI_T uint8_t
#define PI_MARKERS 0
#include "pinflate_loop.h"
#undef PI_NAME
#undef PI_T
#undef PI_MARKERS

/* Is `bitpos` plausibly the start of a block?  0 no; 1 dynamic with a valid header; 2 stored with matching lengths;
 * 3 fixed. */
static int plausible_header(const uint8_t *in, size_t n, uint64_t bitpos) {
  if ((bitpos >> 3) + 4 > n) return 0;
  bits_t b;
  bits_init(&b, in, n, bitpos);
  const uint32_t hdr = bits_take(&b, 3);
  const uint32_t type = hdr >> 1;
  if (type == 3) return 0;
  if (type == 0) {
    const int pad = (int)(b.cnt & 7);
    bits_take(&b, pad);
    bits_refill(&b);
    const uint32_t len = bits_take(&b, 16), nlen = bits_take(&b, 16);
    return (len ^ nlen) == 0xffffu ? 2 : 0;
  }
  if (type == 1) return 3;
  huff_t lit, dist;
  return dynamic_header(&b, &lit, &dist) == PI_OK && bits_pos(&b) <= (uint64_t)n * 8u ? 1 : 0;
}

int64_t csh_deflate_find_block(const uint8_t *in, int64_t n_bytes, int64_t from_bit, int64_t until_bit) {
  const size_t n = (size_t)n_bytes;
  huff_t *lit = (huff_t *)malloc(sizeof(huff_t)), *dist = (huff_t *)malloc(sizeof(huff_t));
  if (!lit || !dist) {
    free(lit);
    free(dist);
    return -1;
  }
  int64_t found = -1;
  for (int64_t p = from_bit; p < until_bit && ((uint64_t)p >> 3) + 8 < n; ++p) {
    /* BFINAL = 0, BTYPE = 10: bits 0, 0, 1 in stream order */
    const uint32_t three = (uint32_t)((in[p >> 3] | ((uint32_t)in[(p >> 3) + 1] << 8)) >> (p & 7)) & 7u;
    if (three != 4u) continue;
    bits_t b;
    bits_init(&b, in, n, (uint64_t)p + 3);
    if (dynamic_header(&b, lit, dist) != PI_OK) continue;
    size_t o = 0;
    if (coded_block(&b, lit, dist, NULL, 0, &o, (size_t)8 << 20) != PI_OK) continue;
    if (o == 0) continue; /* (an empty block proves little) */
    if (!plausible_header(in, n, bits_pos(&b))) continue;
    found = p;
    break;
  }
  free(lit);
  free(dist);
  return found;
}

/* Decode from the block boundary `start_bit` up to the first block boundary at or behind `stop_bit`, or the end of the
 * final block.  out: `cap` symbols.  *end_bit: where it stopped; *n_out: symbols written; *final: 1 when the final block
 * was decoded (end_bit is then the bit behind it).  PI_OK / PI_ERR_*. */
static int inflate_blocks(const uint8_t *in, int64_t n_bytes, int64_t start_bit, int64_t stop_bit, void *out_any, int wide,
                          int64_t cap, int64_t *end_bit, int64_t *n_out, int32_t *final) {
  const size_t n = (size_t)n_bytes;
  uint16_t *out16 = wide ? (uint16_t *)out_any : NULL;
  uint8_t *out8 = wide ? NULL : (uint8_t *)out_any;
  if (!g_fixed_ready) fixed_codes();
  huff_t *lit = (huff_t *)malloc(sizeof(huff_t)), *dist = (huff_t *)malloc(sizeof(huff_t));
  if (!lit || !dist) {
    free(lit);
    free(dist);
    return PI_ERR_SPACE;
  }
  bits_t b;
  bits_init(&b, in, n, (uint64_t)start_bit);
  size_t o = 0;
  int rc = PI_OK, last = 0;
  for (;;) {
    if (bits_pos(&b) + 3 > (uint64_t)n * 8u) {
      rc = PI_ERR_INPUT;
      break;
    }
    bits_refill(&b);
    last = (int)bits_take(&b, 1);
    const uint32_t type = bits_take(&b, 2);
    if (type == 0) {
      bits_take(&b, b.cnt & 7);
      bits_refill(&b);
      const uint32_t len = bits_take(&b, 16), nlen = bits_take(&b, 16);
      if ((len ^ nlen) != 0xffffu) {
        rc = PI_ERR_DATA;
        break;
      }
      const uint64_t at = bits_pos(&b) >> 3; /* byte-aligned here */
      if (at + len > n) {
        rc = PI_ERR_INPUT;
        break;
      }
      if (o + len > (size_t)cap) {
        rc = PI_ERR_SPACE;
        break;
      }
      if (wide)
        for (uint32_t i = 0; i < len; ++i) out16[o + i] = in[at + i];
      else
        memcpy(out8 + o, in + at, len);
      o += len;
      bits_init(&b, in, n, (at + len) * 8u);
    } else if (type == 1) {
      rc = wide ? coded_block_fast(&b, &g_fixed_lit, &g_fixed_dist, out16, (size_t)cap, &o)
                : coded_block_bytes(&b, &g_fixed_lit, &g_fixed_dist, out8, (size_t)cap, &o);
      if (rc) break;
    } else if (type == 2) {
      rc = dynamic_header(&b, lit, dist);
      if (rc) break;
      huff_widen(lit);
      huff_widen_dist(dist);
      huff_combine(lit, dist);
      rc = wide ? coded_block_fast(&b, lit, dist, out16, (size_t)cap, &o) : coded_block_bytes(&b, lit, dist, out8, (size_t)cap, &o);
      if (rc) break;
    } else {
      rc = PI_ERR_DATA;
      break;
    }
    if (last) break;
    if ((int64_t)bits_pos(&b) >= stop_bit) {
      /* Stop at a boundary the FINDER can name: it only looks for dynamic blocks that are not the final one (bits 0, 0,
       * 1 in stream order).  A stored or fixed block here -- pigz and every other writer that joins pieces with sync
       * flushes puts an empty stored block every 128 KB of text -- is decoded with this chunk; stopping in front of it
       * made the chunk behind it (which starts at the dynamic block BEHIND the flush) fail its proof and fall back to the
       * serial decoder: a third of the chunks of a pigz file (tools/micro/single_member_pieces.py: 3.1 instead of 9.4
       * GB/s of text). */
      const uint64_t p = bits_pos(&b);
      if ((p >> 3) + 1 >= n) break;
      const uint32_t three = (uint32_t)((in[p >> 3] | ((uint32_t)in[(p >> 3) + 1] << 8)) >> (p & 7)) & 7u;
      if (three == 4u) break;
    }
  }
  free(lit);
  free(dist);
  *end_bit = (int64_t)bits_pos(&b);
  *n_out = (int64_t)o;
  *final = (rc == PI_OK && last) ? 1 : 0;
  return rc;
}

int csh_inflate_chunk(const uint8_t *in, int64_t n_bytes, int64_t start_bit, int64_t stop_bit, uint16_t *out, int64_t cap,
                      int64_t *end_bit, int64_t *n_out, int32_t *final) {
  return inflate_blocks(in, n_bytes, start_bit, stop_bit, out, 1, cap, end_bit, n_out, final);
}

/* A whole deflate stream from its first block (bit `start_bit`: behind a gzip member's header) to the end of its final
 * block, as plain bytes: nothing precedes it, so no markers and no second pass.  out: `cap` bytes (the loop wants 320
 * bytes of room in front of every step: PI_ERR_SPACE a little earlier than strictly needed).  *end_bit: the bit behind
 * the final block (the member's trailer starts at the next byte boundary).  PI_OK only when the final block was
 * decoded. */
int csh_inflate_stream(const uint8_t *in, int64_t n_bytes, int64_t start_bit, uint8_t *out, int64_t cap, int64_t *end_bit,
                       int64_t *n_out) {
  int32_t final = 0;
  const int rc = inflate_blocks(in, n_bytes, start_bit, INT64_MAX, out, 0, cap, end_bit, n_out, &final);
  if (rc != PI_OK) return rc;
  return final ? PI_OK : PI_ERR_INPUT;
}

/* symbols -> bytes.  `window`: the WINDOW bytes in front of the chunk (marker i stands for window[i]); NULL when
 * nothing can precede the chunk (first chunk of a stream: a marker is corrupt data then).  Returns >= 0 (a rough count
 * of markers met, diagnostics only), -1 for a marker without a window. */
int64_t csh_resolve_markers(const uint16_t *sym, int64_t n, const uint8_t *window, uint8_t *out) {
  int64_t markers = 0;
  if (!window) {
    uint16_t any = 0;
    for (int64_t i = 0; i < n; ++i) {
      any |= sym[i];
      out[i] = (uint8_t)sym[i];
    }
    return (any & 0x8000u) ? -1 : 0;
  }
  /* one table for both kinds of symbol (a branch per symbol mispredicts half the time when four symbols in five are
   * markers): bytes stand for themselves, 0x8000 | i for window[i] */
  uint8_t *lut = (uint8_t *)malloc(65536);
  if (!lut) return -1;
  for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)i;
  memset(lut + 256, 0, 0x8000 - 256);
  memcpy(lut + 0x8000, window, WINDOW);
  int64_t i = 0;
  for (; i + 16 <= n; i += 16) {
    /* sixteen symbols without a marker among them: one pack, one store (x86-64: SSE2 is part of the baseline) */
    __m128i a, b;
    memcpy(&a, sym + i, 16);
    memcpy(&b, sym + i + 8, 16);
    if ((_mm_movemask_epi8(_mm_or_si128(a, b)) & 0xAAAA) == 0) {
      const __m128i packed = _mm_packus_epi16(a, b);
      memcpy(out + i, &packed, 16);
      continue;
    }
    for (int j = 0; j < 16; j += 8) {
      const uint16_t *q = sym + i + j;
      uint16_t s0 = q[0], s1 = q[1], s2 = q[2], s3 = q[3], s4 = q[4], s5 = q[5], s6 = q[6], s7 = q[7];
      uint8_t *w = out + i + j;
      w[0] = lut[s0];
      w[1] = lut[s1];
      w[2] = lut[s2];
      w[3] = lut[s3];
      w[4] = lut[s4];
      w[5] = lut[s5];
      w[6] = lut[s6];
      w[7] = lut[s7];
      markers += ((s0 | s1 | s2 | s3 | s4 | s5 | s6 | s7) >> 15);  /* (a count of 8-symbol groups with markers: diagnostics only) */
    }
  }
  for (; i < n; ++i) {
    out[i] = lut[sym[i]];
    markers += sym[i] >> 15;
  }
  free(lut);
  return markers;
}

/* The window behind a chunk: the last WINDOW bytes of (window in front of it, its symbols resolved).  `next` may
 * alias nothing else.  Same return value as csh_resolve_markers (for the part it looks at). */
int64_t csh_next_window(const uint16_t *sym, int64_t n, const uint8_t *window, uint8_t *next) {
  if (n >= WINDOW) return csh_resolve_markers(sym + (n - WINDOW), WINDOW, window, next);
  const int64_t keep = WINDOW - n;
  if (window)
    memmove(next, window + n, (size_t)keep);
  else
    memset(next, 0, (size_t)keep);
  /* (markers of a short chunk index the OLD window: resolve before the bytes above are looked at -- they were
   * copied, `window` itself is untouched) */
  return csh_resolve_markers(sym, n, window, next + keep);
}

/* First position p in [from, to) with buf[p .. p+2] == 1f 8b 08 (a gzip member header with the deflate method), -1 if
 * none; `to` may be at most n - 2.  (mmap.find holds the interpreter lock and walks 1 GB/s; this is memchr.) */
int64_t csh_find_gzip_magic(const uint8_t *buf, int64_t from, int64_t to) {
  while (from < to) {
    const uint8_t *hit = (const uint8_t *)memchr(buf + from, 0x1f, (size_t)(to - from));
    if (!hit) return -1;
    if (hit[1] == 0x8b && hit[2] == 0x08) return (int64_t)(hit - buf);
    from = (int64_t)(hit - buf) + 1;
  }
  return -1;
}
