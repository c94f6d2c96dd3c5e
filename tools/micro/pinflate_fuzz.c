/* Sanitizer fuzz of the host decoder (csrc/pinflate.c): valid, truncated and bit-flipped deflate streams, exact-size heap
 * buffers for input and output (too small ones included), through csh_inflate_stream, csh_deflate_find_block +
 * csh_inflate_chunk + csh_resolve_markers and csh_find_gzip_magic.  AddressSanitizer / UBSan report and abort on the first
 * out-of-bounds access; valid streams must decode to their text.
 *   gcc -O1 -g -fsanitize=address,undefined -std=gnu11 -Icutseq_amd/csrc -o /tmp/pinflate_fuzz tools/micro/pinflate_fuzz.c \
 *       cutseq_amd/csrc/pinflate.c -lz && /tmp/pinflate_fuzz 30000
 * (30 000 iterations, three minutes: "fuzz done: 10974 accepted, 19026 rejected", no report -- end of round 3) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
int csh_inflate_stream(const uint8_t *in, int64_t n_bytes, int64_t start_bit, uint8_t *out, int64_t cap, int64_t *end_bit, int64_t *n_out);
int csh_inflate_chunk(const uint8_t *in, int64_t n_bytes, int64_t start_bit, int64_t stop_bit, uint16_t *out, int64_t cap, int64_t *end_bit, int64_t *n_out, int32_t *final);
int64_t csh_deflate_find_block(const uint8_t *in, int64_t n_bytes, int64_t from_bit, int64_t until_bit);
int64_t csh_resolve_markers(const uint16_t *sym, int64_t n, const uint8_t *window, uint8_t *out);
int64_t csh_find_gzip_magic(const uint8_t *buf, int64_t from, int64_t to);
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  long ok = 0, err = 0;
  for (int it = 0; it < iters; ++it) {
    /* a plausible FASTQ-ish text */
    size_t n = 200 + rnd() % 60000;
    uint8_t *text = malloc(n);
    for (size_t i = 0; i < n; ++i) { uint32_t r = rnd(); text[i] = (r & 7) == 0 ? "FFFF:,#\n"[(r >> 3) & 7] : "ACGT"[(r >> 3) & 3]; }
    uLongf cl = compressBound(n) + 64;
    uint8_t *comp = malloc(cl);
    z_stream z; memset(&z, 0, sizeof z);
    int level = 1 + rnd() % 9; int strat = (rnd() % 5 == 0) ? Z_FIXED : (rnd() % 7 == 0 ? Z_HUFFMAN_ONLY : Z_DEFAULT_STRATEGY);
    deflateInit2(&z, level, Z_DEFLATED, -15, 8, strat);
    z.next_in = text; z.avail_in = n; z.next_out = comp; z.avail_out = cl;
    deflate(&z, Z_FINISH); size_t cn = z.total_out; deflateEnd(&z);
    /* exact-size heap copy of the input (the library may read up to 8 bytes past? it must not) */
    int mode = rnd() % 4;
    size_t in_n = cn;
    if (mode == 1) in_n = cn ? rnd() % cn : 0;             /* truncated */
    uint8_t *in = malloc(in_n ? in_n : 1);
    memcpy(in, comp, in_n);
    if (mode >= 2) for (int k = 0, m = 1 + rnd() % 8; k < m && in_n; ++k) in[rnd() % in_n] ^= (uint8_t)(1u << (rnd() % 8)); /* bit flips */
    size_t cap = (mode == 3) ? n / 2 + rnd() % (n / 2 + 1) : n + 400 + rnd() % 100;
    uint8_t *out = malloc(cap ? cap : 1);
    int64_t end, no;
    int rc = csh_inflate_stream(in, (int64_t)in_n, mode >= 2 && (rnd() & 3) == 0 ? rnd() % (in_n * 8 + 1) : 0, out, (int64_t)cap, &end, &no);
    if (rc == 0) { ok++; if (mode == 0 && ((size_t)no != n || memcmp(out, text, n))) { printf("MISMATCH it %d\n", it); return 1; } } else err++;
    if (no < 0 || (rc == 0 && (size_t)no > cap)) { printf("BAD n_out\n"); return 1; }
    /* chunk decoder from a found block, symbols + resolve */
    uint16_t *sym = malloc((cap ? cap : 1) * 2);
    int64_t p = csh_deflate_find_block(in, (int64_t)in_n, 0, (int64_t)in_n * 8);
    if (p >= 0) {
      int32_t fin;
      rc = csh_inflate_chunk(in, (int64_t)in_n, p, (int64_t)in_n * 8, sym, (int64_t)cap, &end, &no, &fin);
      if (rc == 0 && no > 0 && (size_t)no <= cap) { uint8_t win[32768]; memset(win, 'A', sizeof win); csh_resolve_markers(sym, no, win, out); }
    }
    csh_find_gzip_magic(in, 0, (int64_t)in_n > 2 ? (int64_t)in_n - 2 : 0);
    free(sym); free(out); free(in); free(comp); free(text);
  }
  printf("fuzz done: %ld accepted, %ld rejected\n", ok, err);
  return 0;
}
