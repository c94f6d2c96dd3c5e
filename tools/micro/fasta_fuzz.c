/* AddressSanitizer / UBSan fuzz of csh_fasta_to_fastq (csrc/cutseq_host.c) on the CPU:
 *   gcc -O1 -g -fsanitize=address,undefined -o /tmp/fasta_fuzz tools/micro/fasta_fuzz.c cutseq_amd/csrc/cutseq_host.c -lpthread -lm
 *   /tmp/fasta_fuzz [iterations] [seed]
 * Random FASTA-like text (headers, wrapped sequences, blank lines, CR LF, comments, junk) goes through the converter
 * (a) whole, with exact-size buffers, and (b) in random pieces with the unconsumed tail carried over, the way
 * codec.FastaSource feeds it; both must give the same bytes, every record must have four lines with equal sequence and
 * quality lengths, and nothing may be read or written out of bounds.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int64_t csh_fasta_to_fastq(const uint8_t *src, int64_t n, uint8_t *dst, int64_t cap, int final, int in_record,
                           int64_t *consumed, int64_t *records, int64_t *err_line);

static uint64_t rng_state;
static uint32_t rnd(void) {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(rng_state >> 33);
}

static int64_t make_text(uint8_t *buf, int64_t cap, int valid) {
  int64_t n = 0;
  const int recs = rnd() % 12;
  if (rnd() % 4 == 0) n += sprintf((char *)buf + n, "# comment %u\n", rnd() % 100);
  for (int r = 0; r < recs && n + 400 < cap; ++r) {
    if (rnd() % 5 == 0) buf[n++] = '\n';
    n += sprintf((char *)buf + n, ">%sname%u %s", rnd() % 7 == 0 ? " " : "", rnd() % 1000, rnd() % 3 ? "desc" : "");
    if (rnd() % 4 == 0) buf[n++] = ' ';
    if (rnd() % 3 == 0) buf[n++] = '\r';
    buf[n++] = '\n';
    const int lines = rnd() % 4;
    for (int l = 0; l < lines; ++l) {
      const int len = rnd() % 70;
      for (int i = 0; i < len; ++i) buf[n++] = "ACGTNacgtn"[rnd() % 10];
      if (rnd() % 6 == 0) buf[n++] = '\t';
      if (rnd() % 3 == 0) buf[n++] = '\r';
      if (l + 1 < lines || rnd() % 8) buf[n++] = '\n';
    }
  }
  if (!valid && n > 0) {  /* junk: random bytes somewhere */
    const int hits = 1 + rnd() % 4;
    for (int h = 0; h < hits; ++h) buf[rnd() % n] = (uint8_t)rnd();
  }
  return n;
}

static int check_records(const uint8_t *out, int64_t n) {
  int64_t pos = 0;
  while (pos < n) {
    int64_t len[4];
    for (int l = 0; l < 4; ++l) {
      const uint8_t *nl = memchr(out + pos, '\n', (size_t)(n - pos));
      if (!nl) return 0;
      len[l] = nl - (out + pos);
      if (l == 0 && out[pos] != '@') return 0;
      if (l == 2 && (len[l] != 1 || out[pos] != '+')) return 0;
      pos = nl - out + 1;
    }
    if (len[1] != len[3]) return 0;
  }
  return 1;
}

int main(int argc, char **argv) {
  const long iters = argc > 1 ? atol(argv[1]) : 20000;
  rng_state = argc > 2 ? (uint64_t)atoll(argv[2]) : 12345;
  long errors = 0, converted = 0;
  for (long it = 0; it < iters; ++it) {
    const int valid = rnd() % 3 != 0;
    uint8_t *text = malloc(8192);
    const int64_t n = make_text(text, 8192, valid);
    uint8_t *src = malloc((size_t)n ? (size_t)n : 1);  /* exact size: ASan sees any over-read */
    memcpy(src, text, (size_t)n);
    free(text);
    const int64_t cap = 3 * n + 16;
    uint8_t *whole = malloc((size_t)cap);
    int64_t consumed, records, err_line;
    const int64_t w = csh_fasta_to_fastq(src, n, whole, cap, 1, 0, &consumed, &records, &err_line);
    if (w == -1) {
      fprintf(stderr, "iteration %ld: cap too small\n", it);
      return 1;
    }
    if (w >= 0) {
      ++converted;
      if (consumed != n || !check_records(whole, w)) {
        fprintf(stderr, "iteration %ld: malformed output (consumed %lld of %lld)\n", it, (long long)consumed, (long long)n);
        return 1;
      }
      /* in pieces, carrying the unconsumed tail */
      uint8_t *pieces = malloc((size_t)cap);
      int64_t pw = 0, fed = 0, carry_n = 0;
      uint8_t *carry = malloc((size_t)n + 1);
      while (fed < n || carry_n) {
        const int64_t take = fed < n ? 1 + rnd() % (n - fed) : 0;
        const int final = fed + take >= n;
        uint8_t *data = malloc((size_t)(carry_n + take) ? (size_t)(carry_n + take) : 1);
        memcpy(data, carry, (size_t)carry_n);
        memcpy(data + carry_n, src + fed, (size_t)take);
        const int64_t dn = carry_n + take;
        fed += take;
        uint8_t *out = malloc((size_t)(3 * dn + 16));
        int64_t c2, r2, e2;
        const int64_t p = csh_fasta_to_fastq(data, dn, out, 3 * dn + 16, final, 0, &c2, &r2, &e2);
        if (p < 0 || c2 > dn) {
          fprintf(stderr, "iteration %ld: piecewise call failed (%lld)\n", it, (long long)p);
          return 1;
        }
        memcpy(pieces + pw, out, (size_t)p);
        pw += p;
        carry_n = dn - c2;
        memcpy(carry, data + c2, (size_t)carry_n);
        free(out);
        free(data);
        if (final) break;
      }
      if (pw != w || memcmp(pieces, whole, (size_t)w)) {
        fprintf(stderr, "iteration %ld: piecewise output differs (%lld vs %lld bytes)\n", it, (long long)pw, (long long)w);
        return 1;
      }
      free(pieces);
      free(carry);
    } else {
      ++errors;
    }
    free(whole);
    free(src);
  }
  printf("%ld iterations: %ld converted, %ld rejected as malformed, no sanitizer report\n", iters, converted, errors);
  return 0;
}
