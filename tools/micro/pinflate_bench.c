/* Raw rate of the chunk decoder (csrc/pinflate.c) on one single-member .gz: best of N on a 6 MB piece of the compressed
 * stream from a block boundary in its middle (marker mode), and of the resolve pass behind it.
 *   gcc -O3 -o pinflate_bench tools/micro/pinflate_bench.c cutseq_amd/csrc/pinflate.c && ./pinflate_bench file.gz [threads] */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
int64_t csh_deflate_find_block(const uint8_t *in, int64_t n_bytes, int64_t from_bit, int64_t until_bit);
int csh_inflate_chunk(const uint8_t *in, int64_t n_bytes, int64_t start_bit, int64_t stop_bit, uint16_t *out, int64_t cap,
                      int64_t *end_bit, int64_t *n_out, int32_t *final);
int64_t csh_resolve_markers(const uint16_t *sym, int64_t n, const uint8_t *window, uint8_t *out);
static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}
static uint8_t *in;
static long n;
static int64_t start_bit;
static int reps = 20;
typedef struct { double decode, resolve, wall; int64_t out; } result_t;
static void *work(void *arg) {
  result_t *r = (result_t *)arg;
  const int64_t cap = 100LL << 20;
  uint16_t *out = malloc(cap * 2);
  uint8_t *bytes = malloc(cap), *win = calloc(32768, 1);
  memset(out, 1, cap * 2);
  memset(bytes, 1, cap);
  r->decode = r->resolve = 0;
  const double w0 = now();
  for (int rep = 0; rep < reps; ++rep) {
    int64_t end, no;
    int32_t fin;
    const double t0 = now();
    if (csh_inflate_chunk(in, n - 8, start_bit, start_bit + (6LL << 20) * 8, out, cap, &end, &no, &fin)) exit(2);
    const double t1 = now();
    csh_resolve_markers(out, no, win, bytes);
    const double t2 = now();
    const double d = no / (t1 - t0) / 1e6, s = no / (t2 - t1) / 1e6;
    if (d > r->decode) r->decode = d;
    if (s > r->resolve) r->resolve = s;
    r->out = no;
  }
  r->wall = now() - w0;
  return NULL;
}
int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 1;
  fseek(f, 0, SEEK_END);
  n = ftell(f);
  fseek(f, 0, SEEK_SET);
  in = malloc(n + 64);
  if (fread(in, 1, n, f) != (size_t)n) return 1;
  memset(in + n, 0, 64);
  const int threads = argc > 2 ? atoi(argv[2]) : 1;
  start_bit = csh_deflate_find_block(in, n - 8, (n / 3) * 8, (n / 3) * 8 + (8 << 20));
  if (start_bit < 0) return 3;
  pthread_t th[256];
  result_t res[256];
  for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, work, &res[t]);
  double agg = 0;
  for (int t = 0; t < threads; ++t) {
    pthread_join(th[t], NULL);
    agg += (double)res[t].out * reps / res[t].wall / 1e6;
  }
  printf("%3d thread(s): best decode %.0f MB/s, best resolve %.0f MB/s per thread; all threads, decode + resolve: %.0f MB/s of text\n",
         threads, res[0].decode, res[0].resolve, agg);
  return 0;
}
