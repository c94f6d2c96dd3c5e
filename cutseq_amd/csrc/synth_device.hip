/*
 * synth_device.hip -- the seeded synthetic read generator of csrc/cutseq_host.c (synth_one / finish_read) as a HIP
 * kernel: bench.py and the tests fill device-resident batches without a host copy ("counter-based so it
 * can run on device or host identically").  Built into libcutseq_synth.so by cutseq_amd/build.py; declared in
 * include/cutseq_synth.h.  Nothing here trims reads, and the trimming library does not link it.
 *
 * Same bytes as the host generator for every pair (tests/test_gpu_synth.py): every pair has its own splitmix64 stream
 * keyed by (seed, global pair index), and the stream is consumed in the host routine's order.  What differs is the
 * shape: the host routine builds head / tail / insert / template arrays and copies them about; here a lane per pair
 * keeps NO arrays -- the k-th draw of a counter-based stream is mix64(state + (k + 1) * golden), so "the j-th base of the
 * insert" is a function of j, and a template position is resolved through the segment it falls into (an indel shifts
 * the index).  The doubles are compared exactly as on the host (no fused multiply-add anywhere: products and sums of
 * the host code are never contracted).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/cutseq_synth.h"

#pragma clang fp contract(off)

namespace {

constexpr int MAX_TPL = 4096;  /* the host generator's template bound */
constexpr int MAX_LIT = 64;    /* adapters / inline barcodes, bytes   */

struct DevSynth {
  uint32_t read_len, stride;
  uint64_t seed, first_index;
  uint8_t p5_fw[MAX_LIT], p7_fw[MAX_LIT], p5_rc[MAX_LIT], p7_rc[MAX_LIT], inline5[MAX_LIT], inline3[MAX_LIT];
  int32_t n_p5_fw, n_p7_fw, n_p5_rc, n_p7_rc, n_i5, n_i3;
  int32_t umi5, umi3, mask5, mask3, strand, single_end;
  double adapter_fraction, partial_fraction, poly_fraction, art5_fraction, sub_rate, indel_frac, n_rate;
};

constexpr uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct Rng {
  uint64_t s;
  __device__ __forceinline__ uint64_t next() { s += GOLDEN; return mix64(s); }
  __device__ __forceinline__ uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
  __device__ __forceinline__ double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  __device__ __forceinline__ void skip(int k) { s += GOLDEN * (uint64_t)k; }
};
/* "ACGT"[rng_below(r, 4)] of the (k + 1)-th draw behind state s0: ((x >> 32) * 4) >> 32 == x >> 62 */
__device__ __forceinline__ uint8_t base_of(uint32_t two_bits) { return (uint8_t)((0x54474341u >> (8 * two_bits)) & 0xFF); }
__device__ __forceinline__ uint8_t drawn_base(uint64_t s0, int k) { return base_of((uint32_t)(mix64(s0 + GOLDEN * (uint64_t)(k + 1)) >> 62)); }
__device__ __forceinline__ uint8_t comp(uint8_t c) {
  return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
}

/* the molecule of one pair: everything the host routine keeps in head[] / tail[] / insert[] */
struct Molecule {
  uint64_t s_head, s_tail, s_ins;
  int h, t, ins_len, n_rand_tail, poly_len, strand;
  const DevSynth *p;
  __device__ __forceinline__ uint8_t head(int j) const { return j < p->n_i5 ? p->inline5[j] : drawn_base(s_head, j - p->n_i5); }
  __device__ __forceinline__ uint8_t tail(int j) const { return j < n_rand_tail ? drawn_base(s_tail, j) : p->inline3[j - n_rand_tail]; }
  __device__ __forceinline__ uint8_t insert(int j) const {
    if (strand > 0 && j >= ins_len - poly_len) return 'A';
    if (strand < 0 && j < poly_len) return 'T';
    return drawn_base(s_ins, j);
  }
  /* template of mate 1: [artefact p5.fw] head insert tail p7.fw ; of mate 2: [artefact p7.rc] rc(tail) rc(insert) rc(head) p5.rc */
  __device__ __forceinline__ uint8_t tpl1(int i, int art) const {
    if (i < art) return p->p5_fw[i];
    i -= art;
    if (i < h) return head(i);
    i -= h;
    if (i < ins_len) return insert(i);
    i -= ins_len;
    if (i < t) return tail(i);
    return p->p7_fw[i - t];
  }
  __device__ __forceinline__ uint8_t tpl2(int i, int art) const {
    if (i < art) return p->p7_rc[i];
    i -= art;
    if (i < t) return comp(tail(t - 1 - i));
    i -= t;
    if (i < ins_len) return comp(insert(ins_len - 1 - i));
    i -= ins_len;
    if (i < h) return comp(head(h - 1 - i));
    return p->p5_rc[i - h];
  }
};

template <int MATE>
__device__ void finish_read(Rng &r, const DevSynth &p, const Molecule &mol, int art, int tlen, int ad_lo, int ad_hi,
                            uint8_t *__restrict__ seq, uint8_t *__restrict__ qual, uint16_t *__restrict__ len) {
  const int L = (int)p.read_len;
  /* one single-base indel inside the adapter region of a few adapter-bearing reads: position d of the template goes
     (deletion: sources shift by one behind it) or is a new base (insertion: sources shift back behind it) */
  int del_at = MAX_TPL + 1, ins_at = MAX_TPL + 1;
  uint8_t ins_base = 0;
  if (ad_lo < L && ad_hi > ad_lo && r.unit() < p.indel_frac) {
    const int hi = ad_hi < L ? ad_hi : L;
    const int d = ad_lo + (int)r.below((uint32_t)(hi - ad_lo));
    if (r.next() & 1) {
      del_at = d;
      tlen--;
    } else if (tlen < MAX_TPL) {
      ins_at = d;
      ins_base = base_of(r.below(4));
      tlen++;
    }
  }
  int tail_from = L + 1;
  if (r.unit() < 0.20) tail_from = (int)(L * 0.6) + (int)r.below((uint32_t)(L - (int)(L * 0.6)) + 1u);
  uint32_t sw = 0, qw = 0;
  uint32_t *seq4 = (uint32_t *)seq, *qual4 = (uint32_t *)qual;
  for (int i = 0; i < L; i++) {
    uint8_t c;
    if (i < tlen) {
      if (i == ins_at) {
        c = ins_base;
      } else {
        const int src = i > ins_at ? i - 1 : (i >= del_at ? i + 1 : i);
        c = MATE == 1 ? mol.tpl1(src, art) : mol.tpl2(src, art);
      }
    } else {
      c = base_of(r.below(4));
    }
    const uint64_t x = r.next();
    const double u = (double)(x >> 40) * (1.0 / 16777216.0);
    const double w = (double)((x >> 16) & 0xFFFFFF) * (1.0 / 16777216.0);
    if (u < p.sub_rate) c = base_of((uint32_t)(x & 3));
    uint8_t q = 'I';
    if (w < 0.08)
      q = '-';
    else if (w < 0.16)
      q = '9';
    if (i >= tail_from) {
      const double t = (double)(r.next() >> 40) * (1.0 / 16777216.0);
      if (t < 0.55)
        q = '#';
      else if (t < 0.80)
        q = '-';
    }
    if (u >= p.sub_rate && u < p.sub_rate + p.n_rate) {
      c = 'N';
      q = '#';
    }
    sw |= (uint32_t)c << (8 * (i & 3));
    qw |= (uint32_t)q << (8 * (i & 3));
    if ((i & 3) == 3) {
      seq4[i >> 2] = sw;
      qual4[i >> 2] = qw;
      sw = qw = 0;
    }
  }
  /* padding behind the read: 'N' in the sequence row, 0 in the quality row (as the host form) */
  if (L & 3) {
    seq4[L >> 2] = sw | (0x4E4E4E4Eu << (8 * (L & 3)));
    qual4[L >> 2] = qw;
  }
  for (uint32_t i = ((uint32_t)L + 3u) >> 2; i < (p.stride >> 2); i++) {
    seq4[i] = 0x4E4E4E4Eu;
    qual4[i] = 0;
  }
  *len = (uint16_t)L;
}

__global__ void __launch_bounds__(256) synth_kernel(const DevSynth p, uint64_t n, uint8_t *__restrict__ seq1, uint8_t *__restrict__ qual1,
                                                    uint16_t *__restrict__ len1, uint8_t *__restrict__ seq2,
                                                    uint8_t *__restrict__ qual2, uint16_t *__restrict__ len2) {
  const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const uint64_t index = p.first_index + row;
  Rng r;
  r.s = mix64(p.seed ^ mix64(index + 0x632BE59BD9B4E019ull));
  const int L = (int)p.read_len;
  const int head_len = p.n_i5 + p.umi5 + p.mask5, tail_len = p.mask3 + p.umi3 + p.n_i3;
  const int max_ins = L + 40;
  const int full = L - head_len;
  int ins_len;
  const double u = r.unit(), v = r.unit();
  if (u < p.adapter_fraction) {
    const int vis = full - tail_len - p.n_p7_fw;
    int span = vis - 25;
    if (span < 1) span = 1;
    ins_len = vis - (int)(v * v * span);
  } else if (u < p.adapter_fraction + p.partial_fraction) {
    ins_len = full - tail_len - 3 - (int)r.below(17);
  } else {
    ins_len = full + (int)r.below(40);
  }
  if (ins_len < 1) ins_len = 1;
  if (ins_len > max_ins) ins_len = max_ins;

  Molecule mol;
  mol.p = &p;
  mol.h = head_len;
  mol.t = tail_len;
  mol.ins_len = ins_len;
  mol.n_rand_tail = p.mask3 + p.umi3;
  mol.s_head = r.s;
  r.skip(p.umi5 + p.mask5);
  mol.s_tail = r.s;
  r.skip(p.mask3 + p.umi3);
  mol.s_ins = r.s;
  r.skip(ins_len);
  mol.poly_len = 0;
  mol.strand = 0;
  if (p.strand != 0 && r.unit() < p.poly_fraction) {
    int pl = 10 + (int)r.below(31);
    if (pl > ins_len) pl = ins_len;
    mol.poly_len = pl;
    mol.strand = p.strand;
  }
  const size_t st = p.stride;
  {
    const int art = r.unit() < p.art5_fraction ? p.n_p5_fw : 0;
    const int ad_lo = art + head_len + ins_len + tail_len, ad_hi = ad_lo + p.n_p7_fw;
    finish_read<1>(r, p, mol, art, ad_hi, ad_lo, ad_hi, seq1 + row * st, qual1 + row * st, len1 + row);
  }
  if (p.single_end) return;
  {
    const int art = r.unit() < p.art5_fraction ? p.n_p7_rc : 0;
    const int ad_lo = art + tail_len + ins_len + head_len, ad_hi = ad_lo + p.n_p5_rc;
    finish_read<2>(r, p, mol, art, ad_hi, ad_lo, ad_hi, seq2 + row * st, qual2 + row * st, len2 + row);
  }
}

thread_local char g_err[256] = "";

int copy_lit(uint8_t *dst, int32_t *n, const char *src) {
  const size_t k = src ? strlen(src) : 0;
  if (k >= (size_t)MAX_LIT) return -1;
  if (k) memcpy(dst, src, k);
  *n = (int32_t)k;
  return 0;
}

}  // namespace

extern "C" {

const char *csd_last_error(void) { return g_err; }

/* see include/cutseq_synth.h */
int csd_synth_pairs(const csh_synth_params *hp, uint64_t n, void *seq1, void *qual1, void *len1, void *seq2, void *qual2,
                    void *len2, void *stream) {
  g_err[0] = 0;
  if (!hp || !seq1 || !qual1 || !len1) return snprintf(g_err, sizeof g_err, "null argument"), -1;
  if (hp->read_len == 0 || hp->read_len > hp->stride || hp->read_len > 2000 || (hp->stride & 3))
    return snprintf(g_err, sizeof g_err, "read_len / stride out of range (stride must be a multiple of 4)"), -1;
  if (!hp->single_end && (!seq2 || !qual2 || !len2)) return snprintf(g_err, sizeof g_err, "paired batch without mate 2 arrays"), -1;
  DevSynth p;
  memset(&p, 0, sizeof p);
  p.read_len = hp->read_len, p.stride = hp->stride, p.seed = hp->seed, p.first_index = hp->first_index;
  if (copy_lit(p.p5_fw, &p.n_p5_fw, hp->p5_fw) || copy_lit(p.p7_fw, &p.n_p7_fw, hp->p7_fw) || copy_lit(p.p5_rc, &p.n_p5_rc, hp->p5_rc) ||
      copy_lit(p.p7_rc, &p.n_p7_rc, hp->p7_rc) || copy_lit(p.inline5, &p.n_i5, hp->inline5) || copy_lit(p.inline3, &p.n_i3, hp->inline3))
    return snprintf(g_err, sizeof g_err, "an adapter / inline barcode of %d bytes or more", MAX_LIT), -1;
  p.umi5 = hp->umi5, p.umi3 = hp->umi3, p.mask5 = hp->mask5, p.mask3 = hp->mask3, p.strand = hp->strand, p.single_end = hp->single_end;
  if (p.umi5 < 0 || p.umi3 < 0 || p.mask5 < 0 || p.mask3 < 0) return snprintf(g_err, sizeof g_err, "negative UMI / mask length"), -1;
  const int head_len = p.n_i5 + p.umi5 + p.mask5, tail_len = p.mask3 + p.umi3 + p.n_i3;
  /* the host routine truncates at MAX_TPL (and skips pairs with a head or tail beyond 1000 bases): shapes that get
     near those bounds are refused here instead of being reproduced */
  if (head_len > 1000 || tail_len > 1000 || MAX_LIT + head_len + (int)p.read_len + 40 + tail_len + MAX_LIT + 1 >= MAX_TPL)
    return snprintf(g_err, sizeof g_err, "molecule too long for the device generator"), -1;
  p.adapter_fraction = hp->adapter_fraction, p.partial_fraction = hp->partial_fraction, p.poly_fraction = hp->poly_fraction;
  p.art5_fraction = hp->art5_fraction, p.sub_rate = hp->sub_rate, p.indel_frac = hp->indel_frac, p.n_rate = hp->n_rate;
  if (n == 0) return 0;
  const uint64_t blocks = (n + 255) / 256;
  if (blocks > 0x7FFFFFFFull) return snprintf(g_err, sizeof g_err, "too many pairs for one launch"), -1;
  hipLaunchKernelGGL(synth_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, p, n, (uint8_t *)seq1, (uint8_t *)qual1,
                     (uint16_t *)len1, (uint8_t *)seq2, (uint8_t *)qual2, (uint16_t *)len2);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return snprintf(g_err, sizeof g_err, "synth_kernel launch: %s", hipGetErrorString(e)), -2;
  return 0;
}

int csd_abi_version(void) { return 1; }

}  // extern "C"
