// gpu_inflate_probe.hip -- feasibility probe, not product code: how fast does the device decode deflate chunks when every
// LANE decodes one chunk on its own (the marker scheme of csrc/pinflate.c makes chunks independent)?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/gpu_inflate_probe tools/micro/gpu_inflate_probe.hip
//   /tmp/gpu_inflate_probe file.gz [chunk_kb ...]
// The host finds the block boundaries (csh_deflate_find_block) and decodes everything once for reference; the kernel
// decodes chunk i from its boundary to the next one into 16-bit symbols; symbols are compared with the host's.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
#include "../../cutseq_amd/csrc/pinflate.c"
}

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);    \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

namespace dev {
constexpr int kFast = 10;
struct Tab {  // per lane, in global memory
  uint16_t fast[1 << kFast];
  uint16_t count[16];
  uint16_t symbol[288];
};
struct Bits {
  const uint8_t *in;
  size_t n, pos;
  uint64_t buf;
  int cnt;
};
__device__ inline void refill(Bits &b) {
  while (b.cnt <= 56) {
    b.buf |= (uint64_t)(b.pos < b.n ? b.in[b.pos] : 0) << b.cnt;
    b.pos++;
    b.cnt += 8;
  }
}
__device__ inline void refill8(Bits &b) {  // unaligned 8-byte load where possible
  if (b.pos + 8 <= b.n) {
    uint64_t w = 0;
    const uint8_t *p = b.in + b.pos;
    for (int i = 0; i < 8; ++i) w |= (uint64_t)p[i] << (8 * i);
    b.buf |= w << b.cnt;
    b.pos += (size_t)((63 - b.cnt) >> 3);
    b.cnt |= 56;
  } else {
    refill(b);
  }
}
__device__ inline uint32_t take(Bits &b, int k) {
  const uint32_t v = (uint32_t)(b.buf & ((1ull << k) - 1ull));
  b.buf >>= k;
  b.cnt -= k;
  return v;
}
__device__ int build(Tab &h, const uint8_t *len, int n) {
  uint16_t offs[16], next[16];
  for (int i = 0; i < 16; ++i) h.count[i] = 0;
  for (int s = 0; s < n; ++s) h.count[len[s]]++;
  for (int i = 0; i < (1 << kFast); ++i) h.fast[i] = 0;
  if (h.count[0] == n) return 1;
  int left = 1;
  for (int l = 1; l <= 15; ++l) {
    left <<= 1;
    left -= h.count[l];
    if (left < 0) return -1;
  }
  offs[1] = 0;
  for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
  for (int s = 0; s < n; ++s)
    if (len[s]) h.symbol[offs[len[s]]++] = (uint16_t)s;
  uint32_t code = 0;
  for (int l = 1; l <= 15; ++l) {
    next[l] = (uint16_t)code;
    code = (code + h.count[l]) << 1;
  }
  for (int s = 0; s < n; ++s) {
    const int l = len[s];
    if (!l) continue;
    const uint32_t c = next[l]++;
    if (l > kFast) continue;
    const uint32_t r = __brev(c) >> (32 - l);
    const uint16_t e = (uint16_t)((s << 4) | l);
    for (uint32_t i = r; i < (1u << kFast); i += 1u << l) h.fast[i] = e;
  }
  return left > 0 ? 1 : 0;
}
__device__ inline int decode(Bits &b, const Tab &h) {
  const uint16_t e = h.fast[b.buf & ((1u << kFast) - 1u)];
  if (e) {
    b.buf >>= (e & 15);
    b.cnt -= (e & 15);
    return e >> 4;
  }
  int code = 0, first = 0, index = 0;
  uint64_t v = b.buf;
  for (int l = 1; l <= 15; ++l) {
    code |= (int)(v & 1u);
    v >>= 1;
    const int count = h.count[l];
    if (code - count < first) {
      b.buf >>= l;
      b.cnt -= l;
      return h.symbol[index + (code - first)];
    }
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}
__constant__ uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t LEXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t DEXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// one lane = one chunk: from start[i] to start[i + 1] (bit positions; the last one runs to the final block)
__global__ void __launch_bounds__(64) inflate_chunks(const uint8_t *in, size_t n, const int64_t *start, int n_chunks, uint16_t *out,
                                                     size_t cap, Tab *tabs, int64_t *n_out, int64_t *end_bit, int *status) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n_chunks) return;
  Tab &lit = tabs[2 * i], &dist = tabs[2 * i + 1];
  uint16_t *o = out + (size_t)i * cap;
  const int64_t stop = i + 1 < n_chunks ? start[i + 1] : (int64_t)1 << 62;
  Bits b;
  b.in = in;
  b.n = n;
  b.pos = (size_t)(start[i] >> 3);
  b.buf = 0;
  b.cnt = 0;
  refill(b);
  b.buf >>= (start[i] & 7);
  b.cnt -= (int)(start[i] & 7);
  size_t at = 0;
  int rc = 0;
  for (;;) {
    refill(b);
    const int last = (int)take(b, 1);
    const uint32_t type = take(b, 2);
    if (type == 0) {
      take(b, b.cnt & 7);
      refill(b);
      const uint32_t len = take(b, 16), nlen = take(b, 16);
      if ((len ^ nlen) != 0xffffu) { rc = -1; break; }
      const size_t p = (size_t)((((uint64_t)b.pos << 3) - (uint64_t)b.cnt) >> 3);
      if (at + len > cap) { rc = -2; break; }
      if (p + len > n) { rc = -1; break; }
      for (uint32_t j = 0; j < len; ++j) o[at + j] = in[p + j];
      at += len;
      b.pos = p + len; b.buf = 0; b.cnt = 0;
    } else if (type == 2) {
      refill(b);
      const int hlit = (int)take(b, 5) + 257, hdist = (int)take(b, 5) + 1, hclen = (int)take(b, 4) + 4;
      uint8_t pre[19];
      for (int j = 0; j < 19; ++j) pre[j] = 0;
      for (int j = 0; j < hclen; ++j) {
        if (b.cnt < 3) refill(b);
        pre[ORDER[j]] = (uint8_t)take(b, 3);
      }
      if (build(lit, pre, 19) != 0) { rc = -1; break; }  // (the literal table doubles as the code-length code's)
      uint8_t len[316];
      int j = 0;
      bool bad = false;
      while (j < hlit + hdist) {
        refill(b);
        const int sym = decode(b, lit);
        if (sym < 0) { bad = true; break; }
        if (sym < 16) { len[j++] = (uint8_t)sym; continue; }
        int rep, val = 0;
        if (sym == 16) { if (j == 0) { bad = true; break; } val = len[j - 1]; rep = 3 + (int)take(b, 2); }
        else if (sym == 17) rep = 3 + (int)take(b, 3);
        else rep = 11 + (int)take(b, 7);
        if (j + rep > hlit + hdist) { bad = true; break; }
        while (rep--) len[j++] = (uint8_t)val;
      }
      if (bad) { rc = -1; break; }
      if (build(lit, len, hlit) < 0 || build(dist, len + hlit, hdist) < 0) { rc = -1; break; }
      // the block's symbols
      for (;;) {
        refill8(b);
        int sym = decode(b, lit);
        if (sym < 0) { rc = -1; break; }
        if (sym < 256) {
          if (at >= cap) { rc = -2; break; }
          o[at++] = (uint16_t)sym;
          continue;
        }
        if (sym == 256) break;
        sym -= 257;
        if (sym >= 29) { rc = -1; break; }
        const int length = LBASE[sym] + (int)take(b, LEXTRA[sym]);
        const int ds = decode(b, dist);
        if (ds < 0 || ds >= 30) { rc = -1; break; }
        const int d = DBASE[ds] + (int)take(b, DEXTRA[ds]);
        if (at + (size_t)length > cap) { rc = -2; break; }
        for (int q = 0; q < length; ++q) {
          const long long from = (long long)at + q - d;
          o[at + q] = from < 0 ? (uint16_t)(0x8000u | (uint32_t)(32768 + from)) : o[from];
        }
        at += (size_t)length;
      }
      if (rc) break;
    } else {
      rc = -3;  // fixed-code blocks: not in this probe
      break;
    }
    if (last || (int64_t)(((uint64_t)b.pos << 3) - (uint64_t)b.cnt) >= stop) break;
  }
  n_out[i] = (int64_t)at;
  end_bit[i] = (int64_t)(((uint64_t)b.pos << 3) - (uint64_t)b.cnt);
  status[i] = rc;
}
}  // namespace dev

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  FILE *f = fopen(argv[1], "rb");
  fseek(f, 0, SEEK_END);
  const size_t size = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> file(size);
  if (fread(file.data(), 1, size, f) != size) return 1;
  fclose(f);
  // plain 10-byte header assumed (gzip -1 < file)
  size_t hdr = 10;
  if (file[3] & 8) while (file[hdr++]) {}
  const uint8_t *in = file.data() + hdr;
  const size_t n = size - hdr;
  uint8_t *d_in;
  CK(hipMalloc(&d_in, n + 64));
  CK(hipMemcpy(d_in, in, n, hipMemcpyHostToDevice));
  for (int a = 2; a < argc || a == 2; ++a) {
    const size_t chunk = (size_t)(a < argc ? atoi(argv[a]) : 64) << 10;
    std::vector<int64_t> start{0};
    auto t0 = std::chrono::steady_clock::now();
    for (size_t c = chunk; c < n; c += chunk) {
      const int64_t p = csh_deflate_find_block(in, (int64_t)n, (int64_t)c * 8, (int64_t)n * 8);
      if (p < 0) break;
      if (p > start.back()) start.push_back(p);
    }
    const double find_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const int nc = (int)start.size();
    const size_t cap = chunk * 10;  // symbols per chunk
    uint16_t *d_out;
    dev::Tab *d_tabs;
    int64_t *d_start, *d_nout, *d_end;
    int *d_status;
    CK(hipMalloc(&d_out, (size_t)nc * cap * 2));
    CK(hipMalloc(&d_tabs, (size_t)nc * 2 * sizeof(dev::Tab)));
    CK(hipMalloc(&d_start, nc * 8));
    CK(hipMalloc(&d_nout, nc * 8));
    CK(hipMalloc(&d_end, nc * 8));
    CK(hipMalloc(&d_status, nc * 4));
    CK(hipMemcpy(d_start, start.data(), nc * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(dev::inflate_chunks, dim3((nc + 63) / 64), dim3(64), 0, 0, d_in, n, d_start, nc, d_out, cap, d_tabs, d_nout, d_end, d_status);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    std::vector<int64_t> nout(nc), endb(nc);
    std::vector<int> status(nc);
    CK(hipMemcpy(nout.data(), d_nout, nc * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(endb.data(), d_end, nc * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(status.data(), d_status, nc * 4, hipMemcpyDeviceToHost));
    int64_t total = 0;
    int bad = 0, unlinked = 0;
    for (int i = 0; i < nc; ++i) {
      total += nout[i];
      bad += status[i] != 0;
      if (i + 1 < nc && endb[i] != start[i + 1]) ++unlinked;
    }
    // spot check of a chunk in the middle against the host decoder
    int mismatch = -1;
    {
      const int i = nc / 2;
      std::vector<uint16_t> ref(cap), got((size_t)nout[i]);
      int64_t eb, no;
      int32_t fin;
      csh_inflate_chunk(in, (int64_t)n, start[i], i + 1 < nc ? start[i + 1] : (int64_t)1 << 62, ref.data(), (int64_t)cap, &eb, &no, &fin);
      CK(hipMemcpy(got.data(), d_out + (size_t)i * cap, (size_t)nout[i] * 2, hipMemcpyDeviceToHost));
      mismatch = (no == nout[i] && memcmp(ref.data(), got.data(), (size_t)no * 2) == 0) ? 0 : 1;
    }
    printf("chunk %4zu KB: %6d chunks, host find %.3f s; device decode %8.3f ms = %6.2f GB/s of text (%lld bytes); bad %d, unlinked %d, spot check %s\n",
           chunk >> 10, nc, find_s, best, total / (best * 1e6), (long long)total, bad, unlinked, mismatch == 0 ? "ok" : "MISMATCH");
    hipFree(d_out); hipFree(d_tabs); hipFree(d_start); hipFree(d_nout); hipFree(d_end); hipFree(d_status);
  }
  return 0;
}
