// gpu_inflate_probe2.hip -- feasibility probe, not product code: can the device inflate ONE gzip
// member at tens of GB/s of text?  Round 3's probe (a lane per chunk, tables and output in global memory, back-references
// copied while decoding) managed ~1 MB/s per lane.  This one splits the work the way the data allows:
//
//   A  token decode     one WAVE per chunk of compressed bytes (the host's finder names the chunk starts, as in
//                       csrc/pinflate.c); lane 0 walks the Huffman codes with its tables in LDS and writes TOKENS --
//                       a literal byte or (length, distance) -- nothing is copied, nothing read back: the only
//                       dependent chain is bit buffer -> table look-up -> bit buffer
//   B  expansion        token lengths -> exclusive scan -> every token's place in the member's text (absolute: no
//                       windows, no markers); literals are written, every byte of a match gets a pointer to the byte it
//                       copies; pointer jumping (src[p] = src[src[p]], log2 of the longest copy chain rounds) resolves
//                       all of them in parallel -- chains across blocks and chunks included
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/gpu_inflate_probe2 tools/micro/gpu_inflate_probe2.hip -lz
//   /tmp/gpu_inflate_probe2 file.gz [chunk_kb ...]
// Checks the result against zlib's inflate of the same file.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
#include "../../cutseq_amd/csrc/pinflate.c"
}

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

namespace dev {
constexpr int kFast = 10;
struct Tab {  // per wave, in LDS
  uint16_t fast[1 << kFast];
  uint16_t count[16];
  uint16_t symbol[288];
};
constexpr uint32_t kRing = 4096;  // bytes of compressed input per wave in LDS, refilled in halves by all 64 lanes
struct Bits {
  const uint32_t *ring;  // LDS, kRing / 4 dwords: byte x of the input sits at ring byte x % kRing
  size_t pos;            // next byte to load
  uint64_t buf;
  int cnt;
};
__device__ inline uint64_t load8(const uint32_t *ring, size_t pos) {
  const uint32_t d = (uint32_t)(pos >> 2) & (kRing / 4 - 1);
  const uint32_t sh = (uint32_t)(pos & 3u) * 8u;
  // (every lane of the wave runs the decoder on the same values: what comes out of LDS is handed to the scalar unit)
  const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ring[d]);
  const uint32_t w1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ring[(d + 1) & (kRing / 4 - 1)]);
  const uint32_t w2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ring[(d + 2) & (kRing / 4 - 1)]);
  const uint64_t lo = (uint64_t)w0 | ((uint64_t)w1 << 32);
  const uint64_t hi = w2;
  return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
}
__device__ inline void refill(Bits &b) {  // at least 56 bits afterwards
  b.buf |= load8(b.ring, b.pos) << b.cnt;
  b.pos += (size_t)((63 - b.cnt) >> 3);
  b.cnt |= 56;
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
  const uint16_t e = (uint16_t)__builtin_amdgcn_readfirstlane((int)h.fast[b.buf & ((1u << kFast) - 1u)]);
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
    const int count = __builtin_amdgcn_readfirstlane((int)h.count[l]);
    if (code - count < first) {
      b.buf >>= l;
      b.cnt -= l;
      return __builtin_amdgcn_readfirstlane((int)h.symbol[index + (code - first)]);
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

// A: one wave per chunk.  Lane 0 walks the codes; all 64 lanes keep a 4 KB ring of the compressed bytes filled in LDS (a
// global load per token, its latency in the way of the next look-up, was what held the first version at 5 MB/s per wave).
// token: literal = byte; match = 0x80000000 | length << 16 | (distance - 1)
__global__ void __launch_bounds__(64) decode_tokens(const uint8_t *in, size_t n, const int64_t *start, int n_chunks, uint32_t *tok,
                                                    size_t cap, int64_t *n_tok, int64_t *n_text, int64_t *end_bit, int *status) {
  __shared__ Tab lit, dist;
  __shared__ uint32_t ring[kRing / 4];
  __shared__ int s_state;  // 0: go on, 1: done
  const int i = blockIdx.x, lane = threadIdx.x;
  uint32_t *o = tok + (size_t)i * cap;
  const int64_t stop = i + 1 < n_chunks ? start[i + 1] : (int64_t)1 << 62;
  const uint32_t *in4 = reinterpret_cast<const uint32_t *>(in);  // (the allocation is padded)
  size_t rbase = (size_t)(start[i] >> 3) & ~(size_t)(kRing / 2 - 1);  // the ring holds [rbase, rbase + kRing)
  auto load_half = [&](size_t from) {  // kRing / 2 bytes at `from` (a multiple of kRing / 2) into their ring slots
    const uint32_t slot = (uint32_t)(from & (kRing - 1)) / 4u;
#pragma unroll
    for (uint32_t q = 0; q < kRing / 8 / 64; ++q) {
      const size_t d = from / 4 + q * 64 + lane;
      ring[slot + q * 64 + lane] = d * 4 < n + 60 ? in4[d] : 0u;
    }
  };
  load_half(rbase);
  load_half(rbase + kRing / 2);
  if (lane == 0) s_state = 0;
  __syncthreads();
  Bits b;
  b.ring = ring;
  b.pos = (size_t)(start[i] >> 3);
  b.buf = 0;
  b.cnt = 0;
  size_t at = 0;
  int64_t text = 0;
  int rc = 0;
  // lane 0's place in the stream: 0 = at a block header, 1 = inside a dynamic block's symbols
  int phase = 0, last = 0;
  {
    refill(b);
    b.buf >>= (start[i] & 7);
    b.cnt -= (int)(start[i] & 7);
  }
  for (;;) {
    {
      // decode while the bytes a refill may touch are in the ring: pos + 16 <= rbase + kRing, and hand over to the
      // refill once pos has left the lower half
      const size_t limit = rbase + kRing - 16, handover = rbase + kRing / 2;
      bool done = false;
      while (!done && b.pos <= limit && b.pos < handover + kRing / 4) {
        if (phase == 0) {
          refill(b);
          last = (int)take(b, 1);
          const uint32_t type = take(b, 2);
          if (type == 2) {
            refill(b);
            const int hlit = (int)take(b, 5) + 257, hdist = (int)take(b, 5) + 1, hclen = (int)take(b, 4) + 4;
            uint8_t pre[19];
            for (int j = 0; j < 19; ++j) pre[j] = 0;
            for (int j = 0; j < hclen; ++j) {
              if (b.cnt < 3) refill(b);
              pre[ORDER[j]] = (uint8_t)take(b, 3);
            }
            int brc = 0;
            if (lane == 0) brc = build(lit, pre, 19);
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(brc) != 0) { rc = -1; done = true; break; }
            uint8_t len[316];
            int j = 0;
            bool bad = false;
            while (j < hlit + hdist) {  // (a header is at most ~300 bytes: inside the margin the hand-over leaves)
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
            __syncthreads();
            if (lane == 0) brc = (build(lit, len, hlit) < 0 || build(dist, len + hlit, hdist) < 0) ? -1 : 0;
            __syncthreads();
            if (bad || __builtin_amdgcn_readfirstlane(brc) < 0) { rc = -1; done = true; break; }
            phase = 1;
          } else {
            rc = -3;  // stored / fixed-code blocks: not in this probe
            done = true;
            break;
          }
        } else {
          refill(b);
          int sym = decode(b, lit);
          if (sym < 0) { rc = -1; done = true; break; }
          if (sym < 256) {
            if (at >= cap) { rc = -2; done = true; break; }
            if (lane == 0) o[at] = (uint32_t)sym;
            ++at;
            ++text;
            continue;
          }
          if (sym == 256) {
            phase = 0;
            if (last || (int64_t)(((uint64_t)b.pos << 3) - (uint64_t)b.cnt) >= stop) done = true;
            continue;
          }
          sym -= 257;
          if (sym >= 29) { rc = -1; done = true; break; }
          const int length = LBASE[sym] + (int)take(b, LEXTRA[sym]);
          const int ds = decode(b, dist);
          if (ds < 0 || ds >= 30) { rc = -1; done = true; break; }
          const int d = DBASE[ds] + (int)take(b, DEXTRA[ds]);
          if (at >= cap) { rc = -2; done = true; break; }
          if (lane == 0) o[at] = 0x80000000u | ((uint32_t)length << 16) | (uint32_t)(d - 1);
          ++at;
          text += length;
        }
      }
      if (lane == 0) s_state = done ? 1 : 0;
    }
    __syncthreads();
    if (s_state) break;
    // lane 0 left the lower half: the next half of the input takes its place
    load_half(rbase + kRing);
    rbase += kRing / 2;
    __syncthreads();
  }
  if (lane == 0) {
    n_tok[i] = (int64_t)at;
    n_text[i] = text;
    end_bit[i] = (int64_t)(((uint64_t)b.pos << 3) - (uint64_t)b.cnt);
    status[i] = rc;
  }
}

// B1: tokens of all chunks into one array (chunk regions are sparse), with their text lengths beside them
__global__ void compact_tokens(const uint32_t *tok, size_t cap, const int64_t *n_tok, const int64_t *tok_base, int n_chunks,
                               uint32_t *all, uint32_t *len) {
  const int c = blockIdx.y;
  const int64_t n = n_tok[c], base = tok_base[c];
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t t = tok[(size_t)c * cap + k];
    all[base + k] = t;
    len[base + k] = (t & 0x80000000u) ? ((t >> 16) & 0x7fffu) : 1u;
  }
}
// B2: literals land, every byte of a match points at the byte it copies (kSelf: a literal / resolved)
constexpr uint32_t kSelf = 0xffffffffu;
__global__ void scatter_tokens(const uint32_t *all, const uint32_t *pos, int64_t n_tokens, uint8_t *out, uint32_t *src) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_tokens; k += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t t = all[k], p = pos[k];
    if (!(t & 0x80000000u)) {
      out[p] = (uint8_t)t;
      src[p] = kSelf;
    } else {
      const uint32_t L = (t >> 16) & 0x7fffu, d = (t & 0xffffu) + 1u;
      for (uint32_t q = 0; q < L; ++q) src[p + q] = p + q - d;
    }
  }
}
// B3: one round of pointer jumping, src_in -> src_out (two arrays: no races to reason about in a probe)
__global__ void jump(const uint32_t *src_in, uint32_t *src_out, uint8_t *out, uint32_t n, unsigned long long *open) {
  uint32_t mine = 0;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const uint32_t s = src_in[p];
    uint32_t next = s;
    if (s != kSelf) {
      const uint32_t ss = src_in[s];
      if (ss == kSelf) {
        out[p] = out[s];  // (out[s] was final before this launch)
        next = kSelf;
      } else {
        next = ss;
        ++mine;
      }
    }
    src_out[p] = next;
  }
  if (mine) atomicAdd(open, (unsigned long long)mine);
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
  // reference text: zlib
  std::vector<uint8_t> ref;
  {
    z_stream z;
    memset(&z, 0, sizeof z);
    inflateInit2(&z, 31);
    ref.resize(size * 8);
    z.next_in = file.data();
    z.avail_in = (uInt)size;
    z.next_out = ref.data();
    z.avail_out = (uInt)ref.size();
    const int rc = inflate(&z, Z_FINISH);
    if (rc != Z_STREAM_END) { fprintf(stderr, "zlib: %d\n", rc); return 1; }
    ref.resize(z.total_out);
    inflateEnd(&z);
  }
  size_t hdr = 10;
  if (file[3] & 8) while (file[hdr++]) {}
  const uint8_t *in = file.data() + hdr;
  const size_t n = size - hdr;
  uint8_t *d_in;
  CK(hipMalloc(&d_in, n + 64));
  CK(hipMemset(d_in, 0, n + 64));
  CK(hipMemcpy(d_in, in, n, hipMemcpyHostToDevice));
  const uint32_t n_text = (uint32_t)ref.size();
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
    const size_t cap = chunk * 4 + 65536;  // tokens per chunk (a token is at least ~9 bits of input, a block may overrun the chunk)
    uint32_t *d_tok, *d_all, *d_len, *d_pos, *d_src[2];
    uint8_t *d_out;
    int64_t *d_start, *d_ntok, *d_ntext, *d_end, *d_base;
    int *d_status;
    unsigned long long *d_open;
    CK(hipMalloc(&d_tok, (size_t)nc * cap * 4));
    CK(hipMalloc(&d_start, nc * 8));
    CK(hipMalloc(&d_ntok, nc * 8));
    CK(hipMalloc(&d_ntext, nc * 8));
    CK(hipMalloc(&d_end, nc * 8));
    CK(hipMalloc(&d_base, nc * 8));
    CK(hipMalloc(&d_status, nc * 4));
    CK(hipMalloc(&d_open, 8));
    CK(hipMemcpy(d_start, start.data(), nc * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best_a = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(dev::decode_tokens, dim3(nc), dim3(64), 0, 0, d_in, n, d_start, nc, d_tok, cap, d_ntok, d_ntext, d_end, d_status);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best_a) best_a = ms;
    }
    std::vector<int64_t> ntok(nc), ntext(nc), endb(nc), base(nc);
    std::vector<int> status(nc);
    CK(hipMemcpy(ntok.data(), d_ntok, nc * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ntext.data(), d_ntext, nc * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(endb.data(), d_end, nc * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(status.data(), d_status, nc * 4, hipMemcpyDeviceToHost));
    int64_t tokens = 0, text = 0;
    int bad = 0, unlinked = 0;
    for (int i = 0; i < nc; ++i) {
      base[i] = tokens;
      tokens += ntok[i];
      text += ntext[i];
      bad += status[i] != 0;
      if (i + 1 < nc && endb[i] != start[i + 1]) ++unlinked;
    }
    printf("chunk %4zu KB: %6d chunks (host find %.3f s); A token decode %8.3f ms = %6.2f GB/s of text; %lld tokens for %lld bytes; bad %d, unlinked %d\n",
           chunk >> 10, nc, find_s, best_a, text / (best_a * 1e6), (long long)tokens, (long long)text, bad, unlinked);
    if (bad || unlinked || (uint64_t)text != n_text) {
      printf("   (decode incomplete: expansion skipped)\n");
      continue;
    }
    CK(hipMemcpy(d_base, base.data(), nc * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_all, (size_t)tokens * 4));
    CK(hipMalloc(&d_len, (size_t)tokens * 4));
    CK(hipMalloc(&d_pos, (size_t)tokens * 4));
    CK(hipMalloc(&d_src[0], (size_t)n_text * 4 + 1024));
    CK(hipMalloc(&d_src[1], (size_t)n_text * 4 + 1024));
    CK(hipMalloc(&d_out, (size_t)n_text + 1024));
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_len, d_pos, (int)tokens);
    CK(hipMalloc(&d_tmp, tmp_bytes));
    float best_b = 1e9f;
    int rounds = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(dev::compact_tokens, dim3(64, nc), dim3(256), 0, 0, d_tok, cap, d_ntok, d_base, nc, d_all, d_len);
      hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_len, d_pos, (int)tokens);
      hipLaunchKernelGGL(dev::scatter_tokens, dim3(4096), dim3(256), 0, 0, d_all, d_pos, tokens, d_out, d_src[0]);
      rounds = 0;
      for (;;) {
        CK(hipMemsetAsync(d_open, 0, 8, 0));
        hipLaunchKernelGGL(dev::jump, dim3(8192), dim3(256), 0, 0, d_src[rounds & 1], d_src[(rounds + 1) & 1], d_out, n_text, d_open);
        ++rounds;
        unsigned long long open = 0;
        CK(hipMemcpy(&open, d_open, 8, hipMemcpyDeviceToHost));
        if (!open || rounds > 64) break;
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best_b) best_b = ms;
    }
    std::vector<uint8_t> got(n_text);
    CK(hipMemcpy(got.data(), d_out, n_text, hipMemcpyDeviceToHost));
    const bool same = memcmp(got.data(), ref.data(), n_text) == 0;
    printf("              B expansion (compact, scan, scatter, %d pointer-jumping rounds incl. host round trips) %8.3f ms = %6.2f GB/s; A + B %8.3f ms = %6.2f GB/s of text; text %s zlib's\n",
           rounds, best_b, n_text / (best_b * 1e6), best_a + best_b, n_text / ((best_a + best_b) * 1e6), same ? "==" : "!=");
    hipFree(d_all); hipFree(d_len); hipFree(d_pos); hipFree(d_src[0]); hipFree(d_src[1]); hipFree(d_out); hipFree(d_tmp);
    hipFree(d_tok); hipFree(d_start); hipFree(d_ntok); hipFree(d_ntext); hipFree(d_end); hipFree(d_base); hipFree(d_status); hipFree(d_open);
  }
  return 0;
}
