// cutseq_hip.hip -- C ABI (include/cutseq_hip.h) + launch code of the trimming kernels (scan + resolve).
// Build: hipcc -O3 --offload-arch=gfx950 -fPIC -shared -o libcutseq_hip.so cutseq_hip.hip
// gfx950 (MI355X) only; there is no CPU path in this library.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <map>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "../../include/cutseq_hip.h"
#include "trim_kernel.hip.inc"
#include "long_kernel.hip.inc"
#include "text_kernels.hip.inc"
#include "deflate_kernels.hip.inc"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return fail(CS_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                      __FILE__, __LINE__);                                         \
  } while (0)

// tile hand-out counters of the scan kernel [2][kTileCounters], then the queue counters (records written
// per mate, batches claimed per mate), every counter on a 128-byte line of its own
constexpr size_t kTileCounterDwords = 2 * csdev::kTileCounters * csdev::kTileCounterStride;
constexpr size_t kTileCounterBytes = (kTileCounterDwords + 6 * csdev::kTileCounterStride) * sizeof(uint32_t);
constexpr size_t kDeferRecordBytes = 32;
constexpr uint32_t kTileRows = 64;  // one wave per block: no block-level synchronisation at all
constexpr uint32_t kLdsBudget = 160 * 1024;

struct Slot {
  uint8_t *d_seq[2] = {nullptr, nullptr};
  uint8_t *d_qual[2] = {nullptr, nullptr};
  uint16_t *d_len[2] = {nullptr, nullptr};
  cs_result *d_out[2] = {nullptr, nullptr};
  cs_cap2 *d_cap2 = nullptr;
  uint8_t *d_bc[2] = {nullptr, nullptr};
  hipEvent_t done = nullptr;
  bool busy = false;
};

// Everything one call keeps in flight: tile hand-out / queue counters, the queue of deferred reads and the
// events around the two kernels.  Calls take the lanes in turn; a call first makes its stream wait for the
// lane's previous user, so calls on different streams (and the pipelined form) never share live state.
struct Lane {
  uint32_t *d_counters = nullptr;
  void *d_defer[2] = {nullptr, nullptr};  // per mate: queue of deferred reads (kDeferRecordBytes each)
  uint32_t defer_capacity = 0;            // records per mate
  hipEvent_t scanned = nullptr;           // scan kernel finished (what the resolve stream waits for)
  hipEvent_t done = nullptr;              // resolve kernel finished
  hipEvent_t ev_start = nullptr, ev_mid = nullptr, ev_stop = nullptr;  // timing: scan = start..mid, resolve = mid..stop
  bool used = false, timed = false;
  bool unharvested = false;  // the timing events of the lane's last call have not been added to the totals yet
};
constexpr int kLanes = 3;

}  // namespace

// CS_OP_DEMUX with m + k > CS_DEMUX_MAX_PREFIX (cs_plan_set_demux_ops): the barcodes' own PrefixAdapter ops, and a
// look-up table over the first `depth` bases of the interval that names the barcodes which can still match
struct DemuxLong {
  std::vector<csdev::DevOp> ops;
  std::vector<uint32_t> first;  // per prefix (layout of cs_plan_set_demux): offset into `pool` << 8 | candidates
  std::vector<uint8_t> pool;    // candidate lists, barcode indices in ascending order
  int depth = 0;
};

struct cs_plan {
  csdev::DevPlan host;
  std::vector<uint16_t> demux[2][CS_MAX_OPS];  // look-up tables of the CS_OP_DEMUX ops
  DemuxLong demux_long[2][CS_MAX_OPS];
};

#include <mutex>
namespace {
std::mutex g_slot_mutex;
bool g_slot_used[64][csdev::kMaxPlanSlots];  // [device][slot]

int acquire_plan_slot(int device) {
  if (device < 0 || device >= 64) return -1;
  std::lock_guard<std::mutex> lock(g_slot_mutex);
  for (int i = 0; i < csdev::kMaxPlanSlots; ++i)
    if (!g_slot_used[device][i]) {
      g_slot_used[device][i] = true;
      return i;
    }
  return -1;
}
void release_plan_slot(int device, int slot) {
  if (device < 0 || device >= 64 || slot < 0 || slot >= csdev::kMaxPlanSlots) return;
  std::lock_guard<std::mutex> lock(g_slot_mutex);
  g_slot_used[device][slot] = false;
}
}  // namespace

struct cs_engine {
  int device = -1;
  hipStream_t stream = nullptr;
  int plan_slot = -1;  // index into the device's __constant__ plan table
  hipStream_t resolve_stream = nullptr;  // pipelined calls: resolve kernel (and the slot's D2H) run here
  unsigned long long *d_stats = nullptr;
  Lane lanes[kLanes];
  int next_lane = 0;
  Lane *last_lane = nullptr;
  uint32_t timed_calls = 0;           // since the last reset of the totals
  double kernel_ms_total[2] = {0, 0};  // scan / resolve kernel, HIP-event durations of those calls
  uint32_t max_reads = 0, max_stride = 0;
  bool paired = false;
  bool coded = false;
  bool wide = false;  // some adapter needs 64-bit bit-vectors
  int n_cus = 256;
  uint32_t n_table_ops = 1;
  uint32_t col_dwords = 0;   // per-wave DP scratch the plan needs (resolve kernel)
  uint32_t waves_per_simd[2] = {4, 4};  // scan / resolve kernel, from their register counts
  bool long_demux = false;  // a CS_OP_DEMUX op with the barcodes' own ops (cs_plan_set_demux_ops)
  int demux_mate = -1;      // 0 / 1: the mate whose chain holds a CS_OP_DEMUX op
  uint32_t demux_bins = 0;  // ... and how many barcodes its table names
  std::vector<Slot> slots;
  uint32_t max_dynamic_lds[2] = {0, 0};
  std::vector<void *> d_tables;           // device copies of the CS_OP_DEMUX tables
  // tuning knobs, read once from the environment when the engine is created
  uint32_t knob_col_bytes = 0, knob_grid_x = 0, knob_batch = 1;  // batch knob: see trim_kernel (resolve)
  uint32_t knob_resolve_waves = 4;  // pipelined calls: resolve waves per CU
  uint32_t knob_item_slots = 0;     // CUTSEQ_ITEM_SLOTS: items per lane in the scan kernel's item log (0: what fits)
  uint32_t knob_fast_recode = 1;    // CUTSEQ_FAST_RECODE=0: the scan kernel re-codes every tile with the exact form (2: tests, see KArgs)
  bool knob_units = false;
  uint32_t knob_big_shift = 0, knob_small_shift = 0, knob_big_pct = 50;
};

namespace {

bool is_acgt(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// DevOp::f: the op's scalar fields as dwords, for the device (trim_kernel.hip.inc, DevFields).  Called on the copies
// an engine uploads, when nothing changes the ops any more.
void sync_fields(csdev::DevOp &d) {
  const cs_op &o = d.op;
  csdev::DevFields &f = d.f;
  f.kind = o.kind, f.align_flags = o.align_flags, f.reversed = o.reversed, f.remove = o.remove, f.shortcut = o.shortcut;
  f.match_flag = o.match_flag, f.required = o.required, f.conditional = o.conditional, f.capture = o.capture;
  f.homopolymer = o.homopolymer, f.q_base = o.q_base, f.stat_slot = o.stat_slot, f.m = o.m, f.k = o.k;
  f.min_overlap = o.min_overlap, f.force_min_len = o.force_min_len, f.cut_len = o.cut_len, f.q_cutoff = o.q_cutoff;
}

int build_dev_op(const cs_op &in, csdev::DevOp &out, int index, int mate) {
  memset(&out, 0, sizeof out);
  out.op = in;
  cs_op &op = out.op;
  switch (op.kind) {
    case CS_OP_ADAPTER: {
      if (op.m < 1 || op.m > CS_MAX_ADAPTER) return fail(CS_ERR_ARG, "mate %d op %d: adapter length %u", mate, index, op.m);
      if (op.min_overlap < 1 || op.min_overlap > op.m)
        return fail(CS_ERR_ARG, "mate %d op %d: min_overlap %u not in [1, m]", mate, index, op.min_overlap);
      if (op.align_flags > 15) return fail(CS_ERR_ARG, "mate %d op %d: bad align_flags", mate, index);
      if (op.k > op.m || op.thr[op.m] != op.k)
        return fail(CS_ERR_ARG, "mate %d op %d: k=%u inconsistent with thr[m]=%u", mate, index, op.k, op.thr[op.m]);
      for (int i = 1; i <= op.m; ++i)
        if (op.thr[i] < op.thr[i - 1] || op.thr[i] > op.thr[i - 1] + 1)
          return fail(CS_ERR_ARG, "mate %d op %d: thr[] must be a non-decreasing unit-step table", mate, index);
      if (op.remove > 1 || op.shortcut > 1) return fail(CS_ERR_ARG, "mate %d op %d: bad remove/shortcut", mate, index);
      bool homo = true, acgt = true;
      for (int i = 0; i < op.m; ++i) {
        homo = homo && op.seq[i] == op.seq[0];
        acgt = acgt && is_acgt(op.seq[i]);
      }
      op.homopolymer = homo ? 1 : 0;
      out.base_char = op.seq[0];
      static const char kBase[4] = {'A', 'C', 'G', 'T'};
      for (int b = 0; b < 4; ++b) {
        uint64_t mask = 0;
        for (int i = 0; i < op.m && i < 64; ++i)
          if (op.seq[i] == (uint8_t)kBase[b]) mask |= 1ull << i;
        out.peq[b] = mask;
      }
      out.acgt_only = (acgt && op.m <= 64) ? 1u : 0u;
      // fixed-point error rate: thr[L] == (L * mul) >> 16 for all L <= m, if such a mul exists
      out.thr_mul = 0;
      if (op.m >= 1 && op.thr[op.m] > 0) {
        const uint32_t guess = (uint32_t)(((uint64_t)op.thr[op.m] << 16) / op.m);
        for (uint32_t mul = guess; mul <= guess + 4096 && !out.thr_mul; ++mul) {
          bool ok = true;
          for (uint32_t L = 0; L <= op.m && ok; ++L) ok = ((L * mul) >> 16) == op.thr[L];
          if (ok) out.thr_mul = mul;
        }
      }
      for (int i = 1; i <= op.m; ++i)
        if (op.thr[i] != op.thr[i - 1]) out.thr_step[(i - 1) >> 6] |= 1ull << ((i - 1) & 63);
      out.exists_only = 0;
      out.filter_mode = csdev::FILTER_NONE;
      if (homo && !op.reversed && op.align_flags == CS_WHERE_BACK_NOT_INTERNAL && op.shortcut == CS_SHORTCUT_NONE)
        out.filter_mode = csdev::FILTER_POLY_TAIL;
      else if (homo && !op.reversed && op.align_flags == CS_WHERE_FRONT_NOT_INTERNAL &&
               op.shortcut == CS_SHORTCUT_NONE)
        out.filter_mode = csdev::FILTER_POLY_HEAD;
      else if (acgt && op.m <= 32) {
        out.filter_mode = csdev::FILTER_MYERS32;
        // RightmostFrontAdapter (the 5' template-switch artefact) hardly ever matches: existence-only scan
        // (trim_kernel.hip.inc, myers_none).  CUTSEQ_EXISTS=0 keeps the exact filter for every op.
        const char *env = getenv("CUTSEQ_EXISTS");
        if (op.reversed && op.align_flags == CS_WHERE_BACK && op.shortcut == CS_SHORTCUT_NONE && !(env && atoi(env) == 0))
          out.exists_only = 1;
      }
      else if (acgt && op.m <= 64 && op.m + op.k <= 127)  // (scores of a column group travel as bytes below 128)
        out.filter_mode = csdev::FILTER_MYERS64;
      break;
    }
    case CS_OP_CUT:
      if (op.capture > 2) return fail(CS_ERR_ARG, "mate %d op %d: capture slot %u", mate, index, op.capture);
      if (op.capture && (op.cut_len > 255 || op.cut_len < -255))
        return fail(CS_ERR_ARG, "mate %d op %d: capturing cuts are limited to 255 bases", mate, index);
      break;
    case CS_OP_QTRIM:
      break;
    case CS_OP_DEMUX:
      if (op.m < 1 || op.k > op.m || op.m + op.k > CS_DEMUX_MAX_LONG)
        return fail(CS_ERR_ARG, "mate %d op %d: demux barcode length %u with %u errors (m + k <= %d)", mate, index, op.m,
                    op.k, CS_DEMUX_MAX_LONG);
      // filter_mode of a demultiplexing op: 0 = table look-up, 1 = the barcodes' own ops (cs_plan_set_demux_ops)
      // (barcodes at the 3' end -- cs_op.reversed, SuffixAdapter ops -- exist in the second form only)
      out.filter_mode = (op.m + op.k > CS_DEMUX_MAX_PREFIX || op.shortcut == CS_DEMUX_BY_OPS || op.reversed) ? 1u : 0u;
      break;
    default:
      return fail(CS_ERR_ARG, "mate %d op %d: unknown op kind %u", mate, index, op.kind);
  }
  if (op.stat_slot >= CS_MAX_OPS) return fail(CS_ERR_ARG, "mate %d op %d: stat_slot out of range", mate, index);
  return CS_OK;
}

// launch geometry for a given row stride
struct Geometry {
  uint32_t tile_rows, lds_stride_dw, col_dwords, lds_bytes, item_slots;
};

template <int MODE>
const void *kernel_of(const cs_engine *eng) {
  if (eng->coded)
    return eng->wide ? reinterpret_cast<const void *>(csdev::trim_kernel<true, true, MODE>)
                     : reinterpret_cast<const void *>(csdev::trim_kernel<true, false, MODE>);
  return eng->wide ? reinterpret_cast<const void *>(csdev::trim_kernel<false, true, MODE>)
                   : reinterpret_cast<const void *>(csdev::trim_kernel<false, false, MODE>);
}
const void *kernel_for(const cs_engine *eng, int mode) {
  if (mode == csdev::MODE_RESOLVE) return kernel_of<csdev::MODE_RESOLVE>(eng);
  return kernel_of<csdev::MODE_SCAN>(eng);
}

int geometry_for(const cs_engine *eng, uint32_t stride, int mode, Geometry &g) {
  if (stride == 0 || stride % 4 || stride > CS_MAX_STRIDE)
    return fail(CS_ERR_ARG, "stride %u must be a multiple of 4 in [4, %d]", stride, CS_MAX_STRIDE);
  // coded plans keep 4 bits per base (8 per dword), raw plans the ASCII bytes; odd dword
  // stride: conflict-free column walks
  g.lds_stride_dw = (eng->coded ? (stride + 7) / 8 : stride / 4) | 1u;
  // one wave per block: the kernel keeps per-block state (private mask table) on that basis.  Only the
  // resolve kernel runs the DP: scratch columns and the survivor queue are its alone.
  g.tile_rows = kTileRows;
  g.col_dwords = 0;
  uint32_t table_ops = eng->n_table_ops;
  uint32_t words = kTileRows * g.lds_stride_dw + table_ops * (csdev::kEqTableBytes / 4 + csdev::kFnibDwords) +
                   csdev::kStatWords +
                   96 /* private mask table + slack in front of the tile, next-tile slot, look-ahead pad */;
  if (mode == csdev::MODE_SCAN) words += csdev::kRingRecords * 8;  // queue records waiting for their reservation
  g.item_slots = 0;
  if (mode == csdev::MODE_SCAN && eng->coded) {
    // item log of the leading adapter walk (trim_kernel.hip.inc, ItemLog): 8 + 1 bytes per item and lane.  As many
    // slots (2..4) as fit without costing a block per CU: LDS is handed out in granules of 1280 bytes on gfx950.
    const uint32_t per_slot = kTileRows * 9u;
    const uint32_t base = words * 4;
    const uint32_t blocks = std::min<uint32_t>(kLdsBudget / ((base + 1279u) / 1280u * 1280u), 4u * eng->waves_per_simd[mode]);
    const uint32_t room = blocks ? kLdsBudget / blocks / 1280u * 1280u : 0u;
    uint32_t slots = room > base ? (room - base) / per_slot : 0u;
    slots = std::max<uint32_t>(2u, std::min<uint32_t>(4u, slots));
    if (eng->knob_item_slots) slots = eng->knob_item_slots;
    g.item_slots = slots;
    words += (slots * per_slot + 3u) / 4u;
  }
  if (mode == csdev::MODE_RESOLVE) {
    g.col_dwords = eng->col_dwords;
    if (eng->knob_col_bytes / 4 > g.col_dwords) g.col_dwords = eng->knob_col_bytes / 4;
    words += g.col_dwords + 64 * csdev::kWaveItemDwords;
  }
  g.lds_bytes = words * 4;
  if (g.lds_bytes > kLdsBudget) return fail(CS_ERR_ARG, "stride %u does not fit the LDS tile", stride);
  return CS_OK;
}

// Adds the event-measured kernel durations of a lane's last call to the engine's totals (waits for that call).
int harvest(cs_engine *eng, Lane &ln) {
  if (!ln.unharvested) return CS_OK;
  float a = 0, b = 0;
  HIP_TRY(hipEventSynchronize(ln.ev_stop));
  HIP_TRY(hipEventElapsedTime(&a, ln.ev_start, ln.ev_mid));
  HIP_TRY(hipEventElapsedTime(&b, ln.ev_mid, ln.ev_stop));  // from "scan kernel done": includes the hand-over to the resolve stream
  eng->kernel_ms_total[0] += a;
  eng->kernel_ms_total[1] += b;
  ++eng->timed_calls;
  ln.unharvested = false;
  return CS_OK;
}

// Scan kernel on `stream`, resolve kernel on `rstream` behind it (the same stream, or the engine's
// resolve stream for the pipelined form: the next call's scan kernel then runs beside this call's resolve
// kernel, whose few latency-bound waves fit into what the scan kernel leaves idle).
int launch(cs_engine *eng, hipStream_t stream, hipStream_t rstream, const cs_reads *r1, const cs_reads *r2,
           uint32_t n_reads, uint32_t stride, bool time_it, const unsigned long long *gate = nullptr) {
  Geometry g[2];
  for (int mode = 0; mode < 2; ++mode) {
    int rc = geometry_for(eng, stride, mode, g[mode]);
    if (rc) return rc;
  }
  if (n_reads == 0) return CS_OK;
  if ((r2 != nullptr) != eng->paired) return fail(CS_ERR_ARG, "plan is %s-end", eng->paired ? "paired" : "single");
  csdev::KArgs a;
  memset(&a, 0, sizeof a);
  const cs_reads *rr[2] = {r1, r2};
  const uint32_t mates = r2 ? 2 : 1;
  for (uint32_t m = 0; m < mates; ++m) {
    if (!rr[m]->seq || !rr[m]->qual || !rr[m]->len || !rr[m]->out) return fail(CS_ERR_ARG, "null array in mate %u", m + 1);
    if (((uintptr_t)rr[m]->seq | (uintptr_t)rr[m]->qual) & 3) return fail(CS_ERR_ARG, "seq/qual must be 4-byte aligned");
    a.mate[m].seq = reinterpret_cast<const uint32_t *>(rr[m]->seq);
    a.mate[m].qual = rr[m]->qual;
    a.mate[m].len = rr[m]->len;
    a.mate[m].out = rr[m]->out;
    a.mate[m].cap2 = rr[m]->cap2;
    a.mate[m].bc = rr[m]->bc;
  }
  Lane &ln = eng->lanes[eng->next_lane];
  eng->next_lane = (eng->next_lane + 1) % kLanes;
  // the lane's previous call, on whatever stream it ran: as a rule it finished long ago (three calls back) and
  // the stream is spared a marker
  if (ln.used && hipEventQuery(ln.done) != hipSuccess) HIP_TRY(hipStreamWaitEvent(stream, ln.done, 0));
  // queue of deferred reads: a read is deferred at most once by the scan kernel, so n_reads records per
  // mate always suffice (32 bytes each; typically a few per cent are used)
  if (n_reads > ln.defer_capacity) {
    // every lane grows at once: the allocations (and their synchronisation) happen at the first call of a new
    // batch size, not one lane at a time over the first calls
    const uint32_t cap = n_reads + n_reads / 8 + 1024;
    for (Lane &l : eng->lanes) {
      if (cap <= l.defer_capacity) continue;
      if (l.used) HIP_TRY(hipEventSynchronize(l.done));  // an earlier launch may still be reading the old queue
      for (int m = 0; m < 2; ++m) {
        if (l.d_defer[m]) (void)hipFree(l.d_defer[m]);
        l.d_defer[m] = nullptr;
      }
      l.defer_capacity = 0;
      for (uint32_t m = 0; m < (eng->paired ? 2u : 1u); ++m)
        HIP_TRY(hipMalloc(&l.d_defer[m], (size_t)cap * kDeferRecordBytes));
      l.defer_capacity = cap;
    }
  }
  a.defer[0] = reinterpret_cast<uint4 *>(ln.d_defer[0]);
  a.defer[1] = reinterpret_cast<uint4 *>(ln.d_defer[1]);
  a.defer_count = ln.d_counters + kTileCounterDwords;
  a.defer_cap = ln.defer_capacity;
  a.stats = eng->d_stats;
  a.n_reads = n_reads;
  a.stride_dw = stride / 4;
  a.plan_slot = (uint32_t)eng->plan_slot;
  a.tile_counter = ln.d_counters;
  a.n_table_ops = eng->n_table_ops;
  a.batch_knob = eng->knob_batch;
  a.fast_recode = eng->knob_fast_recode;
  a.gate = gate;
  for (int mode = 0; mode < 2; ++mode)
    if (g[mode].lds_bytes > eng->max_dynamic_lds[mode]) {
      HIP_TRY(hipFuncSetAttribute(kernel_for(eng, mode), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g[mode].lds_bytes));
      eng->max_dynamic_lds[mode] = g[mode].lds_bytes;
    }
  // persistent blocks: as many as stay resident (LDS- or register-limited), each loops over its tiles
  const uint32_t n_tiles = (n_reads + kTileRows - 1) / kTileRows;
  uint32_t gx[2];
  for (int mode = 0; mode < 2; ++mode) {
    uint32_t per_cu = kLdsBudget / g[mode].lds_bytes;
    const uint32_t wave_cap = 4 * eng->waves_per_simd[mode];
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) per_cu = 1;
    // scan kernel: 2x the resident set, late blocks even out the tail; resolve kernel: the resident set
    // pipelined resolve kernel: it runs beside the next call's scan kernel -- one wave per SIMD (its 128
    // VGPRs are what four scan waves leave free), the scan kernel keeps the rest of the machine
    // (more resolve waves for the command line's small launches were tried in round 5 -- 8 per CU so that every batch of
    // 16 records has a wave of its own: 122 -> 137 us, dropped: the kernel's duration there is one wave's chain of round
    // trips, not the number of batches per wave)
    if (mode == csdev::MODE_RESOLVE && rstream != stream && per_cu > eng->knob_resolve_waves) per_cu = eng->knob_resolve_waves;
    uint32_t resident = (uint32_t)eng->n_cus * per_cu * (mode == csdev::MODE_SCAN ? 2u : 1u);
    gx[mode] = resident / mates;
    if (mode == csdev::MODE_SCAN && eng->knob_grid_x) gx[mode] = eng->knob_grid_x;
    if (gx[mode] < 1) gx[mode] = 1;
    if (gx[mode] > n_tiles) gx[mode] = n_tiles;
  }
  // tile hand-out (see the kernel): big units, about a quarter of a block's share (at most 8 tiles), for
  // the first half of the batch, single tiles for the rest (the counter's answer is not waited for any more:
  // small units cost nothing and even out the end of the kernel).
  a.static_units = gx[0] / 2 > 0 ? gx[0] / 2 : 1;  // the half of the grid that is resident from the start
  const uint32_t share = n_tiles / a.static_units;
  a.big_shift = share >= 32 ? 3 : share >= 16 ? 2 : 1;
  a.small_shift = 0;
  uint32_t big_pct = 50;
  if (eng->knob_units) {
    a.big_shift = eng->knob_big_shift;
    a.small_shift = eng->knob_small_shift;
    big_pct = eng->knob_big_pct;
  }
  a.big_tiles = (uint32_t)((uint64_t)n_tiles * big_pct / 100) & ~((1u << a.big_shift) - 1u);
  // (the lane's counters are zero: cleared at creation and again behind every resolve kernel, off the scan stream)
  if (time_it) {
    int rc = harvest(eng, ln);  // the call three before this one: long finished
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ln.ev_start, stream));
  }
  for (int mode = 0; mode < 2; ++mode) {
    a.lds_stride_dw = g[mode].lds_stride_dw;
    a.col_dwords = g[mode].col_dwords;
    a.n_table_ops = eng->n_table_ops;
    a.item_slots = g[mode].item_slots;
    void *kargs[] = {&a};
    hipStream_t st = mode == csdev::MODE_SCAN ? stream : rstream;
    if (mode == csdev::MODE_RESOLVE && rstream != stream) {
      // (one marker on the scan stream serves the timing and the hand-over: ev_mid when the call is timed)
      if (!time_it) HIP_TRY(hipEventRecord(ln.scanned, stream));
      HIP_TRY(hipStreamWaitEvent(rstream, time_it ? ln.ev_mid : ln.scanned, 0));
    }
    HIP_TRY(hipLaunchKernel(kernel_for(eng, mode), dim3(gx[mode], mates, 1), dim3(kTileRows, 1, 1), kargs,
                            g[mode].lds_bytes, st));
    HIP_TRY(hipGetLastError());
    if (time_it && mode == csdev::MODE_SCAN) HIP_TRY(hipEventRecord(ln.ev_mid, stream));
  }
  if (time_it) HIP_TRY(hipEventRecord(ln.ev_stop, rstream));
  HIP_TRY(hipMemsetAsync(ln.d_counters, 0, kTileCounterBytes, rstream));  // ready for the lane's next call
  HIP_TRY(hipEventRecord(ln.done, rstream));
  ln.used = true;
  ln.timed = time_it;
  ln.unharvested = time_it;
  eng->last_lane = &ln;
  return CS_OK;
}

}  // namespace

extern "C" {

int cs_abi_version(void) { return CS_ABI_VERSION; }
const char *cs_last_error(void) { return g_err; }

int cs_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(CS_ERR_NO_GPU, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

int cs_plan_create(const cs_op *ops_r1, int n1, const cs_op *ops_r2, int n2, const cs_params *params,
                   cs_plan **out) {
  if (!out) return fail(CS_ERR_ARG, "out is null");
  *out = nullptr;
  if (!params || params->abi_version != CS_ABI_VERSION) return fail(CS_ERR_ARG, "cs_params.abi_version mismatch");
  if (params->select_rule > CS_SELECT_SCORE) return fail(CS_ERR_ARG, "unknown select_rule");
  if (n1 < 0 || n1 > CS_MAX_OPS || n2 < 0 || n2 > CS_MAX_OPS) return fail(CS_ERR_ARG, "op count out of range");
  if ((n1 && !ops_r1) || (n2 && !ops_r2)) return fail(CS_ERR_ARG, "null op table");
  cs_plan *p = new (std::nothrow) cs_plan();
  if (!p) return fail(CS_ERR_NOMEM, "out of memory");
  memset(&p->host, 0, sizeof p->host);
  p->host.params = *params;
  p->host.n_ops[0] = n1;
  p->host.n_ops[1] = n2;
  const cs_op *src[2] = {ops_r1, ops_r2};
  const int cnt[2] = {n1, n2};
  for (int m = 0; m < 2; ++m)
    for (int i = 0; i < cnt[m]; ++i) {
      int rc = build_dev_op(src[m][i], p->host.ops[m][i], i, m + 1);
      if (rc) {
        delete p;
        return rc;
      }
    }
  // (one existence-only op per mate: its reads share ONE reservation of the queue's top region per tile)
  for (int m = 0; m < 2; ++m) {
    bool seen = false;
    for (int i = 0; i < cnt[m]; ++i) {
      if (seen) p->host.ops[m][i].exists_only = 0;
      seen = seen || p->host.ops[m][i].exists_only != 0;
    }
  }
  // coded tile: possible when every adapter base is A/C/G/T (always true for cutseq schemes)
  bool coded = true;
  for (int m = 0; m < 2; ++m)
    for (int i = 0; i < cnt[m]; ++i) {
      const cs_op &op = p->host.ops[m][i].op;
      if (op.kind != CS_OP_ADAPTER) continue;
      for (int j = 0; j < op.m; ++j) coded = coded && is_acgt(op.seq[j]);
    }
  p->host.coded = coded ? 1 : 0;
  if (coded)
    for (int m = 0; m < 2; ++m)
      for (int i = 0; i < cnt[m]; ++i) {
        csdev::DevOp &d = p->host.ops[m][i];
        if (d.op.kind != CS_OP_ADAPTER) continue;
        for (int j = 0; j < d.op.m; ++j) d.op.seq[j] = (uint8_t)csdev::base_code(d.op.seq[j]);
        d.base_char = csdev::base_code(d.base_char);
      }
  // the tables a block copies into LDS when it starts
  static_assert(csdev::kFnibDwords == 6, "DevOp::fnib");
  for (int m = 0; m < 2; ++m)
    for (int i = 0; i < cnt[m]; ++i) {
      csdev::DevOp &d = p->host.ops[m][i];
      memset(d.eq_tab, 0, sizeof d.eq_tab);
      memset(d.fnib, 0, sizeof d.fnib);
      if (d.op.kind != CS_OP_ADAPTER) continue;
      const int mm = d.op.m;
      for (int ent = 0; ent < 4; ++ent) {
        const int which = (ent == 0) ? 0 : (ent == 1) ? 1 : (ent == 2) ? 3 : 2;  // sel 2 = T, sel 3 = G
        const uint64_t mask = d.peq[which];
        d.eq_tab[ent * 2] = (uint32_t)mask;
        d.eq_tab[ent * 2 + 1] = (mm >= 1 && mm <= 32) ? ((uint32_t)mask << (32 - mm)) : (uint32_t)(mask >> 32);
      }
      if (coded)
        for (int idx = 0; idx < mm && idx < 32; ++idx)
          d.fnib[idx >> 3] |= (uint32_t)((d.op.seq[d.op.reversed ? mm - 1 - idx : idx] >> 3) & 7u) << (4 * (idx & 7));
    }
  // runs of consecutive CUT ops: walked in one turn of the op loop from the run's first op (DevOp::cut_run / cut_pack)
  for (int m = 0; m < 2; ++m)
    for (int i = 0; i < cnt[m]; ++i) {
      csdev::DevOp &d = p->host.ops[m][i];
      if (d.op.kind != CS_OP_CUT) continue;
      auto pack = [](const cs_op &o) {
        return (uint32_t)(uint16_t)o.cut_len | ((uint32_t)(o.conditional ? 1u : 0u) << 16) | ((uint32_t)(o.capture & 3u) << 17) |
               ((uint32_t)(o.force_min_len > 0x1fff ? 0x1fffu : o.force_min_len) << 19);
      };
      d.cut_run = 1;
      d.cut_pack[0] = pack(d.op);
    }
  {
    const char *env = getenv("CUTSEQ_CUT_RUNS");
    if (!(env && atoi(env) == 0))
      for (int m = 0; m < 2; ++m)
        for (int i = 0; i < cnt[m];) {
          if (p->host.ops[m][i].op.kind != CS_OP_CUT) {
            ++i;
            continue;
          }
          int j = i;
          while (j < cnt[m] && j - i < csdev::kCutRun && p->host.ops[m][j].op.kind == CS_OP_CUT) {
            p->host.ops[m][i].cut_pack[j - i] = p->host.ops[m][j].cut_pack[0];
            ++j;
          }
          p->host.ops[m][i].cut_run = (uint32_t)(j - i);
          i = j;
        }
  }
  // The pair that opens the reference's chains (cutseq/run.py:332-355, 544-590): an existence-only 5' op at the head
  // of the chain with a forward, free-ended Myers-32 op right behind it -> one merged forward walk in the scan kernel
  // (trim_kernel.hip.inc, myers_pair).  CUTSEQ_PAIR=0 keeps the two scans apart.
  {
    const char *env = getenv("CUTSEQ_PAIR");
    const bool allow = coded && !(env && atoi(env) == 0);
    // The walk LOGS the groups of eight columns that hold a candidate of its exact op in a few LDS slots per lane and
    // sends a read with more flagged groups than slots through the resolve kernel (trim_kernel.hip.inc, ItemLog).  That
    // is the rare case for the adapters the reference compiles (20 bases, four errors: a random 150-base read holds a
    // candidate one time in a thousand); a short adapter with a loose error bound matches random sequence all the
    // time, and every such read would take the slow road.  Estimate: candidates per 300 random bases ~ 300 * sum over
    // e <= k of C(m, e) * 6^e / 4^m (three substitutions and about as many indel variants per error); above 3 % the op
    // keeps the op loop's own filter, which does its book-keeping in place.
    // CUTSEQ_LOG_ALWAYS=1 (tests): every op counts as log-friendly -- the exhaustive small-universe sweep drives short
    // adapters through the merged walk that way (tests/test_gpu_exhaustive.py); results are the same either way.
    const char *env_log = getenv("CUTSEQ_LOG_ALWAYS");
    const bool log_always = env_log && atoi(env_log) != 0;
    auto log_friendly = [log_always](int mm, int kk) {
      if (log_always) return true;
      double variants = 0, choose = 1, pw = 1;
      for (int e = 0; e <= kk && e <= mm; ++e) {
        variants += choose * pw;
        choose = choose * (double)(mm - e) / (double)(e + 1);
        pw *= 6.0;
      }
      double space = 1;
      for (int i = 0; i < mm; ++i) space *= 4.0;
      return 300.0 * variants / space < 0.03;
    };
    for (int m = 0; m < 2; ++m) {
      if (allow && cnt[m] >= 1) {  // a forward, free-ended Myers-32 op that opens the chain alone: the same walk, one recurrence
        csdev::DevOp &f = p->host.ops[m][0];
        const bool free_ends = (f.op.align_flags & (CS_QUERY_START | CS_QUERY_STOP)) == (CS_QUERY_START | CS_QUERY_STOP);
        if (f.op.kind == CS_OP_ADAPTER && f.filter_mode == csdev::FILTER_MYERS32 && !f.op.reversed && free_ends &&
            f.op.shortcut == CS_SHORTCUT_NONE && !f.exists_only && log_friendly(f.op.m, f.op.k))
          f.solo_first = 1;
      }
      if (!allow || cnt[m] < 2) continue;
      csdev::DevOp &a = p->host.ops[m][0];
      const csdev::DevOp &b = p->host.ops[m][1];
      const int ma = a.op.m, ka = a.op.k, mo = a.op.min_overlap;
      const bool ends_free = (b.op.align_flags & (CS_QUERY_START | CS_QUERY_STOP)) == (CS_QUERY_START | CS_QUERY_STOP);
      if (!a.exists_only || a.op.kind != CS_OP_ADAPTER || b.op.kind != CS_OP_ADAPTER || b.exists_only ||
          b.filter_mode != csdev::FILTER_MYERS32 || b.op.reversed || !ends_free || b.op.shortcut != CS_SHORTCUT_NONE ||
          ma + ka > 31 || ka > 14 || !log_friendly(b.op.m, b.op.k))
        continue;
      a.pair_next = 1;
      for (int ent = 0; ent < 4; ++ent) {
        const int which = (ent == 0) ? 0 : (ent == 1) ? 1 : (ent == 2) ? 3 : 2;  // sel 2 = T, sel 3 = G
        uint32_t fwd = 0;
        for (int i = 0; i < ma; ++i)
          if ((a.peq[which] >> (ma - 1 - i)) & 1ull) fwd |= 1u << i;  // a.op.seq holds the REVERSED adapter
        a.eq_fwd[ent] = fwd << (32 - ma);
      }
      // T(j): the largest thr[i] over suffix lengths i in [min_overlap, m] that a candidate ending in column j can
      // have (|i - j| <= thr[i]: an alignment of cost c moves at most c columns off the diagonal); none: -1
      for (int j = 1; j <= 32; ++j) {
        int best = -1;
        if (j > ma + ka) best = ka;
        else
          for (int i = mo; i <= ma; ++i) {
            const int d = i > j ? i - j : j - i;
            if (d <= (int)a.op.thr[i] && (int)a.op.thr[i] > best) best = a.op.thr[i];
          }
        a.pair_tn[(j - 1) >> 3] |= (uint32_t)(best + 1) << (4 * ((j - 1) & 7));
      }
    }
  }
  *out = p;
  return CS_OK;
}

void cs_plan_destroy(cs_plan *plan) { delete plan; }

int cs_plan_set_demux(cs_plan *plan, int mate, int op_index, const uint16_t *table, size_t entries) {
  if (!plan || !table) return fail(CS_ERR_ARG, "null plan or table");
  if (mate < 1 || mate > 2 || op_index < 0 || op_index >= plan->host.n_ops[mate - 1])
    return fail(CS_ERR_ARG, "mate %d op %d: no such op", mate, op_index);
  const cs_op &op = plan->host.ops[mate - 1][op_index].op;
  if (op.kind != CS_OP_DEMUX) return fail(CS_ERR_ARG, "mate %d op %d is not a CS_OP_DEMUX op", mate, op_index);
  if (plan->host.ops[mate - 1][op_index].filter_mode)
    return fail(CS_ERR_ARG, "mate %d op %d: m + k = %d > %d (or CS_DEMUX_BY_OPS) takes cs_plan_set_demux_ops, not a table", mate,
                op_index, op.m + op.k, CS_DEMUX_MAX_PREFIX);
  size_t want = 0, pw = 1;
  for (int l = 0; l <= op.m + op.k; ++l, pw *= 5) want += pw;
  if (entries != want) return fail(CS_ERR_ARG, "demux table: %zu entries, expected %zu for m + k = %d", entries, want, op.m + op.k);
  try {
    plan->demux[mate - 1][op_index].assign(table, table + entries);
  } catch (const std::bad_alloc &) {
    return fail(CS_ERR_NOMEM, "out of memory");
  }
  return CS_OK;
}

namespace {

// Which barcodes can still match a read that starts with a given prefix?  Depth-first over the prefixes (alphabet
// A, C, T, G, other -- the digits of the table index), one edit-distance column per barcode that is still alive; a
// barcode is alive while some cell of its column is <= k (Ukkonen: the column minimum never falls again) or its last
// row was <= k at an earlier column (a match that ends inside the prefix).  k = thr[m] bounds every length's
// threshold, so the lists are supersets of the barcodes whose PrefixAdapter matches; the device runs those ops
// themselves on the candidates (trim_kernel.hip.inc, CS_OP_DEMUX).
// The walk is cut at depth 2: the subtrees below are independent and go to a few threads; every piece of the
// pre-order (the top of the tree between two subtrees, a subtree) writes a SEGMENT of its own, and the segments are
// laid into the table in pre-order afterwards -- the same table as a serial walk, whatever the thread count.
struct DemuxSegment {
  std::vector<uint8_t> pool;                          // candidate lists of this piece
  std::vector<std::pair<size_t, uint32_t>> first;     // (table slot, offset into `pool` << 8 | candidates)
};

struct DemuxWalk {
  const std::vector<std::string> &digits;  // per barcode: its bases as digits 0..3
  int m, k, depth;
  const std::vector<size_t> &block;  // where the prefixes of each length start

  struct Alive {
    int id;
    bool done;
    uint8_t col[CS_MAX_ADAPTER + 1];
  };
  struct Task {
    int len;
    size_t idx, pw, segment;
    std::vector<Alive> alive;
  };
  std::vector<std::vector<Alive>> scratch;  // walk(): the children's lists, one per depth
  DemuxSegment *seg = nullptr;              // the piece being written
  bool overflow = false;
  // top of the tree only: where to cut, the pieces so far, the subtrees left to others
  int split = -1;
  std::vector<DemuxSegment> *segments = nullptr;
  std::vector<Task> *tasks = nullptr;

  void emit(size_t at, const std::vector<Alive> &alive) {
    if (alive.empty()) return;
    if (seg->pool.size() + alive.size() >= (1u << 24) || alive.size() > 255) {
      overflow = true;
      return;
    }
    seg->first.emplace_back(at, (uint32_t)(seg->pool.size() << 8) | (uint32_t)alive.size());
    for (const Alive &a : alive) seg->pool.push_back((uint8_t)a.id);
  }

  void walk(int len, size_t idx, size_t pw, const std::vector<Alive> &alive) {
    if (len == split && len < depth) {  // a subtree for the threads: its segment keeps this place in the order
      segments->emplace_back();
      tasks->push_back(Task{len, idx, pw, segments->size() - 1, alive});
      segments->emplace_back();  // the top of the tree goes on in a fresh piece behind it
      seg = nullptr;             // (re-pointed by the caller: `segments` may have moved)
      return;
    }
    if (!seg) seg = &segments->back();
    emit(block[len] + idx, alive);
    if (len == depth || overflow) return;
    // (one list per depth, reused by every node of that depth: two million nodes would otherwise allocate one each)
    if (scratch.size() <= (size_t)len) scratch.resize((size_t)depth + 1);
    for (int c = 0; c < 5; ++c) {
      std::vector<Alive> &next = scratch[len];
      next.clear();
      for (const Alive &a : alive) {
        next.emplace_back();
        Alive &b = next.back();
        b.id = a.id;
        if (a.done) {  // matched inside the prefix already: a candidate whatever follows, its column is not looked at again
          b.done = true;
          continue;
        }
        const char *bc = digits[a.id].data();
        const int cap = k + 1;  // costs are clamped there: nothing above k is ever told apart
        // row i of the column behind len + 1 bases costs at least |i - (len + 1)| (the prefix is anchored): only the
        // 2k + 1 rows around the diagonal can be below the cap, the others hold it
        const int lo = std::max(1, len + 1 - k), hi = std::min(m, len + 1 + k);
        memset(b.col, cap, (size_t)m + 1);
        int best = b.col[0] = (uint8_t)std::min(len + 1, cap);
        for (int i = lo; i <= hi; ++i) {
          const int sub = a.col[i - 1] + ((c < 4 && bc[i - 1] == c) ? 0 : 1);
          const int v = std::min(std::min(sub, std::min(a.col[i] + 1, b.col[i - 1] + 1)), cap);
          b.col[i] = (uint8_t)v;
          best = std::min(best, v);
        }
        b.done = b.col[m] <= k;
        if (!b.done && best > k) next.pop_back();
      }
      if (!next.empty()) walk(len + 1, idx + (size_t)c * pw, pw * 5, next);
    }
  }
};

// the whole table: top of the tree here, subtrees on up to eight threads, segments merged in pre-order
bool demux_walk_all(const std::vector<std::string> &digits, int m, int k, DemuxLong &dl, const std::vector<size_t> &block) {
  const int n = (int)digits.size();
  std::vector<DemuxWalk::Alive> all((size_t)n);
  for (int b = 0; b < n; ++b) {
    all[b].id = b;
    all[b].done = false;
    for (int i = 0; i <= m; ++i) all[b].col[i] = (uint8_t)std::min(i, k + 1);
  }
  std::vector<DemuxSegment> segments(1);
  std::vector<DemuxWalk::Task> tasks;
  DemuxWalk top{digits, m, k, dl.depth, block};
  top.split = 2;
  top.segments = &segments;
  top.tasks = &tasks;
  top.walk(0, 0, 1, all);
  if (top.overflow) return false;
  std::atomic<size_t> next_task{0};
  std::atomic<bool> overflow{false};
  auto worker = [&]() {
    DemuxWalk w{digits, m, k, dl.depth, block};
    for (size_t t; (t = next_task.fetch_add(1)) < tasks.size();) {
      w.seg = &segments[tasks[t].segment];
      w.walk(tasks[t].len, tasks[t].idx, tasks[t].pw, tasks[t].alive);
      if (w.overflow) overflow = true;
    }
  };
  unsigned n_threads = std::min<unsigned>(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())), (unsigned)tasks.size());
  if (const char *env = getenv("CUTSEQ_HOST_THREADS"))  // (the command line's -t/--threads, fastq.set_threads)
    n_threads = std::max(1, std::min((int)n_threads, atoi(env)));
  std::vector<std::thread> pool;
  for (unsigned i = 1; i < n_threads; ++i) pool.emplace_back(worker);
  worker();
  for (std::thread &t : pool) t.join();
  if (overflow) return false;
  for (const DemuxSegment &sg : segments) {
    const size_t base = dl.pool.size();
    if (base + sg.pool.size() >= (1u << 24)) return false;
    for (const auto &f : sg.first) dl.first[f.first] = f.second + (uint32_t)(base << 8);
    dl.pool.insert(dl.pool.end(), sg.pool.begin(), sg.pool.end());
  }
  return true;
}

}  // namespace

int cs_plan_set_demux_ops(cs_plan *plan, int mate, int op_index, const cs_op *ops, int n_ops) {
  if (!plan || !ops) return fail(CS_ERR_ARG, "null plan or ops");
  if (mate < 1 || mate > 2 || op_index < 0 || op_index >= plan->host.n_ops[mate - 1])
    return fail(CS_ERR_ARG, "mate %d op %d: no such op", mate, op_index);
  const cs_op &op = plan->host.ops[mate - 1][op_index].op;
  if (op.kind != CS_OP_DEMUX) return fail(CS_ERR_ARG, "mate %d op %d is not a CS_OP_DEMUX op", mate, op_index);
  if (!plan->host.ops[mate - 1][op_index].filter_mode)
    return fail(CS_ERR_ARG, "mate %d op %d: m + k = %d <= %d takes a table (cs_plan_set_demux) unless the op says CS_DEMUX_BY_OPS",
                mate, op_index, op.m + op.k, CS_DEMUX_MAX_PREFIX);
  if (n_ops < 1 || n_ops > 255) return fail(CS_ERR_ARG, "between 1 and 255 barcodes");
  try {
    DemuxLong dl;
    std::vector<std::string> digits;
    for (int b = 0; b < n_ops; ++b) {
      const cs_op &in = ops[b];
      const bool at_end = op.reversed != 0;  // barcodes at the 3' end: SuffixAdapter ops, the table walks the read backwards
      if (in.kind != CS_OP_ADAPTER || in.align_flags != (at_end ? CS_WHERE_SUFFIX : CS_WHERE_PREFIX) || in.reversed ||
          in.remove != (at_end ? CS_REMOVE_AFTER : CS_REMOVE_BEFORE) || in.shortcut != CS_SHORTCUT_NONE || in.m != op.m ||
          in.k != op.k || in.min_overlap != in.m)
        return fail(CS_ERR_ARG, "barcode %d: not the %s op of a %u-base barcode with %u errors", b,
                    at_end ? "SuffixAdapter" : "PrefixAdapter", op.m, op.k);
      std::string dg;
      for (int i = 0; i < in.m; ++i) {
        const uint8_t c = in.seq[at_end ? in.m - 1 - i : i];
        if (!is_acgt(c)) return fail(CS_ERR_ARG, "barcode %d: bases other than A, C, G, T", b);
        dg.push_back((char)((c >> 1) & 3));
      }
      digits.push_back(dg);
      csdev::DevOp d;
      int rc = build_dev_op(in, d, b, mate);
      if (rc) return rc;
      if (plan->host.coded)
        for (int j = 0; j < d.op.m; ++j) d.op.seq[j] = (uint8_t)csdev::base_code(d.op.seq[j]);
      dl.ops.push_back(d);
    }
    // nine bases deep where the lists fit the table format (16 M entries); many long barcodes with many errors: one
    // base less at a time (shorter prefixes, fewer and longer lists -- the device tries a few more barcodes per read)
    bool fits = false;
    for (int depth = std::min((int)op.m + (int)op.k, 9); depth >= 1 && !fits; --depth) {
      dl.depth = depth;
      dl.pool.clear();
      std::vector<size_t> block(depth + 2, 0);
      size_t pw = 1;
      for (int l = 0; l <= depth; ++l, pw *= 5) block[l + 1] = block[l] + pw;
      dl.first.assign(block[depth + 1], 0u);
      fits = demux_walk_all(digits, (int)op.m, (int)op.k, dl, block);
    }
    if (!fits) return fail(CS_ERR_ARG, "demux: candidate lists exceed the table format");
    plan->demux_long[mate - 1][op_index] = std::move(dl);
  } catch (const std::bad_alloc &) {
    return fail(CS_ERR_NOMEM, "out of memory");
  }
  return CS_OK;
}

void cs_engine_destroy(cs_engine *eng) {
  if (!eng) return;
  if (eng->device >= 0) (void)hipSetDevice(eng->device);
  for (Slot &s : eng->slots) {
    for (int m = 0; m < 2; ++m) {
      if (s.d_seq[m]) (void)hipFree(s.d_seq[m]);
      if (s.d_qual[m]) (void)hipFree(s.d_qual[m]);
      if (s.d_len[m]) (void)hipFree(s.d_len[m]);
      if (s.d_out[m]) (void)hipFree(s.d_out[m]);
      if (s.d_bc[m]) (void)hipFree(s.d_bc[m]);
    }
    if (s.d_cap2) (void)hipFree(s.d_cap2);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  if (eng->plan_slot >= 0) release_plan_slot(eng->device, eng->plan_slot);
  if (eng->d_stats) (void)hipFree(eng->d_stats);
  for (Lane &ln : eng->lanes) {
    if (ln.used) (void)hipEventSynchronize(ln.done);
    if (ln.d_counters) (void)hipFree(ln.d_counters);
    for (int m = 0; m < 2; ++m) {
      if (ln.d_defer[m]) (void)hipFree(ln.d_defer[m]);
    }
    for (hipEvent_t ev : {ln.scanned, ln.done, ln.ev_start, ln.ev_mid, ln.ev_stop})
      if (ev) (void)hipEventDestroy(ev);
  }
  for (void *t : eng->d_tables) (void)hipFree(t);
  if (eng->resolve_stream) (void)hipStreamDestroy(eng->resolve_stream);
  if (eng->stream) (void)hipStreamDestroy(eng->stream);
  delete eng;
}

int cs_engine_create(const cs_plan *plan, int device, uint32_t n_slots, uint32_t max_reads, uint32_t max_stride,
                     cs_engine **out) {
  if (!out) return fail(CS_ERR_ARG, "out is null");
  *out = nullptr;
  if (!plan) return fail(CS_ERR_ARG, "plan is null");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(CS_ERR_NO_GPU, "no HIP device visible (%s); this engine has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail(CS_ERR_ARG, "device %d out of range (0..%d)", device, ndev - 1);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(CS_ERR_NO_GPU, "device %d is %s; the kernels are built for gfx950 (MI355X) only", device, prop.gcnArchName);
  if (n_slots) {
    if (max_stride == 0 || max_stride % 4 || max_stride > CS_MAX_STRIDE)
      return fail(CS_ERR_ARG, "max_stride %u must be a multiple of 4 in [4, %d]", max_stride, CS_MAX_STRIDE);
    if (!max_reads) return fail(CS_ERR_ARG, "max_reads is 0");
  }
  cs_engine *eng = new (std::nothrow) cs_engine();
  if (!eng) return fail(CS_ERR_NOMEM, "out of memory");
  eng->device = device;
  eng->paired = plan->host.n_ops[1] > 0;
  eng->coded = plan->host.coded != 0;
  eng->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  eng->n_table_ops = (uint32_t)(plan->host.n_ops[0] > plan->host.n_ops[1] ? plan->host.n_ops[0] : plan->host.n_ops[1]);
  if (eng->n_table_ops < 1) eng->n_table_ops = 1;
  // DP scratch per wave: 16 survivors x (m+1) cells for the cooperative strips (m <= 32, ACGT),
  // two column slots for everything that falls back to the one-lane-per-survivor DP
  eng->col_dwords = 64;
  bool has_long_demux = false;
  for (int mt = 0; mt < 2; ++mt)
    for (int i = 0; i < plan->host.n_ops[mt]; ++i) {
      const csdev::DevOp &d = plan->host.ops[mt][i];
      if (d.op.kind == CS_OP_DEMUX) eng->demux_mate = mt;
      if (d.op.kind == CS_OP_DEMUX && d.filter_mode) {
        // the resolve kernel runs the barcodes' own ops, one column per lane
        has_long_demux = true;
        if (64u * (d.op.m + 1u) > eng->col_dwords) eng->col_dwords = 64u * (d.op.m + 1u);
      }
      if (d.op.kind != CS_OP_ADAPTER) continue;
      if (d.filter_mode == csdev::FILTER_MYERS64) eng->wide = true;
      const uint32_t need = (d.acgt_only && d.op.m <= 32) ? 16u * (d.op.m + 1u) : 2u * (d.op.m + 1u);
      if (need > eng->col_dwords) eng->col_dwords = need;
    }
  eng->long_demux = has_long_demux;
  eng->max_reads = max_reads;
  eng->max_stride = max_stride;
#define ENG_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      fail(CS_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      cs_engine_destroy(eng);                                                             \
      return CS_ERR_HIP;                                                                  \
    }                                                                                     \
  } while (0)
  ENG_TRY(hipSetDevice(device));
  ENG_TRY(hipStreamCreateWithFlags(&eng->stream, hipStreamNonBlocking));
  {
    // the resolve stream outranks the scan stream: when one batch's resolve kernel and the next batch's scan
    // kernel become ready together, the resolve kernel's few blocks are placed first (the scan kernel would
    // fill every SIMD and leave them waiting until it ends)
    int least = 0, greatest = 0;
    ENG_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    ENG_TRY(hipStreamCreateWithPriority(&eng->resolve_stream, hipStreamNonBlocking, greatest));
  }
  for (Lane &ln : eng->lanes) {
    ENG_TRY(hipMalloc(&ln.d_counters, kTileCounterBytes));
    ENG_TRY(hipMemset(ln.d_counters, 0, kTileCounterBytes));
    ENG_TRY(hipEventCreateWithFlags(&ln.scanned, hipEventDisableTiming));
    ENG_TRY(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
    ENG_TRY(hipEventCreate(&ln.ev_start));
    ENG_TRY(hipEventCreate(&ln.ev_mid));
    ENG_TRY(hipEventCreate(&ln.ev_stop));
  }
  eng->plan_slot = acquire_plan_slot(device);
  if (eng->plan_slot < 0) {
    fail(CS_ERR_STATE, "more than %d live engines on device %d", csdev::kMaxPlanSlots, device);
    cs_engine_destroy(eng);
    return CS_ERR_STATE;
  }
  {
    // the device copy of the plan carries this device's addresses of the demux tables
    csdev::DevPlan dp = plan->host;
    for (int mt = 0; mt < 2; ++mt)
      for (int i = 0; i < plan->host.n_ops[mt]; ++i) {
        sync_fields(dp.ops[mt][i]);
        if (dp.ops[mt][i].op.kind != CS_OP_DEMUX) continue;
        if (dp.ops[mt][i].filter_mode) {
          // the barcodes' own ops, the table of candidates, the candidate lists: one allocation
          const DemuxLong &dl = plan->demux_long[mt][i];
          if (dl.ops.empty()) {
            fail(CS_ERR_STATE, "mate %d op %d: CS_OP_DEMUX without its barcode ops (cs_plan_set_demux_ops)", mt + 1, i);
            cs_engine_destroy(eng);
            return CS_ERR_STATE;
          }
          const size_t b_ops = dl.ops.size() * sizeof(csdev::DevOp), b_first = dl.first.size() * sizeof(uint32_t);
          const size_t o_first = (b_ops + 255) & ~(size_t)255, o_pool = (o_first + b_first + 255) & ~(size_t)255;
          uint8_t *d = nullptr;
          ENG_TRY(hipMalloc(&d, o_pool + dl.pool.size() + 256));
          eng->d_tables.push_back(d);
          std::vector<csdev::DevOp> bops = dl.ops;
          for (csdev::DevOp &bo : bops) sync_fields(bo);
          ENG_TRY(hipMemcpy(d, bops.data(), b_ops, hipMemcpyHostToDevice));
          ENG_TRY(hipMemcpy(d + o_first, dl.first.data(), b_first, hipMemcpyHostToDevice));
          if (!dl.pool.empty()) ENG_TRY(hipMemcpy(d + o_pool, dl.pool.data(), dl.pool.size(), hipMemcpyHostToDevice));
          dp.ops[mt][i].peq[0] = (uint64_t)(uintptr_t)(d + o_first);
          dp.ops[mt][i].peq[1] = (uint64_t)(uintptr_t)(d + o_pool);
          dp.ops[mt][i].peq[2] = (uint64_t)(uintptr_t)d;
          dp.ops[mt][i].peq[3] = (uint64_t)dl.depth;
          continue;
        }
        const std::vector<uint16_t> &tab = plan->demux[mt][i];
        if (tab.empty()) {
          fail(CS_ERR_STATE, "mate %d op %d: CS_OP_DEMUX without a table (cs_plan_set_demux)", mt + 1, i);
          cs_engine_destroy(eng);
          return CS_ERR_STATE;
        }
        void *d = nullptr;
        ENG_TRY(hipMalloc(&d, tab.size() * sizeof(uint16_t)));
        eng->d_tables.push_back(d);
        ENG_TRY(hipMemcpy(d, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        dp.ops[mt][i].peq[0] = (uint64_t)(uintptr_t)d;
        dp.ops[mt][i].peq[1] = (uint64_t)tab.size();
      }
    ENG_TRY(hipMemcpyToSymbol(HIP_SYMBOL(csdev::c_plans), &dp, sizeof(csdev::DevPlan),
                              (size_t)eng->plan_slot * sizeof(csdev::DevPlan), hipMemcpyHostToDevice));
  }
  ENG_TRY(hipMalloc(&eng->d_stats, 2 * sizeof(cs_stats)));
  ENG_TRY(hipMemset(eng->d_stats, 0, 2 * sizeof(cs_stats)));
  eng->slots.resize(n_slots);
  const size_t bytes = (size_t)max_reads * max_stride;
  for (Slot &s : eng->slots) {
    for (int m = 0; m < (eng->paired ? 2 : 1); ++m) {
      ENG_TRY(hipMalloc(&s.d_seq[m], bytes));
      ENG_TRY(hipMalloc(&s.d_qual[m], bytes));
      ENG_TRY(hipMalloc(&s.d_len[m], (size_t)max_reads * sizeof(uint16_t)));
      ENG_TRY(hipMalloc(&s.d_out[m], (size_t)max_reads * sizeof(cs_result)));
      ENG_TRY(hipMalloc(&s.d_bc[m], (size_t)max_reads));
    }
    ENG_TRY(hipMalloc(&s.d_cap2, (size_t)max_reads * sizeof(cs_cap2)));
    ENG_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  }
  ENG_TRY(hipDeviceSynchronize());  // the memsets above ran on the NULL stream; the engine's streams do not wait for it
#undef ENG_TRY
  for (int mode = 0; mode < 2; ++mode) {
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, kernel_for(eng, mode)) == hipSuccess && fa.numRegs > 0) {
      const uint32_t alloc = ((uint32_t)fa.numRegs + 7u) / 8u * 8u;
      uint32_t w = 512u / alloc;
      eng->waves_per_simd[mode] = w < 1 ? 1 : (w > 8 ? 8 : w);
    }
  }
  // tuning knobs (diagnostic): read here, once, not on the launch path
  if (const char *env = getenv("CUTSEQ_COL_BYTES")) {  // LDS bytes of DP scratch per wave
    const long v = atol(env);
    if (v >= 1024 && v <= 32768) eng->knob_col_bytes = (uint32_t)v;
  }
  if (const char *env = getenv("CUTSEQ_BATCH_KNOB")) eng->knob_batch = (uint32_t)atol(env);
  if (const char *env = getenv("CUTSEQ_ITEM_SLOTS")) {
    const long v = atol(env);
    if (v >= 1 && v <= 16) eng->knob_item_slots = (uint32_t)v;
  }
  if (const char *env = getenv("CUTSEQ_FAST_RECODE")) eng->knob_fast_recode = (uint32_t)atoi(env) <= 2u ? (uint32_t)atoi(env) : 1u;
  if (const char *env = getenv("CUTSEQ_RESOLVE_WAVES")) {
    const long v = atol(env);
    if (v >= 1 && v <= 16) eng->knob_resolve_waves = (uint32_t)v;
  }
  if (const char *env = getenv("CUTSEQ_GRID_X")) {
    const long v = atol(env);
    if (v > 0) eng->knob_grid_x = (uint32_t)v;
  }
  if (const char *env = getenv("CUTSEQ_UNITS")) {  // "big_shift,small_shift,big_pct"
    unsigned bs, ss, pc;
    if (sscanf(env, "%u,%u,%u", &bs, &ss, &pc) == 3 && bs <= 6 && ss <= bs && pc <= 100) {
      eng->knob_units = true;
      eng->knob_big_shift = bs;
      eng->knob_small_shift = ss;
      eng->knob_big_pct = pc;
    }
  }
  *out = eng;
  return CS_OK;
}

int cs_trim_device(cs_engine *eng, void *stream, const cs_reads *r1, const cs_reads *r2, uint32_t n_reads,
                   uint32_t stride) {
  if (!eng || !r1) return fail(CS_ERR_ARG, "null engine or reads");
  HIP_TRY(hipSetDevice(eng->device));
  hipStream_t st = stream ? (hipStream_t)stream : eng->stream;
  return launch(eng, st, st, r1, r2, n_reads, stride, true);
}

int cs_trim_device_pipelined(cs_engine *eng, void *stream, const cs_reads *r1, const cs_reads *r2, uint32_t n_reads,
                             uint32_t stride) {
  if (!eng || !r1) return fail(CS_ERR_ARG, "null engine or reads");
  HIP_TRY(hipSetDevice(eng->device));
  hipStream_t st = stream ? (hipStream_t)stream : eng->stream;
  return launch(eng, st, eng->resolve_stream, r1, r2, n_reads, stride, true);
}

int cs_join(cs_engine *eng, void *stream) {
  if (!eng) return fail(CS_ERR_ARG, "null engine");
  HIP_TRY(hipSetDevice(eng->device));
  hipStream_t st = stream ? (hipStream_t)stream : eng->stream;
  // every lane's last resolve kernel (the resolve stream is in order, but joined calls on other streams
  // record their `done` elsewhere)
  for (Lane &ln : eng->lanes)
    if (ln.used) HIP_TRY(hipStreamWaitEvent(st, ln.done, 0));
  return CS_OK;
}

int cs_trim_batch(cs_engine *eng, uint32_t slot, const cs_reads *r1, const cs_reads *r2, uint32_t n_reads,
                  uint32_t stride) {
  if (!eng || !r1) return fail(CS_ERR_ARG, "null engine or reads");
  if (slot >= eng->slots.size()) return fail(CS_ERR_ARG, "slot %u out of range", slot);
  if (n_reads > eng->max_reads || stride > eng->max_stride)
    return fail(CS_ERR_ARG, "batch %u x %u exceeds slot capacity %u x %u", n_reads, stride, eng->max_reads, eng->max_stride);
  if ((r2 != nullptr) != eng->paired) return fail(CS_ERR_ARG, "plan is %s-end", eng->paired ? "paired" : "single");
  Slot &s = eng->slots[slot];
  if (s.busy) return fail(CS_ERR_STATE, "slot %u still in flight: call cs_sync first", slot);
  HIP_TRY(hipSetDevice(eng->device));
  const cs_reads *rr[2] = {r1, r2};
  cs_reads dev[2];
  const size_t bytes = (size_t)n_reads * stride;
  for (int m = 0; m < (r2 ? 2 : 1); ++m) {
    if (!rr[m]->seq || !rr[m]->qual || !rr[m]->len || !rr[m]->out) return fail(CS_ERR_ARG, "null array in mate %d", m + 1);
    HIP_TRY(hipMemcpyAsync(s.d_seq[m], rr[m]->seq, bytes, hipMemcpyHostToDevice, eng->stream));
    HIP_TRY(hipMemcpyAsync(s.d_qual[m], rr[m]->qual, bytes, hipMemcpyHostToDevice, eng->stream));
    HIP_TRY(hipMemcpyAsync(s.d_len[m], rr[m]->len, (size_t)n_reads * sizeof(uint16_t), hipMemcpyHostToDevice, eng->stream));
    dev[m].seq = s.d_seq[m];
    dev[m].qual = s.d_qual[m];
    dev[m].len = s.d_len[m];
    dev[m].out = s.d_out[m];
    dev[m].cap2 = (m == 0 && rr[m]->cap2) ? s.d_cap2 : nullptr;
    dev[m].bc = rr[m]->bc ? s.d_bc[m] : nullptr;
  }
  // H2D and scan kernel on the engine stream, resolve kernel and D2H on the resolve stream: the next slot's
  // copies and scan kernel overlap this slot's resolve kernel and write-back
  hipStream_t rs = eng->resolve_stream;
  int rc = launch(eng, eng->stream, rs, &dev[0], r2 ? &dev[1] : nullptr, n_reads, stride, false);
  if (rc) return rc;
  for (int m = 0; m < (r2 ? 2 : 1); ++m) {
    HIP_TRY(hipMemcpyAsync(rr[m]->out, s.d_out[m], (size_t)n_reads * sizeof(cs_result), hipMemcpyDeviceToHost, rs));
    if (m == 0 && rr[m]->cap2)
      HIP_TRY(hipMemcpyAsync(rr[m]->cap2, s.d_cap2, (size_t)n_reads * sizeof(cs_cap2), hipMemcpyDeviceToHost, rs));
    if (rr[m]->bc) HIP_TRY(hipMemcpyAsync(rr[m]->bc, s.d_bc[m], (size_t)n_reads, hipMemcpyDeviceToHost, rs));
  }
  HIP_TRY(hipEventRecord(s.done, rs));
  s.busy = true;
  return CS_OK;
}

int cs_sync(cs_engine *eng, uint32_t slot) {
  if (!eng) return fail(CS_ERR_ARG, "null engine");
  if (slot >= eng->slots.size()) return fail(CS_ERR_ARG, "slot %u out of range", slot);
  Slot &s = eng->slots[slot];
  if (!s.busy) return CS_OK;
  HIP_TRY(hipSetDevice(eng->device));
  HIP_TRY(hipEventSynchronize(s.done));
  s.busy = false;
  return CS_OK;
}

int cs_stats_fetch(cs_engine *eng, cs_stats stats[2], int reset) {
  if (!eng || !stats) return fail(CS_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(eng->device));
  // behind every launch this engine has issued, on whatever stream
  for (Lane &ln : eng->lanes)
    if (ln.used) HIP_TRY(hipStreamWaitEvent(eng->stream, ln.done, 0));
  HIP_TRY(hipMemcpyAsync(stats, eng->d_stats, 2 * sizeof(cs_stats), hipMemcpyDeviceToHost, eng->stream));
  if (reset) HIP_TRY(hipMemsetAsync(eng->d_stats, 0, 2 * sizeof(cs_stats), eng->stream));
  HIP_TRY(hipStreamSynchronize(eng->stream));
  return CS_OK;
}

int cs_last_kernel_ms(cs_engine *eng, float *ms) {
  if (!eng || !ms) return fail(CS_ERR_ARG, "null argument");
  float split[2];
  int rc = cs_last_kernel_split_ms(eng, split);
  if (rc) return rc;
  *ms = split[0] + split[1];
  return CS_OK;
}

int cs_last_kernel_split_ms(cs_engine *eng, float ms[2]) {
  if (!eng || !ms) return fail(CS_ERR_ARG, "null argument");
  const Lane *ln = eng->last_lane;
  if (!ln || !ln->timed) return fail(CS_ERR_STATE, "no timed launch yet");
  HIP_TRY(hipSetDevice(eng->device));
  HIP_TRY(hipEventSynchronize(ln->ev_stop));
  HIP_TRY(hipEventElapsedTime(&ms[0], ln->ev_start, ln->ev_mid));
  HIP_TRY(hipEventElapsedTime(&ms[1], ln->ev_mid, ln->ev_stop));
  return CS_OK;
}

int cs_kernel_time_totals(cs_engine *eng, uint32_t *calls, float ms[2], int reset) {
  if (!eng || !calls || !ms) return fail(CS_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(eng->device));
  for (Lane &ln : eng->lanes) {
    int rc = harvest(eng, ln);
    if (rc) return rc;
  }
  *calls = eng->timed_calls;
  ms[0] = (float)eng->kernel_ms_total[0];
  ms[1] = (float)eng->kernel_ms_total[1];
  if (reset) {
    eng->timed_calls = 0;
    eng->kernel_ms_total[0] = eng->kernel_ms_total[1] = 0;
  }
  return CS_OK;
}

// Page-locked host memory, two ways.  hipHostMalloc locks 4 KB pages at ~7.5 GB/s -- a quarter of a second per gigabyte,
// which a short command-line run feels (tools/cold_runs.py: 8 M pairs of plain text 1.27 -> 0.90 s); an anonymous
// mapping on transparent huge pages, touched and then registered, gets there at 17 GB/s (512 times fewer pages to
// pin).  Copies from / to it run at the same 53 GB/s (tier T, three alternating runs of each kind on one box).
// cs_alloc_pinned_huge is what the Python layer's buffer arena uses; cs_alloc_pinned stays what it was.
namespace {
std::mutex g_pin_mutex;
std::map<void *, std::pair<void *, size_t>> g_pin_maps;  // what cs_alloc_pinned returned -> (mapping, its length)

bool huge_pages_usable() {
  const char *env = getenv("CUTSEQ_PINNED_THP");
  if (env && atoi(env) == 0) return false;
  FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  if (!f) return false;
  char line[128] = {0};
  const bool got = fgets(line, sizeof line, f) != nullptr;
  fclose(f);
  return got && !strstr(line, "[never]");
}
}  // namespace

void *cs_alloc_pinned_huge(size_t bytes) {
  static const bool huge = huge_pages_usable();
  constexpr size_t kHuge = (size_t)2 << 20;
  if (huge && bytes >= 2 * kHuge) {
    const size_t use = (bytes + kHuge - 1) & ~(kHuge - 1), len = use + kHuge;
    void *base = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base != MAP_FAILED) {
      void *p = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(base) + kHuge - 1) & ~(uintptr_t)(kHuge - 1));
      (void)madvise(p, use, MADV_HUGEPAGE);
      memset(p, 0, use);  // first touch, two megabytes at a time
      if (hipHostRegister(p, use, hipHostRegisterPortable) == hipSuccess) {
        std::lock_guard<std::mutex> lock(g_pin_mutex);
        g_pin_maps[p] = {base, len};
        return p;
      }
      (void)hipGetLastError();
      munmap(base, len);
    }
  }
  return cs_alloc_pinned(bytes);
}

void *cs_alloc_pinned(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {  // usable by every GPU of the node
    fail(CS_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
    return nullptr;
  }
  return p;
}
void cs_free_pinned(void *p) {
  if (!p) return;
  std::pair<void *, size_t> map{nullptr, 0};
  {
    std::lock_guard<std::mutex> lock(g_pin_mutex);
    auto it = g_pin_maps.find(p);
    if (it != g_pin_maps.end()) {
      map = it->second;
      g_pin_maps.erase(it);
    }
  }
  if (map.first) {
    (void)hipHostUnregister(p);
    munmap(map.first, map.second);
  } else {
    (void)hipHostFree(p);
  }
}
void *cs_alloc_device(int device, size_t bytes) {
  void *p = nullptr;
  if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) {
    fail(CS_ERR_NOMEM, "hipMalloc(%zu) on device %d failed", bytes, device);
    return nullptr;
  }
  return p;
}
void cs_free_device(int device, void *p) {
  if (p && hipSetDevice(device) == hipSuccess) (void)hipFree(p);
}
int cs_copy_to_device(int device, void *dst, const void *src, size_t bytes) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return CS_OK;
}
int cs_copy_to_host(int device, void *dst, const void *src, size_t bytes) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return CS_OK;
}

}  // extern "C"

// =====================================================================================================
// Text path (include/cutseq_hip.h, "text path"): upload -> record index -> rows -> trimming kernels -> output text
// =====================================================================================================

namespace {

struct RouteBlock {  // demultiplexing plans: per route what TextMeta holds for the three ordinary routes
  unsigned long long bytes[cstext::kMaxRoutes][2];
  unsigned long long gz_bytes[cstext::kMaxRoutes][2];
  uint32_t count[cstext::kMaxRoutes];
  uint32_t _pad[2];
};

struct TextSlot {
  uint8_t *d_arena = nullptr;  // the slot's one device allocation
  uint8_t *d_text[2] = {nullptr, nullptr};
  uint32_t *d_nl[2] = {nullptr, nullptr};
  cstext::Rec *d_rec[2] = {nullptr, nullptr};
  uint32_t *d_idr[2] = {nullptr, nullptr};
  uint8_t *d_seq[2] = {nullptr, nullptr}, *d_qual[2] = {nullptr, nullptr};
  uint16_t *d_len[2] = {nullptr, nullptr};
  cs_result *d_res[2] = {nullptr, nullptr};
  cs_cap2 *d_cap2 = nullptr;
  uint32_t *d_dst[2] = {nullptr, nullptr};
  uint8_t *d_out[2] = {nullptr, nullptr};
  cslong::LongRec *d_lrec[2] = {nullptr, nullptr};  // reads longer than the rows: their place in the text ...
  cslong::LongRes *d_lres[2] = {nullptr, nullptr};  // ... and their results
  uint32_t *d_long_of[2] = {nullptr, nullptr};      // per record: index into the two, or kNotLong
  uint8_t *d_gzstage[2] = {nullptr, nullptr};       // compressed output: one slot per 32 KB chunk ...
  uint8_t *d_gz[2] = {nullptr, nullptr};            // ... laid out as gzip members
  csdefl::ChunkInfo *d_chunk[2] = {nullptr, nullptr};
  uint32_t *d_chunk_dst[2] = {nullptr, nullptr};
  uint32_t *d_blk = nullptr;              // block sums of the newline passes (per mate) and of the format passes
  unsigned long long *d_totals = nullptr;  // [2] line totals, [6] format column sums
  cstext::TextMeta *d_meta = nullptr;
  cstext::TextMeta *h_meta = nullptr;      // pinned
  uint8_t *d_bc = nullptr;                 // demultiplexing plans: barcode index per record
  RouteBlock *d_routes = nullptr, *h_routes = nullptr;  // ... and the sizes of their 3 + n_bins route streams
  hipEvent_t uploaded = nullptr, formatted = nullptr, fetched = nullptr;
  uint32_t n = 0;
  bool busy = false, waited = false;
};

}  // namespace

struct cs_text {
  cs_engine *eng = nullptr;
  cstext::TextParams tp;
  uint64_t max_text = 0, out_cap = 0;
  uint32_t max_records = 0, stride = 0, max_tag = 0;
  uint32_t seg_blocks = 0, fmt_blocks = 0;
  bool needs_cap2 = false;
  bool compress = false;      // the output streams leave the device as gzip members
  bool literal_only = false;  // ... from literal-only blocks (CUTSEQ_GPU_LZ=0: no run / record-name matches)
  uint32_t max_chunks = 0;
  uint32_t n_routes = 3;      // 3 + cs_text_params.n_bins
  hipStream_t h2d = nullptr, d2h = nullptr;
  std::vector<TextSlot> slots;
};

namespace {

__global__ void text_init_meta(cstext::TextMeta *m) {
  if (threadIdx.x == 0) {
    memset(m, 0, sizeof *m);
    m->err = ~0ull;
  }
}

void free_text(cs_text *t) {
  if (!t) return;
  if (t->eng) (void)hipSetDevice(t->eng->device);
  for (TextSlot &s : t->slots) {
    if (s.busy && s.formatted) (void)hipEventSynchronize(s.formatted);
    if (s.d_arena) (void)hipFree(s.d_arena);  // (every device array of the slot is a piece of it)
    if (s.h_meta) (void)hipHostFree(s.h_meta);
    if (s.h_routes) (void)hipHostFree(s.h_routes);
    for (hipEvent_t ev : {s.uploaded, s.formatted, s.fetched})
      if (ev) (void)hipEventDestroy(ev);
  }
  if (t->h2d) (void)hipStreamDestroy(t->h2d);
  if (t->d2h) (void)hipStreamDestroy(t->d2h);
  delete t;
}

}  // namespace

extern "C" {

void cs_text_destroy(cs_text *t) { free_text(t); }

int cs_text_create(cs_engine *eng, const cs_text_params *params, uint32_t n_slots, uint64_t max_text_bytes,
                   uint32_t max_records, uint32_t stride, cs_text **out) {
  if (!out) return fail(CS_ERR_ARG, "out is null");
  *out = nullptr;
  if (!eng || !params) return fail(CS_ERR_ARG, "null engine or params");
  if (!n_slots || !max_records || !max_text_bytes) return fail(CS_ERR_ARG, "slots, records and text bytes must be positive");
  if (stride == 0 || stride % 4 || stride > CS_MAX_STRIDE)
    return fail(CS_ERR_ARG, "stride %u must be a multiple of 4 in [4, %d]", stride, CS_MAX_STRIDE);
  const uint64_t out_cap = max_text_bytes + (uint64_t)max_records * (params->max_tag + 8u) + 64u;
  if (max_text_bytes >= (1ull << 30) || out_cap >= (1ull << 30))
    return fail(CS_ERR_ARG, "a batch is limited to 1 GiB of text per mate (%llu requested)", (unsigned long long)max_text_bytes);
  cs_text *t = new (std::nothrow) cs_text();
  if (!t) return fail(CS_ERR_NOMEM, "out of memory");
  t->eng = eng;
  t->max_text = max_text_bytes;
  t->out_cap = out_cap;
  t->max_records = max_records;
  t->stride = stride;
  t->max_tag = params->max_tag;
  memset(&t->tp, 0, sizeof t->tp);
  t->tp.paired = eng->paired ? 1 : 0;
  t->tp.has_umi = params->has_umi ? 1 : 0;
  t->tp.untrimmed_filter = params->untrimmed_filter ? 1 : 0;
  t->tp.reverse_complement = params->reverse_complement ? 1 : 0;
  t->tp.fasta_out = params->fasta_out ? 1 : 0;
  t->tp.flag_too_short = CS_F_TOO_SHORT;
  t->tp.flag_untrimmed = CS_F_UNTRIMMED;
  const char *const *suf[2] = {params->suffix1, params->suffix2};
  for (int m = 0; m < 2; ++m)
    for (int k = 0; k < 2; ++k) {
      const char *lit = suf[m][k];
      const size_t len = lit ? strlen(lit) : 0;
      if (len > cstext::kSuffixMax) {
        delete t;
        return fail(CS_ERR_ARG, "name suffix literal longer than %u bytes", cstext::kSuffixMax);
      }
      t->tp.suffix_len[m][k] = (uint8_t)len;
      if (len) memcpy(t->tp.suffix[m][k], lit, len);
    }
  // a second capture exists only in single-end chains (cs_cap2)
  t->needs_cap2 = !eng->paired && params->has_umi;
  t->compress = params->compress != 0;
  {
    const char *env = getenv("CUTSEQ_GPU_LZ");
    t->literal_only = env && atoi(env) == 0;
  }
  if (params->n_bins) {
    if (params->n_bins > 255u || eng->demux_mate < 0) {
      delete t;
      return fail(CS_ERR_ARG, "n_bins = %u needs a plan with a demultiplexing op (and at most 255 barcodes)", params->n_bins);
    }
    t->tp.n_bins = (uint16_t)params->n_bins;
    t->tp.demux_mate = (uint8_t)eng->demux_mate;
    t->n_routes = 3u + params->n_bins;
  }
  t->max_chunks = (uint32_t)((out_cap + csdefl::kChunk - 1) / csdefl::kChunk) + t->n_routes;
  t->seg_blocks = (uint32_t)((max_text_bytes + cstext::kSeg - 1) / cstext::kSeg);
  t->fmt_blocks = (max_records + 255u) / 256u;
#define TXT_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      fail(CS_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      free_text(t);                                                                         \
      return CS_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)
  TXT_TRY(hipSetDevice(eng->device));
  if (t->compress) {
    // constants of the CRC-32 combination (zlib's x2n_table; shifts by whole 128-byte slices)
    auto mult = [](uint32_t a, uint32_t b) {
      uint32_t m = 1u << 31, p = 0;
      for (;;) {
        if (a & m) {
          p ^= b;
          if ((a & (m - 1u)) == 0u) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ csdefl::kPoly : b >> 1;
      }
      return p;
    };
    uint32_t x2n[32], shift[256];
    x2n[0] = 1u << 30;
    for (int i = 1; i < 32; ++i) x2n[i] = mult(x2n[i - 1], x2n[i - 1]);
    const uint32_t step = x2n[10];  // x^(2^10) = x^(8 * 128)
    shift[0] = 1u << 31;
    for (int i = 1; i < 256; ++i) shift[i] = mult(shift[i - 1], step);
    TXT_TRY(hipMemcpyToSymbol(HIP_SYMBOL(csdefl::c_x2n), x2n, sizeof x2n));
    TXT_TRY(hipMemcpyToSymbol(HIP_SYMBOL(csdefl::c_shift128), shift, sizeof shift));
  }
  TXT_TRY(hipStreamCreateWithFlags(&t->h2d, hipStreamNonBlocking));
  TXT_TRY(hipStreamCreateWithFlags(&t->d2h, hipStreamNonBlocking));
  t->slots.resize(n_slots);
  const int mates = eng->paired ? 2 : 1;
  const size_t rows = (size_t)max_records * stride;
  for (TextSlot &s : t->slots) {
    // ONE device allocation per slot, carved up (a hundred separate hipMalloc calls cost a short run more than its
    // kernels).  The part in front is zeroed -- a batch that is rejected half-way leaves some of those arrays unwritten,
    // and whatever a later kernel of that batch still reads through them must stay inside the allocations -- the
    // "not a long read" markers behind it are set to all ones.
    size_t need = 0;
    auto reserve = [&](size_t bytes) {
      const size_t at = need;
      need += (bytes + 255) & ~(size_t)255;
      return at;
    };
    size_t o_nl[2], o_rec[2], o_idr[2], o_long_of[2], o_text[2], o_seq[2], o_qual[2], o_len[2], o_res[2], o_dst[2], o_out[2],
        o_lrec[2], o_lres[2], o_gzstage[2] = {0, 0}, o_gz[2] = {0, 0}, o_chunk[2] = {0, 0}, o_chunk_dst[2] = {0, 0};
    for (int m = 0; m < mates; ++m) {
      o_nl[m] = reserve(((size_t)max_records * 4 + 8) * sizeof(uint32_t));
      o_rec[m] = reserve((size_t)max_records * sizeof(cstext::Rec));
      o_idr[m] = reserve((size_t)max_records * sizeof(uint32_t));
    }
    const size_t zero_bytes = need;
    for (int m = 0; m < mates; ++m) o_long_of[m] = reserve((size_t)max_records * sizeof(uint32_t));
    const size_t ones_bytes = need - zero_bytes;
    for (int m = 0; m < mates; ++m) {
      o_text[m] = reserve(max_text_bytes + 64);
      o_seq[m] = reserve(rows);
      o_qual[m] = reserve(rows);
      o_len[m] = reserve((size_t)max_records * sizeof(uint16_t));
      o_res[m] = reserve((size_t)max_records * sizeof(cs_result));
      o_dst[m] = reserve((size_t)max_records * sizeof(uint32_t));
      o_out[m] = reserve(out_cap);
      o_lrec[m] = reserve((size_t)max_records * sizeof(cslong::LongRec));
      o_lres[m] = reserve((size_t)max_records * sizeof(cslong::LongRes));
      if (t->compress) {
        o_gzstage[m] = reserve((size_t)t->max_chunks * csdefl::kSlot);
        o_gz[m] = reserve((size_t)t->max_chunks * csdefl::kSlot + 64);
        o_chunk[m] = reserve((size_t)t->max_chunks * sizeof(csdefl::ChunkInfo));
        o_chunk_dst[m] = reserve((size_t)t->max_chunks * sizeof(uint32_t));
      }
    }
    const size_t o_cap2 = t->needs_cap2 ? reserve((size_t)max_records * sizeof(cs_cap2)) : 0;
    const size_t o_blk = reserve(((size_t)2 * (t->seg_blocks + 1) + (size_t)2 * t->n_routes * (t->fmt_blocks + 1)) * sizeof(uint32_t));
    const size_t o_totals = reserve((2 + 2 * (size_t)t->n_routes) * sizeof(unsigned long long));
    const size_t o_meta = reserve(sizeof(cstext::TextMeta));
    const size_t o_bc = t->tp.n_bins ? reserve(max_records) : 0;
    const size_t o_routes = t->tp.n_bins ? reserve(sizeof(RouteBlock)) : 0;
    TXT_TRY(hipMalloc(&s.d_arena, need));
    uint8_t *base = s.d_arena;
    TXT_TRY(hipMemsetAsync(base, 0, zero_bytes, t->h2d));
    TXT_TRY(hipMemsetAsync(base + zero_bytes, 0xff, ones_bytes, t->h2d));
    for (int m = 0; m < mates; ++m) {
      s.d_nl[m] = reinterpret_cast<uint32_t *>(base + o_nl[m]);
      s.d_rec[m] = reinterpret_cast<cstext::Rec *>(base + o_rec[m]);
      s.d_idr[m] = reinterpret_cast<uint32_t *>(base + o_idr[m]);
      s.d_long_of[m] = reinterpret_cast<uint32_t *>(base + o_long_of[m]);
      s.d_text[m] = base + o_text[m];
      s.d_seq[m] = base + o_seq[m];
      s.d_qual[m] = base + o_qual[m];
      s.d_len[m] = reinterpret_cast<uint16_t *>(base + o_len[m]);
      s.d_res[m] = reinterpret_cast<cs_result *>(base + o_res[m]);
      s.d_dst[m] = reinterpret_cast<uint32_t *>(base + o_dst[m]);
      s.d_out[m] = base + o_out[m];
      s.d_lrec[m] = reinterpret_cast<cslong::LongRec *>(base + o_lrec[m]);
      s.d_lres[m] = reinterpret_cast<cslong::LongRes *>(base + o_lres[m]);
      if (t->compress) {
        s.d_gzstage[m] = base + o_gzstage[m];
        s.d_gz[m] = base + o_gz[m];
        s.d_chunk[m] = reinterpret_cast<csdefl::ChunkInfo *>(base + o_chunk[m]);
        s.d_chunk_dst[m] = reinterpret_cast<uint32_t *>(base + o_chunk_dst[m]);
      }
    }
    if (t->needs_cap2) s.d_cap2 = reinterpret_cast<cs_cap2 *>(base + o_cap2);
    s.d_blk = reinterpret_cast<uint32_t *>(base + o_blk);
    s.d_totals = reinterpret_cast<unsigned long long *>(base + o_totals);
    s.d_meta = reinterpret_cast<cstext::TextMeta *>(base + o_meta);
    if (t->tp.n_bins) {
      s.d_bc = base + o_bc;
      s.d_routes = reinterpret_cast<RouteBlock *>(base + o_routes);
      TXT_TRY(hipHostMalloc(&s.h_routes, sizeof(RouteBlock), hipHostMallocPortable));
    }
    TXT_TRY(hipHostMalloc(&s.h_meta, sizeof(cstext::TextMeta), hipHostMallocPortable));
    TXT_TRY(hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming));
    TXT_TRY(hipEventCreateWithFlags(&s.formatted, hipEventDisableTiming));
    TXT_TRY(hipEventCreateWithFlags(&s.fetched, hipEventDisableTiming));
  }
  // The memsets above must be through before a slot's first upload (same stream: they are) and before a kernel on one
  // of the engine's other streams reads the arrays (a memset still in flight there was once seen as a flaky
  // line-count error on a shared GPU): wait for them here.
  TXT_TRY(hipStreamSynchronize(t->h2d));
#undef TXT_TRY
  *out = t;
  return CS_OK;
}

int cs_text_submit(cs_text *t, uint32_t slot, const void *text1, uint64_t bytes1, const void *text2, uint64_t bytes2,
                   uint32_t n_records) {
  if (!t || !text1) return fail(CS_ERR_ARG, "null text engine or text");
  if (slot >= t->slots.size()) return fail(CS_ERR_ARG, "slot %u out of range", slot);
  cs_engine *eng = t->eng;
  if ((text2 != nullptr) != eng->paired) return fail(CS_ERR_ARG, "plan is %s-end", eng->paired ? "paired" : "single");
  if (bytes1 > t->max_text || bytes2 > t->max_text || n_records > t->max_records)
    return fail(CS_ERR_ARG, "batch of %u records, %llu / %llu bytes exceeds the slot (%u records, %llu bytes)", n_records,
                (unsigned long long)bytes1, (unsigned long long)bytes2, t->max_records, (unsigned long long)t->max_text);
  TextSlot &s = t->slots[slot];
  if (s.busy) return fail(CS_ERR_STATE, "slot %u still in flight: cs_text_wait + cs_text_fetch first", slot);
  HIP_TRY(hipSetDevice(eng->device));
  const int mates = eng->paired ? 2 : 1;
  const void *src[2] = {text1, text2};
  const uint64_t bytes[2] = {bytes1, bytes2};
  for (int m = 0; m < mates; ++m)
    if (n_records && !bytes[m]) return fail(CS_ERR_ARG, "mate %d: %u records in 0 bytes of text", m + 1, n_records);
  s.n = n_records;
  s.waited = false;
  hipStream_t st = eng->stream, rs = eng->resolve_stream;
  for (int m = 0; m < mates; ++m)
    if (bytes[m]) HIP_TRY(hipMemcpyAsync(s.d_text[m], src[m], bytes[m], hipMemcpyHostToDevice, t->h2d));
  HIP_TRY(hipEventRecord(s.uploaded, t->h2d));
  HIP_TRY(hipStreamWaitEvent(st, s.uploaded, 0));
  hipLaunchKernelGGL(text_init_meta, dim3(1), dim3(64), 0, st, s.d_meta);
  uint32_t *blk_nl[2] = {s.d_blk, s.d_blk + t->seg_blocks + 1};
  uint32_t *blk_fmt = s.d_blk + 2 * (t->seg_blocks + 1);
  if (n_records) {
    for (int m = 0; m < mates; ++m) {
      const uint32_t nb = (uint32_t)((bytes[m] + cstext::kSeg - 1) / cstext::kSeg);
      hipLaunchKernelGGL(cstext::text_count_nl, dim3(nb), dim3(cstext::kSegThreads), 0, st, s.d_text[m], (uint32_t)bytes[m],
                         blk_nl[m]);
      hipLaunchKernelGGL(cstext::text_scan_blocks, dim3(1), dim3(1024), 0, st, blk_nl[m], nb, 1u, s.d_totals + m);
      hipLaunchKernelGGL(cstext::text_write_nl, dim3(nb), dim3(cstext::kSegThreads), 0, st, s.d_text[m], (uint32_t)bytes[m],
                         blk_nl[m], s.d_totals + m, 4u * n_records, s.d_nl[m], s.d_meta, m);
      hipLaunchKernelGGL(cstext::text_parse_records, dim3((n_records + 255u) / 256u), dim3(256), 0, st, s.d_text[m],
                         s.d_nl[m], n_records, t->stride, t->tp, m, s.d_rec[m], s.d_idr[m], s.d_len[m], s.d_lrec[m],
                         s.d_long_of[m], s.d_meta);
    }
    if (mates == 2)
      hipLaunchKernelGGL(cstext::text_check_pairs, dim3((n_records + 255u) / 256u), dim3(256), 0, st, s.d_text[0], s.d_rec[0],
                         s.d_text[1], s.d_rec[1], n_records, s.d_meta);
    for (int m = 0; m < mates; ++m) {
      const unsigned long long dwords = (unsigned long long)n_records * (t->stride / 4) * 2ull;
      const uint32_t grid = (uint32_t)((dwords + 255ull) / 256ull > 65536ull ? 65536ull : (dwords + 255ull) / 256ull);
      hipLaunchKernelGGL(cstext::text_restride, dim3(grid), dim3(256), 0, st, s.d_text[m], s.d_rec[m], n_records, t->stride / 4,
                         reinterpret_cast<uint32_t *>(s.d_seq[m]), reinterpret_cast<uint32_t *>(s.d_qual[m]));
    }
    // reads that do not fit the rows walk their chain in a kernel of their own (usually there are none); in front of
    // the scan kernel, so that "scan kernel done" covers them for the streams that wait on it
    for (int m = 0; m < mates; ++m) {
      cslong::LongArgs la;
      la.text = s.d_text[m];
      la.rec = s.d_lrec[m];
      la.res = s.d_lres[m];
      la.n_long = &s.d_meta->n_long[m];
      la.cap = n_records;
      la.stats = eng->d_stats + (size_t)m * csdev::kStatWords;
      la.plan_slot = (uint32_t)eng->plan_slot;
      la.mate = m;
      la.gate = &s.d_meta->err;
      hipLaunchKernelGGL(cslong::long_kernel, dim3(256), dim3(64), 0, st, la);
    }
    HIP_TRY(hipGetLastError());
    cs_reads rd[2];
    for (int m = 0; m < mates; ++m) {
      rd[m].seq = s.d_seq[m];
      rd[m].qual = s.d_qual[m];
      rd[m].len = s.d_len[m];
      rd[m].out = s.d_res[m];
      rd[m].cap2 = (m == 0) ? s.d_cap2 : nullptr;
      rd[m].bc = (t->tp.n_bins && m == eng->demux_mate) ? s.d_bc : nullptr;
    }
    int rc = launch(eng, st, rs, &rd[0], mates == 2 ? &rd[1] : nullptr, n_records, t->stride, false, &s.d_meta->err);
    if (rc) return rc;
    cstext::FormatArgs fa;
    memset(&fa, 0, sizeof fa);
    for (int m = 0; m < mates; ++m) {
      fa.text[m] = s.d_text[m];
      fa.rec[m] = s.d_rec[m];
      fa.idr[m] = s.d_idr[m];
      fa.seq[m] = s.d_seq[m];
      fa.qual[m] = s.d_qual[m];
      fa.res[m] = s.d_res[m];
      fa.dst[m] = s.d_dst[m];
      fa.out[m] = s.d_out[m];
      fa.long_of[m] = s.d_long_of[m];
      fa.lrec[m] = s.d_lrec[m];
      fa.lres[m] = s.d_lres[m];
    }
    fa.cap2 = s.d_cap2;
    fa.n = n_records;
    fa.stride = t->stride;
    fa.blk = blk_fmt;
    fa.totals = s.d_totals + 2;
    fa.meta = s.d_meta;
    const uint32_t fb = (n_records + 255u) / 256u;
    if (t->tp.n_bins) {  // one route per barcode: LDS sums per route instead of a block scan per route
      fa.bc = s.d_bc;
      fa.route_bytes = &s.d_routes->bytes[0][0];
      fa.route_count = s.d_routes->count;
      const uint32_t cols = 2u * t->n_routes;
      HIP_TRY(hipMemsetAsync(s.d_routes, 0, sizeof(RouteBlock), rs));
      hipLaunchKernelGGL(cstext::format_sizes_bins, dim3(fb), dim3(256), 0, rs, fa, t->tp);
      hipLaunchKernelGGL(cstext::text_scan_blocks, dim3(cols < 64u ? cols : 64u), dim3(1024), 0, rs, blk_fmt, fb, cols, fa.totals);
      hipLaunchKernelGGL(cstext::format_offsets_bins, dim3(fb), dim3(256), 0, rs, fa, t->tp);
    } else {
      hipLaunchKernelGGL(cstext::format_sizes, dim3(fb), dim3(256), 0, rs, fa, t->tp);
      hipLaunchKernelGGL(cstext::text_scan_blocks, dim3(1), dim3(1024), 0, rs, blk_fmt, fb, 6u, fa.totals);
      hipLaunchKernelGGL(cstext::format_offsets, dim3(fb), dim3(256), 0, rs, fa, t->tp);
    }
    const unsigned long long items = (unsigned long long)n_records * mates;
    const unsigned long long want = (items * 32ull + 255ull) / 256ull;
    hipLaunchKernelGGL(cstext::format_copy, dim3((uint32_t)(want > 32768ull ? 32768ull : want)), dim3(256), 0, rs, fa, t->tp);
    HIP_TRY(hipGetLastError());
    if (t->compress) {
      for (int m = 0; m < mates; ++m) {
        csdefl::DeflateArgs da;
        da.text = s.d_out[m];
        da.route_bytes = t->tp.n_bins ? &s.d_routes->bytes[0][m] : &s.d_meta->route_bytes[0][m];
        da.n_routes = t->n_routes;
        da.stage = s.d_gzstage[m];
        da.info = s.d_chunk[m];
        da.max_chunks = t->max_chunks;
        da.gate = &s.d_meta->err;
        da.marker = t->tp.fasta_out ? '>' : '@';
        da.lines = t->tp.fasta_out ? 2u : 4u;
        da.literal_only = t->literal_only ? 1u : 0u;
        hipLaunchKernelGGL(csdefl::deflate_chunks, dim3(t->max_chunks), dim3(256), 0, rs, da);
        csdefl::LayoutArgs la;
        la.info = s.d_chunk[m];
        la.route_bytes = da.route_bytes;
        la.n_routes = t->n_routes;
        la.chunk_dst = s.d_chunk_dst[m];
        la.gz = s.d_gz[m];
        la.gz_route_bytes = t->tp.n_bins ? &s.d_routes->gz_bytes[0][m] : &s.d_meta->gz_route_bytes[0][m];
        la.gz_total = &s.d_meta->gz_bytes[m];
        la.gate = da.gate;
        hipLaunchKernelGGL(csdefl::deflate_layout, dim3(1), dim3(256), 0, rs, la);
        csdefl::CompactArgs ca;
        ca.info = s.d_chunk[m];
        ca.chunk_dst = s.d_chunk_dst[m];
        ca.stage = s.d_gzstage[m];
        ca.gz = s.d_gz[m];
        ca.route_bytes = da.route_bytes;
        ca.n_routes = t->n_routes;
        ca.gate = da.gate;
        hipLaunchKernelGGL(csdefl::deflate_compact, dim3(t->max_chunks), dim3(256), 0, rs, ca);
      }
      HIP_TRY(hipGetLastError());
    }
  } else {
    HIP_TRY(hipEventRecord(s.fetched, st));  // order the resolve stream behind the (empty) batch's meta
    HIP_TRY(hipStreamWaitEvent(rs, s.fetched, 0));
  }
  HIP_TRY(hipMemcpyAsync(s.h_meta, s.d_meta, sizeof(cstext::TextMeta), hipMemcpyDeviceToHost, rs));
  if (t->tp.n_bins) {
    if (!n_records) HIP_TRY(hipMemsetAsync(s.d_routes, 0, sizeof(RouteBlock), rs));
    HIP_TRY(hipMemcpyAsync(s.h_routes, s.d_routes, sizeof(RouteBlock), hipMemcpyDeviceToHost, rs));
  }
  HIP_TRY(hipEventRecord(s.formatted, rs));
  s.busy = true;
  return CS_OK;
}

int cs_text_wait(cs_text *t, uint32_t slot, cs_text_result *res) {
  if (!t || !res) return fail(CS_ERR_ARG, "null argument");
  if (slot >= t->slots.size()) return fail(CS_ERR_ARG, "slot %u out of range", slot);
  TextSlot &s = t->slots[slot];
  if (!s.busy) return fail(CS_ERR_STATE, "slot %u holds no batch", slot);
  HIP_TRY(hipSetDevice(t->eng->device));
  HIP_TRY(hipEventSynchronize(s.formatted));
  memset(res, 0, sizeof *res);
  const cstext::TextMeta &m = *s.h_meta;
  res->n_records = s.n;
  res->max_len = m.max_len;
  if (m.err != ~0ull) {
    res->error = (int32_t)(m.err & 0xffu);
    res->error_record = (uint32_t)(m.err >> 8);
  }
  for (int q = 0; q < 3; ++q) {
    res->route_count[q] = m.route_count[q];
    for (int k = 0; k < 2; ++k) res->route_bytes[q][k] = m.route_bytes[q][k];
  }
  res->out_bytes[0] = m.out_bytes[0];
  res->out_bytes[1] = m.out_bytes[1];
  if (t->compress && m.err == ~0ull) {  // what is fetched (and written) are the gzip members
    for (int q = 0; q < 3; ++q)
      for (int k = 0; k < 2; ++k) {
        res->text_bytes[q][k] = m.route_bytes[q][k];
        res->route_bytes[q][k] = m.gz_route_bytes[q][k];
      }
    res->out_bytes[0] = m.gz_bytes[0];
    res->out_bytes[1] = m.gz_bytes[1];
    if (t->tp.n_bins)  // (the gzip sizes of a demultiplexing plan live in the route block: [0] = every barcode together)
      for (int k = 0; k < 2; ++k) {
        const RouteBlock &rb = *s.h_routes;
        res->route_bytes[1][k] = rb.gz_bytes[1][k];
        res->route_bytes[2][k] = rb.gz_bytes[2][k];
        res->route_bytes[0][k] = rb.gz_bytes[0][k];
        for (uint32_t q = 3; q < t->n_routes; ++q) res->route_bytes[0][k] += rb.gz_bytes[q][k];
      }
  }
  res->written_bp[0] = m.written_bp[0];
  res->written_bp[1] = m.written_bp[1];
  res->n_lines[0] = m.n_lines[0];
  res->n_lines[1] = m.n_lines[1];
  res->n_long[0] = m.n_long[0];
  res->n_long[1] = m.n_long[1];
  if (res->error) {  // nothing to fetch: the slot is free again
    res->out_bytes[0] = res->out_bytes[1] = 0;
    s.busy = false;
  }
  s.waited = true;
  return CS_OK;
}

int cs_text_routes(cs_text *t, uint32_t slot, uint64_t *bytes, uint64_t *text_bytes, uint32_t *count) {
  if (!t) return fail(CS_ERR_ARG, "null text engine");
  if (slot >= t->slots.size()) return fail(CS_ERR_ARG, "slot %u out of range", slot);
  TextSlot &s = t->slots[slot];
  if (!s.waited) return fail(CS_ERR_STATE, "slot %u: cs_text_wait first", slot);
  const cstext::TextMeta &m = *s.h_meta;
  for (uint32_t q = 0; q < t->n_routes; ++q)
    for (int k = 0; k < 2; ++k) {
      const unsigned long long raw = t->tp.n_bins ? s.h_routes->bytes[q][k] : m.route_bytes[q][k];
      const unsigned long long gz = t->tp.n_bins ? s.h_routes->gz_bytes[q][k] : m.gz_route_bytes[q][k];
      if (text_bytes) text_bytes[q * 2 + k] = raw;
      if (bytes) bytes[q * 2 + k] = t->compress ? gz : raw;
    }
  if (count)
    for (uint32_t q = 0; q < t->n_routes; ++q) count[q] = t->tp.n_bins ? s.h_routes->count[q] : m.route_count[q];
  return CS_OK;
}

int cs_text_fetch(cs_text *t, uint32_t slot, void *dst1, void *dst2) {
  if (!t) return fail(CS_ERR_ARG, "null text engine");
  if (slot >= t->slots.size()) return fail(CS_ERR_ARG, "slot %u out of range", slot);
  TextSlot &s = t->slots[slot];
  if (!s.busy || !s.waited) return fail(CS_ERR_STATE, "slot %u: cs_text_wait first", slot);
  HIP_TRY(hipSetDevice(t->eng->device));
  void *dst[2] = {dst1, dst2};
  for (int m = 0; m < (t->eng->paired ? 2 : 1); ++m) {
    const size_t bytes = (size_t)(t->compress ? s.h_meta->gz_bytes[m] : s.h_meta->out_bytes[m]);
    if (!bytes) continue;
    if (!dst[m]) return fail(CS_ERR_ARG, "mate %d: %zu bytes of output and no buffer", m + 1, bytes);
    HIP_TRY(hipMemcpyAsync(dst[m], t->compress ? s.d_gz[m] : s.d_out[m], bytes, hipMemcpyDeviceToHost, t->d2h));
  }
  HIP_TRY(hipEventRecord(s.fetched, t->d2h));
  HIP_TRY(hipEventSynchronize(s.fetched));
  s.busy = false;
  return CS_OK;
}

}  // extern "C"
