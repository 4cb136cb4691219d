// hbm_pattern_bench.hip — the CIM step kernel's HBM access pattern with NO simulation work: the measured ceiling
// ("what does gfx950's memory system deliver for exactly these pieces?") that bench.py reports as
// roofline.pattern_ceiling next to the kernel's own fraction.
//
// One 64-lane workgroup per env, sorted launch (full-path envs first), G groups on G streams, the step kernel's LDS
// reservation (=> the same number of resident waves per CU).  Per env-step, as mrx_k_cim_step_tab_obs does for
// global_trade.22p_l0.8 (sizes are command-line parameters; defaults = that plan):
//   fast path  (59 %): 64-byte private header + 7 rows x (<= 46 lanes x 4 B) + 4 plan rows + 3 observation rows, one
//                      wait, then ~20 scalar-sized stores from lane 0 (state words, decision, metrics, hint).
//   full path  (41 %): LDS-DMA in 16-byte pieces of frame (--fw words; round 3: 1372 = 5488 B) + private head (--pwh: 312) +
//                      RNG state (624) + the shared topology table (--ctw: 608, L2-resident), the pending-return ring as
//                      plain 4-byte row loads into registers (--ringw: 471 words), the action words; one wait; then per
//                      tick (1.1 on average) the order-table row + stop-table / discharge-record words, one wait, and the
//                      frame-sized snapshot as non-temporal stores out of LDS; then observation block, decision, metrics,
//                      frame + private head (+ RNG state every third step) as non-temporal stores, the ring as 4-byte rows.
// `--work C` inserts C dependent LDS round trips between the load wait and the stores (a stand-in for the simulation's
// latency chain) so that the effect of the wave lifetime / occupancy on the achieved rate can be read off as well.
//
// Output: one JSON object per configuration (bytes are the pattern's own algorithmic bytes, i.e. what the loop nest above
// requests, not counters).  Build: hipcc -O3 --offload-arch=gfx950 -o tools/hbm_pattern_bench tools/hbm_pattern_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

struct Pat {
  int n_envs, FW, PW, PWH, RINGW, MTW, CTW, S, NTP, T, order_bytes, order_tiled, REC_W, SROW, V, obs_words, work;
  int rotate;     // 1: the live frame lives in ring slot fi % S (no post_step copy: the step's first snapshot is a 3-row patch; the write-back goes to the next slot)
  int mt_load;    // words of the RNG state a full-path step loads (624 = all; by-need loading: ~384)
  int fast_table; // 1: the fast path reads the header + a 256-byte decision table instead of 14 rows of the frame
  int ring_half;  // 1: the pending-return ring is stored as uint16 (two pairs per word)
  int32_t *live, *ring, *ring_fi, *priv, *mt, *rec, *tick, *dec, *obsv;
  uint32_t* stops;
  const int32_t* ctab;
  const void* orders;
  double* obs;
  long long* met;
  uint8_t *done, *hint;
  const int32_t* act;
};
#define FULL_FLAG 0x40000000

typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ void dma16(int32_t* lds_chunk, const int32_t* g) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_chunk, 16, 0, 0);
}
__device__ __forceinline__ void dma_rows(int32_t* lds, const int32_t* g, int words) {
  const int l = lane_id(), n4 = words >> 2;
  for (int c = 0; c < n4; c += 64)
    if (c + l < n4) dma16(lds + c * 4, g + (size_t)(c + l) * 4);
}
__device__ __forceinline__ void store_rows_nt(int32_t* g, const int32_t* lds, int words) {
  const int l = lane_id(), n4 = words >> 2;
  for (int i = l; i < n4; i += 64) __builtin_nontemporal_store(((const v4i*)lds)[i], (v4i*)g + i);
}
__device__ __host__ inline unsigned mixu(unsigned a, unsigned b) {
  unsigned x = a * 0x9E3779B9u + b * 0x85EBCA6Bu + 0xC2B2AE35u;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
  return x;
}
// ticks a full-path step advances: mean 1.115 (92 % one, 6 % two, 1.5 % three, 0.5 % six)
__device__ __host__ inline int ticks_of(unsigned h) {
  const unsigned r = h % 1000u;
  return r < 920 ? 1 : r < 980 ? 2 : r < 995 ? 3 : 6;
}

extern "C" __global__ void __launch_bounds__(64) k_pattern(Pat P, const int32_t* __restrict__ order, int step) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const int e = order[blockIdx.x];
  if (e < 0) return;
  const int env = e & (FULL_FLAG - 1), lane = lane_id();
  int32_t* g_live = P.live + (size_t)env * P.FW;
  int32_t* g_priv = P.priv + (size_t)env * P.PW;
  if (!(e & FULL_FLAG)) {  // ---------------- fast path: one trip, rows into registers
    const int v = lane < P.V ? lane : 0;
    int acc = lane < 16 ? g_priv[lane] : 0;
    if (P.fast_table) {
      acc += g_priv[16 + lane];   // 256-byte table of the tick's pending decisions (behind the header)
    } else {
#pragma unroll
      for (int r = 0; r < 7; r++) acc += g_live[264 + r * P.V + v];
#pragma unroll
      for (int k = 0; k < 4; k++) acc += g_live[P.FW - 256 + lane + 64 * k];
#pragma unroll
      for (int a = 0; a < 3; a++) acc += g_live[264 + (8 + a) * P.V + v];
    }
    acc += P.act[(size_t)env * 4 + (lane & 3)];
    // the words the step changes (lane 0), decision / metrics / hint
    const int x = __builtin_amdgcn_readfirstlane(acc);
    if (lane == 0) {
      g_live[22 + (x & 15)] = x; g_live[264 - 22 + (x & 15)] = x; g_live[264 + P.V + (x & 31)] = x; g_live[264 + 3 * P.V + (x & 31)] = x;
      g_live[P.FW - 200 + (x & 127)] = x;
      g_priv[2] = x; g_priv[3] = x; g_priv[4] = x; g_priv[5] = x; g_priv[6] = x;
      if (P.fast_table) { g_priv[16 + (x & 7) * 8] = x; g_priv[17 + (x & 7) * 8] = x; g_priv[18 + ((x >> 3) & 7) * 8] = x; }
      P.obs[(size_t)env * P.obs_words + (x & 63)] = (double)x;
      P.obsv[(size_t)env * 4] = x; P.obsv[(size_t)env * 4 + 1] = x; P.obsv[(size_t)env * 4 + 2] = x;
      P.met[(size_t)env * 3] = x; P.met[(size_t)env * 3 + 1] = x; P.met[(size_t)env * 3 + 2] = x;
      P.done[env] = 0; P.hint[env] = (uint8_t)(x & 1);
    }
    if (lane < 8) P.dec[(size_t)env * 8 + lane] = x;
    return;
  }
  // ---------------- full path
  int32_t* l_frame = lds;
  int32_t* l_priv = l_frame + P.FW;
  int32_t* l_mt = l_priv + P.PWH;
  int32_t* l_ctab = l_mt + P.MTW;
  int t0r = 0;
  if (P.rotate) {  // the live frame's slot depends on the env's tick: one dependent word first
    t0r = __builtin_amdgcn_readfirstlane(P.tick[env]) & 0x3ff;
    g_live = P.ring + ((size_t)env * P.S + (t0r % P.S)) * P.FW;
  }
  dma_rows(l_frame, g_live, P.FW);
  dma_rows(l_priv, g_priv, P.PWH);
  int ring[12];
  if (P.ring_half) {
    const uint16_t* gr = (const uint16_t*)(g_priv + P.PWH);
#pragma unroll
    for (int r = 0; r < 12; r++) ring[r] = (r * 64 < P.RINGW) ? (int)gr[r * 64 + lane < P.RINGW ? r * 64 + lane : 0] : 0;
  } else {
#pragma unroll
    for (int r = 0; r < 12; r++) ring[r] = (r * 64 < P.RINGW) ? g_priv[P.PWH + (r * 64 + lane < P.RINGW ? r * 64 + lane : 0)] : 0;
  }
  dma_rows(l_mt, P.mt + ((size_t)env * 3 + 1) * P.MTW, P.mt_load);
  dma_rows(l_ctab, P.ctab, P.CTW);
  int a = P.act[(size_t)env * 4 + (lane & 3)];
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  int t = __builtin_amdgcn_readfirstlane(l_priv[0]) & 0x3ff;
  if (t >= P.T - 8) t = 0;
  const int nt = ticks_of(mixu((unsigned)env, (unsigned)step));
  int acc = a;
  for (int i = 0; i < nt; i++) {
    // tick inputs: the order-table row, stop-table entries, discharge records
    const int tt = t + i;
    if (P.order_bytes == 4) {
      const int32_t* row = P.order_tiled ? (const int32_t*)P.orders + (((size_t)(tt >> 6) * P.n_envs + env) * 64 + (tt & 63)) * P.NTP
                                         : (const int32_t*)P.orders + ((size_t)env * P.T + tt) * P.NTP;
#pragma unroll
      for (int b = 0; b < 3; b++) acc += row[b * 64 + lane < P.NTP ? b * 64 + lane : 0];
    } else {
      const uint16_t* row = P.order_tiled ? (const uint16_t*)P.orders + (((size_t)(tt >> 6) * P.n_envs + env) * 64 + (tt & 63)) * P.NTP
                                          : (const uint16_t*)P.orders + ((size_t)env * P.T + tt) * P.NTP;
#pragma unroll
      for (int b = 0; b < 3; b++) acc += row[b * 64 + lane < P.NTP ? b * 64 + lane : 0];
    }
    const uint32_t* srow = P.stops + (size_t)env * P.SROW;
    const int vsel = (int)(mixu((unsigned)env, (unsigned)(step * 7 + i)) % (unsigned)P.V);
    acc += (int)srow[(size_t)vsel * (P.SROW / P.V) + (lane & 3)];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      acc += P.rec[(size_t)env * P.REC_W + ((unsigned)(vsel * 97 + q * 31 + lane) % (unsigned)P.REC_W)];
      acc += (int)srow[(size_t)((vsel + q) % P.V) * (P.SROW / P.V) + (lane & 7)];
    }
    asm volatile("" : "+v"(acc));
    __builtin_amdgcn_s_waitcnt(0);
    // the simulation's latency chain (stand-in): dependent LDS round trips
    for (int w = 0; w < P.work; w++) {
      const int x = l_priv[16 + ((acc + lane) & 255)];
      l_priv[16 + lane] = x + w;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc += x;
    }
    // post_step snapshot (resolution 1: every tick)
    if (P.rotate && i == 0) {   // the slot already holds the frame as loaded: patch the rows the action / post_step changed
      int32_t* slot = P.ring + ((size_t)env * P.S + (tt % P.S)) * P.FW;
      if (lane < 22) { slot[22 + lane] = acc; slot[242 + lane] = acc; slot[220 + lane] = acc; }
      if (lane == 0) { slot[264 + 46 + (acc & 31)] = acc; slot[264 + 3 * 46 + (acc & 31)] = acc; slot[P.FW - 200 + (acc & 127)] = acc; }
    } else {
      store_rows_nt(P.ring + ((size_t)env * P.S + (tt % P.S)) * P.FW, l_frame, P.FW);
    }
    if (lane == 0) P.ring_fi[(size_t)env * P.S + (tt % P.S)] = tt;
  }
  t += nt;
  // outputs
  for (int j = lane; j < P.obs_words; j += 64) P.obs[(size_t)env * P.obs_words + j] = (double)acc;
  if (lane < 8) P.dec[(size_t)env * 8 + lane] = acc;
  if (lane == 0) {
    P.obsv[(size_t)env * 4] = acc; P.obsv[(size_t)env * 4 + 1] = acc; P.obsv[(size_t)env * 4 + 2] = acc;
    P.met[(size_t)env * 3] = acc; P.met[(size_t)env * 3 + 1] = acc; P.met[(size_t)env * 3 + 2] = acc;
    P.done[env] = 0; P.tick[env] = t; P.hint[env] = 1;
    l_priv[0] = t;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  store_rows_nt(P.rotate ? P.ring + ((size_t)env * P.S + ((t0r + nt) % P.S)) * P.FW : g_live, l_frame, P.FW);
  store_rows_nt(g_priv, l_priv, P.PWH);
  if (P.ring_half) {
    uint16_t* gr = (uint16_t*)(g_priv + P.PWH);
#pragma unroll
    for (int r = 0; r < 12; r++)
      if (r * 64 + lane < P.RINGW) gr[r * 64 + lane] = (uint16_t)(ring[r] + acc);
  } else {
#pragma unroll
    for (int r = 0; r < 12; r++)
      if (r * 64 + lane < P.RINGW) g_priv[P.PWH + r * 64 + lane] = ring[r] + acc;
  }
  if (mixu((unsigned)env, (unsigned)step + 77u) % 3u == 0u) store_rows_nt(P.mt + ((size_t)env * 3 + 1) * P.MTW, l_mt, P.MTW);
}


// ---- dispatch / residency probes (no HBM traffic): `mode` 1 = empty workgroups, 2 = only the latency chain (`work` dependent
// LDS round trips for the full-path share of the list).  Every workgroup registers on its CU (HW_ID / XCC_ID) so that the
// peak number of co-resident workgroups per CU can be read back.
extern "C" __global__ void __launch_bounds__(64) k_probe(const int32_t* __restrict__ order, int mode, int work, int32_t* cu_now, int32_t* cu_peak) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const int lane = lane_id();
  const int e = order[blockIdx.x];
  unsigned cu = 0;
  if (cu_now) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    cu = ((xcc & 15u) << 8) | ((hw >> 8) & 0xffu);   // xcc | se, sh, cu
    if (lane == 0) { const int n = atomicAdd(&cu_now[cu], 1) + 1; atomicMax(&cu_peak[cu], n); }
  }
  if (mode == 2 && e >= 0 && (e & FULL_FLAG)) {
    int acc = e;
    lds[16 + lane] = lane;
    for (int w = 0; w < work; w++) {
      const int x = lds[16 + ((acc + lane) & 255)];
      lds[16 + lane] = x + w;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc += x;
    }
    if (acc == 0x7fffffff) lds[0] = acc;
  }
  if (cu_now && lane == 0) atomicSub(&cu_now[cu], 1);
}

// pure gather: each wave reads `row_bytes` contiguous bytes at a random row of a region (TLB / small-piece reference)
extern "C" __global__ void __launch_bounds__(64) k_gather(const int32_t* base, long long n_rows, int row_words, int rows_per_wave, int step, int32_t* sink) {
  const int lane = lane_id();
  int acc = 0;
  for (int r = 0; r < rows_per_wave; r++) {
    const unsigned long long h = ((unsigned long long)mixu(blockIdx.x, step * 131 + r) << 20) ^ mixu(blockIdx.x * 7 + 1, step + r);
    const int32_t* row = base + (size_t)(h % (unsigned long long)n_rows) * row_words;
    for (int w = lane; w < row_words; w += 64) acc += row[w];
  }
  if (acc == 0x7fffffff) sink[0] = acc;
}

struct Args {
  int envs = 5461, streams = 3, launches = 150, warm = 20, lds = 12528, order_bytes = 4, tiled = 0, work = 0, T = 1120;
  double full_frac = 0.41;
  int gather = 0, shuffle = 0, fast_lanes = 0, probe = 0, residency = 0;
  int fw = 1372, pwh = 312, ringw = 471, ctw = 608;
  int rotate = 0, mt_load = 624, fast_table = 0, ring_half = 0;
  const char* json_out = nullptr;
  long long gather_mb = 3900;
  int row_bytes = 640;
};

static double bytes_per_step_full(const Pat& P, double mean_ticks, double mt_store_frac) {
  const double ringb = (P.ring_half ? 2.0 : 4.0) * P.RINGW;
  const double rd = 4.0 * (P.FW + P.PWH + P.mt_load) + ringb + 16 + (P.rotate ? 4 : 0) + mean_ticks * (P.NTP * P.order_bytes + 16 + 4 * 64 * 4 * 2 * 0.25);
  const double snap = P.rotate ? (mean_ticks - 1.0) * (4.0 * P.FW) + 3 * 88 + 12 + 4 * mean_ticks : mean_ticks * (4.0 * P.FW + 4);
  const double wr = snap + 8.0 * P.obs_words + 32 + 12 + 24 + 6 + 4.0 * (P.FW + P.PWH) + ringb + mt_store_frac * 4.0 * P.MTW;
  return rd + wr;
}
static double bytes_per_step_fast(const Pat& P) { return 64 + (P.fast_table ? 256 + 12 : (7 + 3) * 46 * 4 + 4 * 256) + 16 + 5 * 4 + 5 * 4 + 8 + 12 + 24 + 2 + 32; }

int main(int argc, char** argv) {
  Args A;
  for (int i = 1; i < argc; i++) {
    auto is = [&](const char* s) { return !strcmp(argv[i], s) && i + 1 < argc; };
    if (is("--envs")) A.envs = atoi(argv[++i]);
    else if (is("--streams")) A.streams = atoi(argv[++i]);
    else if (is("--launches")) A.launches = atoi(argv[++i]);
    else if (is("--lds")) A.lds = atoi(argv[++i]);
    else if (is("--order-bytes")) A.order_bytes = atoi(argv[++i]);
    else if (is("--tiled")) A.tiled = atoi(argv[++i]);
    else if (is("--work")) A.work = atoi(argv[++i]);
    else if (is("--full-frac")) A.full_frac = atof(argv[++i]);
    else if (is("--gather")) A.gather = atoi(argv[++i]);
    else if (is("--shuffle")) A.shuffle = atoi(argv[++i]);
    else if (is("--probe")) A.probe = atoi(argv[++i]);
    else if (is("--fw")) A.fw = atoi(argv[++i]);
    else if (is("--pwh")) A.pwh = atoi(argv[++i]);
    else if (is("--ringw")) A.ringw = atoi(argv[++i]);
    else if (is("--ctw")) A.ctw = atoi(argv[++i]);
    else if (is("--rotate")) A.rotate = atoi(argv[++i]);
    else if (is("--mt-load")) A.mt_load = atoi(argv[++i]);
    else if (is("--fast-table")) A.fast_table = atoi(argv[++i]);
    else if (is("--ring-half")) A.ring_half = atoi(argv[++i]);
    else if (is("--json-out")) A.json_out = argv[++i];
    else if (is("--residency")) A.residency = atoi(argv[++i]);
    else if (is("--fast-lanes")) A.fast_lanes = atoi(argv[++i]);
    else if (is("--gather-mb")) A.gather_mb = atoll(argv[++i]);
    else if (is("--row-bytes")) A.row_bytes = atoi(argv[++i]);
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  CHECK(hipSetDevice(0));
  if (A.gather) {
    const size_t bytes = (size_t)A.gather_mb << 20;
    int32_t *buf, *sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    const int row_words = A.row_bytes / 4, rows_per_wave = 16, waves = 1 << 16;
    const long long n_rows = (long long)(bytes / (size_t)A.row_bytes);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int s = 0; s < 3; s++) hipLaunchKernelGGL(k_gather, dim3(waves), dim3(64), 0, 0, buf, n_rows, row_words, rows_per_wave, s, sink);
    CHECK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int s = 0; s < reps; s++) hipLaunchKernelGGL(k_gather, dim3(waves), dim3(64), 0, 0, buf, n_rows, row_words, rows_per_wave, 10 + s, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double gb = (double)reps * waves * rows_per_wave * A.row_bytes / 1e9;
    printf("{\"bench\": \"gather\", \"region_mb\": %lld, \"row_bytes\": %d, \"GBps\": %.1f}\n", A.gather_mb, A.row_bytes, gb / (ms * 1e-3));
    return 0;
  }
  Pat base;
  memset(&base, 0, sizeof(base));
  base.n_envs = A.envs; base.FW = A.fw; base.PWH = A.pwh; base.RINGW = A.ringw < 768 ? A.ringw : 768; base.PW = (A.pwh + base.RINGW + 3) / 4 * 4; base.MTW = 624; base.CTW = A.ctw; base.S = 4; base.NTP = 160; base.T = A.T;
  base.order_bytes = A.order_bytes; base.order_tiled = A.tiled; base.REC_W = 4096; base.V = 46; base.SROW = 46 * 208; base.obs_words = 22 * 7;
  base.work = A.work;
  base.rotate = A.rotate; base.mt_load = A.mt_load / 4 * 4; base.fast_table = A.fast_table; base.ring_half = A.ring_half;
  const int G = A.streams, N = A.envs;
  std::vector<Pat> P(G, base);
  std::vector<hipStream_t> st(G);
  const int n_lists = 8;
  std::vector<std::vector<int32_t*>> lists(G, std::vector<int32_t*>(n_lists));
  int32_t* ctab;
  CHECK(hipMalloc(&ctab, 4 * base.CTW));
  CHECK(hipMemset(ctab, 0, 4 * base.CTW));
  double full_steps = 0, fast_steps = 0;
  for (int g = 0; g < G; g++) {
    Pat& p = P[g];
    CHECK(hipStreamCreateWithFlags(&st[g], hipStreamNonBlocking));
    auto alloc = [&](size_t bytes) { void* q; CHECK(hipMalloc(&q, bytes)); CHECK(hipMemset(q, 0, bytes)); return q; };
    p.live = (int32_t*)alloc((size_t)N * p.FW * 4); p.ring = (int32_t*)alloc((size_t)N * p.S * p.FW * 4); p.ring_fi = (int32_t*)alloc((size_t)N * p.S * 4);
    p.priv = (int32_t*)alloc((size_t)N * p.PW * 4); p.mt = (int32_t*)alloc((size_t)N * 3 * p.MTW * 4); p.rec = (int32_t*)alloc((size_t)N * p.REC_W * 4);
    p.tick = (int32_t*)alloc((size_t)N * 4); p.dec = (int32_t*)alloc((size_t)N * 32); p.obsv = (int32_t*)alloc((size_t)N * 16);
    p.stops = (uint32_t*)alloc((size_t)N * p.SROW * 4); p.ctab = ctab;
    const size_t tiles = (size_t)(p.T + 63) / 64;
    p.orders = alloc(p.order_tiled ? tiles * N * 64 * p.NTP * p.order_bytes : (size_t)N * p.T * p.NTP * p.order_bytes);
    p.obs = (double*)alloc((size_t)N * p.obs_words * 8); p.met = (long long*)alloc((size_t)N * 24);
    p.done = (uint8_t*)alloc(N); p.hint = (uint8_t*)alloc(N); p.act = (const int32_t*)alloc((size_t)N * 16);
    // per-env ticks spread like a mid-episode batch (mean 560, +-60)
    std::vector<int32_t> pv((size_t)N * p.PW, 0);
    for (int e = 0; e < N; e++) pv[(size_t)e * p.PW] = 500 + (int)(mixu(e, g) % 120u);
    CHECK(hipMemcpy(p.priv, pv.data(), pv.size() * 4, hipMemcpyHostToDevice));
    for (int k = 0; k < n_lists; k++) {
      std::vector<int32_t> full, fast;
      for (int e = 0; e < N; e++) ((mixu(e * 31 + g, k) % 10000u) < (unsigned)(A.full_frac * 10000) ? full : fast).push_back(e);
      std::vector<int32_t> ord;
      if (A.shuffle) {   // unsorted launch: workgroup b = env b
        for (int e = 0; e < N; e++) ord.push_back((mixu(e * 31 + g, k) % 10000u) < (unsigned)(A.full_frac * 10000) ? (e | FULL_FLAG) : e);
      } else {
        for (int e : full) ord.push_back(e | FULL_FLAG);
        if (A.fast_lanes == 0) for (int e : fast) ord.push_back(e);   // fast_lanes = 1: no fast-path workgroups at all (what would a launch cost without them?)
        while ((int)ord.size() < N) ord.push_back(-1);
      }
      CHECK(hipMalloc(&lists[g][k], 4 * (size_t)N));
      CHECK(hipMemcpy(lists[g][k], ord.data(), 4 * (size_t)N, hipMemcpyHostToDevice));
      if (g == 0) { full_steps += full.size(); fast_steps += fast.size(); }
    }
  }
  full_steps /= n_lists; fast_steps /= n_lists;
  CHECK(hipFuncSetAttribute((const void*)k_pattern, hipFuncAttributeMaxDynamicSharedMemorySize, A.lds));
  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_pattern, 64, (size_t)A.lds));
  int32_t *cu_now = nullptr, *cu_peak = nullptr;
  if (A.residency) {
    CHECK(hipMalloc(&cu_now, 4 * 4096)); CHECK(hipMalloc(&cu_peak, 4 * 4096));
    CHECK(hipMemset(cu_now, 0, 4 * 4096)); CHECK(hipMemset(cu_peak, 0, 4 * 4096));
  }
  if (A.probe) CHECK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, A.lds > 2048 ? A.lds : 2048));
  auto run = [&](int n, int step0) {
    for (int s = 0; s < n; s++)
      for (int g = 0; g < G; g++) {
        if (A.probe) hipLaunchKernelGGL(k_probe, dim3(N), dim3(64), (size_t)(A.lds > 2048 ? A.lds : 2048), st[g], (const int32_t*)lists[g][(s + g) % n_lists], A.probe, A.work, cu_now, cu_peak);
        else hipLaunchKernelGGL(k_pattern, dim3(N), dim3(64), (size_t)A.lds, st[g], P[g], (const int32_t*)lists[g][(s + g) % n_lists], step0 + s);
      }
  };
  run(A.warm, 0);
  CHECK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  run(A.launches, A.warm);
  CHECK(hipDeviceSynchronize());
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double per_launch = full_steps * bytes_per_step_full(base, 1.115, 1.0 / 3.0) + fast_steps * bytes_per_step_fast(base);
  const double gbps = per_launch * G * A.launches / sec / 1e9;
  if (A.residency) {
    std::vector<int32_t> pk(4096);
    CHECK(hipMemcpy(pk.data(), cu_peak, 4 * 4096, hipMemcpyDeviceToHost));
    int used = 0, mx = 0; long long sum = 0;
    for (int v : pk) if (v > 0) { used++; sum += v; if (v > mx) mx = v; }
    fprintf(stderr, "residency: %d CUs seen, peak workgroups per CU: max %d, mean %.2f\n", used, mx, used ? (double)sum / used : 0.0);
  }
  printf("{\"bench\": \"%s\",", A.probe == 1 ? "probe_empty" : A.probe == 2 ? "probe_work" : "pattern");
  printf(" \"envs_per_launch\": %d, \"streams\": %d, \"lds_bytes\": %d, \"waves_per_cu\": %d, \"order_bytes\": %d, \"order_tiled\": %d, "
         "\"work\": %d, \"rotate\": %d, \"mt_load\": %d, \"fast_table\": %d, \"ring_half\": %d, \"shuffle\": %d, \"fast_lanes\": %d, \"full_frac\": %.3f, \"bytes_per_launch\": %.0f, \"us_per_batch_step\": %.2f, \"env_steps_per_s\": %.4g, \"GBps\": %.1f, \"frac_of_8TBps\": %.3f}\n",
         N, G, A.lds, per_cu, A.order_bytes, A.tiled, A.work, A.rotate, base.mt_load, A.fast_table, A.ring_half, A.shuffle, A.fast_lanes, A.full_frac, per_launch, sec / A.launches * 1e6, (double)N * G * A.launches / sec, gbps, gbps / 8000.0);
  if (A.json_out) {
    FILE* f = fopen(A.json_out, "w");
    if (f) {
      fprintf(f, "{\"topology\": \"global_trade.22p_l0.8\", \"envs_per_launch\": %d, \"streams\": %d, \"lds_bytes\": %d, \"waves_per_cu\": %d, \"frame_words\": %d, "
                 "\"priv_head_words\": %d, \"ring_words\": %d, \"table_words\": %d, \"full_frac\": %.3f, \"bytes_per_launch\": %.0f, \"us_per_batch_step\": %.2f, "
                 "\"env_steps_per_s\": %.6g, \"GBps\": %.1f, \"source\": \"tools/hbm_pattern_bench (no simulation work: the memory system's rate for the step kernel's pieces)\"}\n",
              N, G, A.lds, per_cu, base.FW, base.PWH, base.RINGW, base.CTW, A.full_frac, per_launch, sec / A.launches * 1e6, (double)N * G * A.launches / sec, gbps);
      fclose(f);
    }
  }
  return 0;
}
