// cim_engine.hip — gfx950 kernels + the C ABI of include/maro_amd.h.
//
// Launch geometry: ONE environment per 64-lane wavefront, one wavefront per workgroup, so a batch
// of N envs is a grid of N workgroups (N >> 256 CUs fills the chip; LDS use per workgroup bounds
// residency).  Each workgroup touches only its own env's HBM rows (coalesced 16 B/lane copies
// HBM <-> LDS), so there is no inter-workgroup traffic, no atomics and no XCD-affinity concern
// beyond the tiny read-only topology tables that every L2 caches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "wave.h"
#include "cim_prof.h"
#include "cim_device.h"
#include "cim_layout.h"

// ------------------------------------------------------------------------------------------ kernels
// Two builds of the step kernel: mrx_k_cim_step generates the tick's orders itself (any order mode);
// mrx_k_cim_step_tab reads them from the order table drawn at reset (CimParams::pregen) and carries neither the
// generator's code nor its LDS (order RNG state, fp64 scratch).
#include "cim_step_kernels.h"


// Order list of the coming step (CimParams::order / sched): the envs whose step needs the full path (hint != 0: a tick
// will run, or the episode starts) first — those are the long waves, so they start first and the short fast-path steps
// fill the tail of the launch — and among them the ones that will run two ticks or more (hint 2) at the very head (longest
// first; lpt = 0 folds them into the others), then the fast-hinted ones, each class in env order; masked-out envs are left out
// and the rest of the list is -1.  One workgroup: thread t owns envs [t * per, (t + 1) * per), a block-wide exclusive scan of
// the class counts gives its output offsets.  16384 envs: one 16-byte load per thread.
// bit b of the result = byte b of the 16-byte piece has a bit of `m` (a byte mask repeated four times) set
__device__ __forceinline__ unsigned mrx_nz16(uint4 v, unsigned m = 0xffffffffu) {
  auto nib = [m](unsigned w) { w &= m; return ((((w | ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u) * 0x01020408u) >> 24; };
  return nib(v.x) | (nib(v.y) << 4) | (nib(v.z) << 8) | (nib(v.w) << 12);
}

__device__ __forceinline__ void mrx_schedule_block(const uint8_t* __restrict__ hint, const uint8_t* __restrict__ mask, int mask_vec, int n, int per,
                                                   int32_t* __restrict__ order, int32_t* __restrict__ sched, int lpt) {
  __shared__ int s_l[16], s_t[16], s_f[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lo = tid * per, hi = lo + per < n ? lo + per : n;  // per is a multiple of 16; the hint array is padded to whole pieces
  // the three classes of a 16-env piece as bit masks
  auto piece = [&](int p0, unsigned& l, unsigned& t, unsigned& f) {
    const unsigned valid = p0 + 16 <= n ? 0xffffu : ((1u << (n - p0)) - 1u);
    const uint4 hv = *(const uint4*)(hint + p0);
    const unsigned tk = mrx_nz16(hv), lg = lpt ? mrx_nz16(hv, 0x02020202u) : 0u;
    unsigned on = 0xffffu;
    if (mask) {
      if (mask_vec && p0 + 16 <= n) on = mrx_nz16(*(const uint4*)(mask + p0));
      else { on = 0; for (int b = 0; b < 16 && p0 + b < n; b++) on |= mask[p0 + b] ? (1u << b) : 0u; }
    }
    l = lg & on & valid;
    t = tk & ~lg & on & valid;
    f = ~tk & on & valid;
  };
  // pass 1: class counts of this thread's envs (kept as bit masks when the thread owns at most 64 envs)
  int cl = 0, ct = 0, cf = 0;
  unsigned long long bl = 0, bt = 0, bf = 0;
  const bool small = per <= 64;
  for (int p0 = lo; p0 < hi; p0 += 16) {
    unsigned l, t, f;
    piece(p0, l, t, f);
    cl += __builtin_popcount(l); ct += __builtin_popcount(t); cf += __builtin_popcount(f);
    if (small) { bl |= (unsigned long long)l << (p0 - lo); bt |= (unsigned long long)t << (p0 - lo); bf |= (unsigned long long)f << (p0 - lo); }
  }
  // block-wide exclusive scans
  int il = cl, it = ct, jf = cf;
  for (int d = 1; d < 64; d <<= 1) {
    const int w = __shfl_up(il, d, 64), u = __shfl_up(it, d, 64), v = __shfl_up(jf, d, 64);
    if (lane >= d) { il += w; it += u; jf += v; }
  }
  if (lane == 63) { s_l[wid] = il; s_t[wid] = it; s_f[wid] = jf; }
  __syncthreads();
  int base_l = 0, base_t = 0, base_f = 0, tot_l = 0, tot_t = 0, tot_f = 0;
  const int n_waves = (int)(blockDim.x >> 6);   // (any multiple of 64 threads up to 1024)
  for (int w = 0; w < n_waves; w++) {
    if (w < wid) { base_l += s_l[w]; base_t += s_t[w]; base_f += s_f[w]; }
    tot_l += s_l[w]; tot_t += s_t[w]; tot_f += s_f[w];
  }
  int ol = base_l + il - cl, ot = tot_l + base_t + it - ct, of = tot_l + tot_t + base_f + jf - cf;
  // pass 2: write the entries
  if (small) {
    for (unsigned long long m = bl; m; m &= m - 1) order[ol++] = (lo + __builtin_ctzll(m)) | MRX_ORDER_TICK;
    for (unsigned long long m = bt; m; m &= m - 1) order[ot++] = (lo + __builtin_ctzll(m)) | MRX_ORDER_TICK;
    for (unsigned long long m = bf; m; m &= m - 1) order[of++] = lo + __builtin_ctzll(m);
  } else {
    for (int p0 = lo; p0 < hi; p0 += 16) {
      unsigned l, t, f;
      piece(p0, l, t, f);
      for (; l; l &= l - 1) order[ol++] = (p0 + __builtin_ctz(l)) | MRX_ORDER_TICK;
      for (; t; t &= t - 1) order[ot++] = (p0 + __builtin_ctz(t)) | MRX_ORDER_TICK;
      for (; f; f &= f - 1) order[of++] = p0 + __builtin_ctz(f);
    }
  }
  const int n_full = tot_l + tot_t;
  for (int i = n_full + tot_f + tid; i < n; i += (int)blockDim.x) order[i] = -1;
  if (tid < 16) sched[tid] = tid == 0 ? n_full : tid == 1 ? n_full + tot_f : tid >= 8 ? -1 : 0;  // [4..7] = 0, [8..11] = -1: dummies of cim::regs_load
}

extern "C" __global__ void __launch_bounds__(1024)
mrx_k_cim_schedule(const uint8_t* __restrict__ hint, const uint8_t* __restrict__ mask, int mask_vec, int n, int per, int32_t* __restrict__ order,
                   int32_t* __restrict__ sched, int lpt) {
  mrx_schedule_block(hint, mask, mask_vec, n, per, order, sched, lpt);
}

#ifndef MRX_DQN_TILE_DEFAULT
#define MRX_DQN_TILE_DEFAULT 16   /* measured (profiles/r04_collect.md): 16-row tiles are faster at every batch size tried */
#endif
#include "cim_dqn.h"   // (after the scheduler: mrx_k_cim_dqn_bin can carry the schedule block of the coming step)
#include "cim_sampler.h"

struct AttrList { int n; int32_t id[16]; };

extern "C" __global__ void __launch_bounds__(256)
mrx_k_cim_query(CimParams K, int node_type, const int32_t* __restrict__ ticks, int nt, int ticks_per_env,
                const int32_t* __restrict__ nodes, int nn, int nodes_per_env, AttrList al, int row_slots, long long total,
                double* __restrict__ out) {
  // one workgroup per (env, tick) row group: the frame lookup (ring slot / aliased live frame / missing) is resolved
  // once, then the threads stream the nn x row_slots elements
  const long long rt = blockIdx.x;  // env * nt + ti
  const int per_row = nn * row_slots;
  for (int j = threadIdx.x; j < per_row; j += blockDim.x) {
    const int ni = j / row_slots;
    out[rt * per_row + j] = cim::query_elem(K, node_type, ticks, nt, ticks_per_env, nodes, nn, nodes_per_env, al.id, al.n, rt * nn + ni, j - ni * row_slots);
  }
}

// Reference random agent (examples/hello_world/cim/hello.py:22-37 style), counter-based so a CPU
// replay can reproduce it: h = mix(seed[env], step); even -> LOAD h'%(scope.load+1) when load>0,
// else DISCHARGE h'%(scope.discharge+1).  Also counts the valid decisions it answered.
__device__ __forceinline__ unsigned long long mrx_mix64(unsigned long long seed, unsigned long long step) {
  unsigned long long x = seed * 0x9E3779B97F4A7C15ull + step * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// sched_per > 0: workgroup 0 builds the order list of the coming step instead (mrx_schedule_block, all envs unmasked) — the
// list only depends on the previous step's hints, so it rides along with the policy instead of costing a launch of its own
extern "C" __global__ void __launch_bounds__(256)
mrx_k_cim_random_policy(CimParams K, const int32_t* __restrict__ decisions, long long step, int32_t* __restrict__ actions,
                        int32_t* __restrict__ n_actions, unsigned long long* __restrict__ counter, int sched_per) {
  if (sched_per > 0 && blockIdx.x == 0) {
    mrx_schedule_block((const uint8_t*)K.hint, nullptr, 0, K.n_envs, sched_per & 0xffffff, K.order, K.sched, sched_per >> 24);
    return;
  }
  const int env = (int)((blockIdx.x - (sched_per > 0 ? 1 : 0)) * blockDim.x + threadIdx.x);
  bool valid = false;
  if (env < K.n_envs) {
    const int32_t* d = decisions + (size_t)env * 8;
    valid = d[7] == 1;
    int32_t* a = actions + (size_t)env * K.max_actions * 4;
    if (valid) {
      // step < 0: key the draw on the decision itself (tick, vessel) so a captured hipGraph needs no per-step scalar
      const unsigned long long key = step >= 0 ? (unsigned long long)step : (((unsigned long long)(unsigned)d[0] << 8) | (unsigned)d[2]) + 0x100000000ull;
      const unsigned long long x = mrx_mix64((unsigned long long)K.seed[env], key);
      const unsigned long long r = x >> 1;
      const int load = d[3], dis = d[4];
      a[0] = d[2]; a[1] = d[1];
      if ((x & 1ull) == 0 && load > 0) { a[2] = (int)(r % (unsigned long long)(load + 1)); a[3] = MRX_ACTION_LOAD; }
      else { a[2] = (int)(r % (unsigned long long)(dis + 1)); a[3] = MRX_ACTION_DISCHARGE; }
    }
    n_actions[env] = valid ? 1 : 0;
  }
  if (counter) {
    const unsigned long long m = __ballot(valid);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(counter, (unsigned long long)__builtin_popcountll(m));
  }
}

// The batched EnvSampler's per-step cache update (mrx_cim_sampler_record): one wave per env, lanes over the state vector.
// The cache of an env is a ring of `cap` slots (a power of two): `head[e]` numbers the elements the env has appended, element q
// lives in slot q & (cap - 1); `last` holds element numbers (-1: none), `prev_j` the SLOT written by the previous step.
template <class T>
__global__ void __launch_bounds__(64)
mrx_k_cim_sampler_record(int n, int P, int D, int cap, int A, int first, const int32_t* __restrict__ dec, const float* __restrict__ state,
                         const int32_t* __restrict__ choice, const int32_t* __restrict__ acts, int32_t* __restrict__ nact, uint8_t* __restrict__ eoe,
                         const uint8_t* __restrict__ done, long long* __restrict__ head, long long* __restrict__ last, long long* __restrict__ prev_j, uint8_t* __restrict__ prev_active,
                         int32_t* __restrict__ c_tick, long long* __restrict__ c_agent, T* __restrict__ c_state, long long* __restrict__ c_action,
                         int32_t* __restrict__ c_env_action, uint8_t* __restrict__ c_terminal, T* __restrict__ c_next_state, T* __restrict__ c_nas,
                         long long* __restrict__ steps_env) {
  const int e = (int)blockIdx.x, lane = (int)threadIdx.x;
  const bool was_over = eoe[e] != 0;
  const bool over = was_over || (done && done[e] != 0);   // the previous step ended the episode: eoe |= done, folded in here
  if (over && !was_over && lane == 0) eoe[e] = 1;
  const float* st = state + (size_t)e * D;
  const long long ring = (long long)cap - 1;
  if (!first && prev_active[e]) {  // the element of the previous step: its next state is this state, or (episode over) its own
    const long long pj = prev_j[e];
    T* ns = c_next_state + ((size_t)e * cap + pj) * D;
    const T* own = c_state + ((size_t)e * cap + pj) * D;
    for (int d = lane; d < D; d += 64) ns[d] = over ? own[d] : (T)st[d];
  }
  if (over) {
    if (lane == 0) { nact[e] = 0; prev_active[e] = 0; }
    return;
  }
  const long long q = head[e], j = q & ring;
  int agent = dec[(size_t)e * 8 + 1];
  agent = agent < 0 ? 0 : (agent >= P ? P - 1 : agent);
  const long long prev = last[(size_t)e * P + agent], pslot = prev & ring;
  T* cs = c_state + ((size_t)e * cap + j) * D;
  for (int d = lane; d < D; d += 64) cs[d] = (T)st[d];
  if (prev >= 0) {
    T* na = c_nas + ((size_t)e * cap + pslot) * D;
    for (int d = lane; d < D; d += 64) na[d] = (T)st[d];
  }
  if (lane < 4) c_env_action[((size_t)e * cap + j) * 4 + lane] = acts[(size_t)e * A * 4 + lane];
  if (lane == 0) {
    c_tick[(size_t)e * cap + j] = dec[(size_t)e * 8];
    c_agent[(size_t)e * cap + j] = agent;
    c_action[(size_t)e * cap + j] = choice[e];
    c_terminal[(size_t)e * cap + j] = 0;
    if (prev >= 0) c_terminal[(size_t)e * cap + pslot] = 0;
    last[(size_t)e * P + agent] = q;
    head[e] = q + 1;
    prev_j[e] = j;
    prev_active[e] = 1;
    steps_env[e] += 1;   // (per env: a single shared counter would serialise 16384 atomics per step)
  }
}

// The batched EnvSampler's EMISSION (mrx_cim_sampler_emit): the oldest n_emit[r] elements of env rows[r]'s transition ring are
// copied out as compact experience rows and their delayed reward is evaluated on the way (examples/cim/rl/env_sampler.py:65-80:
// sum over the `window` ticks after the decision of decay^k * (ff * fulfillment - sf * shortage) of the deciding port, from the
// per-tick port history mrx_cim_set_port_history keeps: int32 [n][frames][2][P], attributes (fulfillment, shortage); ticks
// beyond the episode contribute zeros like the reference's snapshot padding).  One 256-thread workgroup per env row — an env's
// elements read overlapping pieces of ITS history rows, which then come out of the cache — its four waves take elements in
// turn: lanes over the state vector for the three row copies, lanes over the window for the reward (float64, one reduction).
template <class T>
__global__ void __launch_bounds__(256)
mrx_k_cim_sampler_emit(int P, int D, int cap, int frames, int window, double ff, double sf, const double* __restrict__ decay,
                       const long long* __restrict__ rows, const long long* __restrict__ tail, const long long* __restrict__ n_emit,
                       const long long* __restrict__ out_off, const int32_t* __restrict__ hist, const int32_t* __restrict__ c_tick,
                       const long long* __restrict__ c_agent, const T* __restrict__ c_state, const long long* __restrict__ c_action,
                       const int32_t* __restrict__ c_env_action, const uint8_t* __restrict__ c_terminal, const T* __restrict__ c_next_state,
                       const T* __restrict__ c_nas, T* __restrict__ o_state, long long* __restrict__ o_action, int32_t* __restrict__ o_env_action,
                       float* __restrict__ o_reward, T* __restrict__ o_next_state, T* __restrict__ o_nas, uint8_t* __restrict__ o_terminal,
                       int32_t* __restrict__ o_env_id, int32_t* __restrict__ o_tick, int32_t* __restrict__ o_agent) {
  const int r = (int)blockIdx.x, w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  const long long e = rows ? rows[r] : (long long)r;
  const long long ne = n_emit[r], t0 = tail[r], o0 = out_off[r];
  for (long long j = w; j < ne; j += 4) {
    const long long slot = (t0 + j) & ((long long)cap - 1);
    const size_t ci = (size_t)e * cap + (size_t)slot, oi = (size_t)(o0 + j);
    for (int d = lane; d < D; d += 64) {
      o_state[oi * D + d] = c_state[ci * D + d];
      o_next_state[oi * D + d] = c_next_state[ci * D + d];
      o_nas[oi * D + d] = c_nas[ci * D + d];
    }
    const int tick = c_tick[ci];
    int agent = (int)c_agent[ci];
    agent = agent < 0 ? 0 : (agent >= P ? P - 1 : agent);
    double af = 0.0, as = 0.0;
    for (int k = lane; k < window; k += 64) {
      const int t = tick + 1 + k;
      if (t < frames) {
        const int32_t* h = hist + (((size_t)e * frames + (size_t)t) * 2) * P + agent;
        const double dk = decay[k];
        af += dk * (double)h[0];
        as += dk * (double)h[P];
      }
    }
    for (int off = 32; off; off >>= 1) { af += __shfl_down(af, off, 64); as += __shfl_down(as, off, 64); }
    if (lane < 4) o_env_action[oi * 4 + lane] = c_env_action[ci * 4 + lane];
    if (lane == 0) {
      o_reward[oi] = (float)(ff * af - sf * as);
      o_action[oi] = c_action[ci];
      o_terminal[oi] = c_terminal[ci];
      o_env_id[oi] = (int32_t)e;
      o_tick[oi] = tick;
      o_agent[oi] = agent;
    }
  }
}

// ------------------------------------------------------------------------------------------ C ABI
// Plan-specialised code objects are shared between the engines of a process (one hipModule per (device, image)): the
// groups of a batch — same plan, one engine each — then execute the SAME code addresses, so the instruction cache (64 KB per
// CU pair) holds one copy of the ~40 KB step kernel instead of one per group.
struct SharedModule { hipModule_t mod; int device; size_t hash; size_t bytes; int refs; };
static std::vector<SharedModule> g_modules;
static std::mutex g_modules_mu;
static size_t image_hash(const void* image, size_t bytes) {
  size_t h = 1469598103934665603ull;
  const unsigned char* p = (const unsigned char*)image;
  for (size_t i = 0; i < bytes; i++) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}
static hipError_t shared_module_acquire(const void* image, size_t bytes, int device, hipModule_t* out) {
  const size_t hs = image_hash(image, bytes);
  std::lock_guard<std::mutex> lk(g_modules_mu);
  for (auto& m : g_modules)
    if (m.device == device && m.hash == hs && m.bytes == bytes) { m.refs++; *out = m.mod; return hipSuccess; }
  hipModule_t mod = nullptr;
  const hipError_t e = hipModuleLoadData(&mod, image);
  if (e != hipSuccess) return e;
  g_modules.push_back({mod, device, hs, bytes, 1});
  *out = mod;
  return hipSuccess;
}
// drops one reference; the last one drains the device (kernels of the module may still be running) and unloads
static void shared_module_release(hipModule_t mod) {
  std::lock_guard<std::mutex> lk(g_modules_mu);
  for (size_t i = 0; i < g_modules.size(); i++) {
    if (g_modules[i].mod != mod) continue;
    if (--g_modules[i].refs == 0) {
      hipDeviceSynchronize();
      hipModuleUnload(mod);
      g_modules.erase(g_modules.begin() + i);
    }
    return;
  }
}

struct mrx_cim_engine {
  CimHostPlan plan;
  int device;
  CimObs obs;  // fused observation (all zero = off)
  hipModule_t spec_module = nullptr;      // plan-specialised step kernels (mrx_cim_load_step_kernels), else the generic ones
  hipFunction_t spec_fn[4] = {nullptr, nullptr, nullptr, nullptr};  // [pregen * 2 + obs]
  hipFunction_t spec_reset = nullptr, spec_order_table = nullptr;
  hipFunction_t spec_fast = nullptr, spec_loop = nullptr;   // launch form 4 (mrx_k_cim_fast_lanes*, mrx_k_cim_step_loop*)
  int loop_waves = 0;                     // grid of the looped full-path kernel (generic or specialised build in use)
  int step_mode = 0;                      // mrx_cim_set_step_mode (0 = automatic)
  bool order_ready = false;               // the order list of the coming step was built by the policy launch (no mask)
  long long agent_key = -1;               // mrx_cim_set_device_agent: the key of the next answering step (< 0: decision-keyed)
  int lpt = 1;                            // longest-first inside the full-path class of the sorted launch (MRX_CIM_LPT=0: off)
  void* order_stream = nullptr;           // ... on this stream: a step issued on another stream is not ordered behind that launch
  // Kernels of a module may still be queued or running on the caller's stream(s): drain the device before unloading it.
  void unload_spec() {
    if (!spec_module) return;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess) {
      if (cur != device) hipSetDevice(device);
      hipDeviceSynchronize();   // (this engine's launches; other engines sharing the module keep it loaded)
      shared_module_release(spec_module);
      if (cur != device && cur >= 0) hipSetDevice(cur);
    }
    spec_module = nullptr;
    spec_reset = spec_order_table = spec_fast = spec_loop = nullptr;
    loop_waves = 0;
    for (auto& f : spec_fn) f = nullptr;
  }
  ~mrx_cim_engine() { unload_spec(); }
};

static thread_local std::string g_err;
int mrx_set_error_(int code, const std::string& m) { g_err = m; return code; }  // shared with cb_engine.hip
static int set_err(int code, const std::string& m) { return mrx_set_error_(code, m); }
#define HIP_TRY(expr)                                                                                    \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) return set_err(MRX_ERR_HIP, std::string(#expr ": ") + hipGetErrorString(_e)); \
  } while (0)

static int use_device(int device) {
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) return set_err(MRX_ERR_NO_DEVICE, "no HIP device available");
  if (cur != device) HIP_TRY(hipSetDevice(device));
  return MRX_OK;
}

extern "C" {

#ifdef MRX_PROFILE_PHASES
int mrx_prof_read(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_mrx_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mrx_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

// Tools: read (and optionally zero) a `__device__` global of the loaded plan-specialised code object, e.g. g_mrx_prof of a
// code object built with -DMRX_PROFILE_PHASES.
int mrx_cim_read_kernel_global(mrx_handle h, const char* name, void* out, int64_t bytes, int reset);

// ---- one-off self-check behind cim::gen_order_table_fast: of_div(n, d, of_recip(d)) must equal the compiler's n / d bit for bit
// on the operand range the plan guarantees ([2^-40, 2^20], numerators also exactly 0).  32 768 pseudo-random pairs per process.
__global__ void mrx_k_cim_of_selfcheck(int* mismatches) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long x = 0x9E3779B97F4A7C15ull * (t + 1);
  int bad = 0;
  for (int i = 0; i < 8; i++) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    const double d = ldexp(1.0 + (double)(x >> 12) * (1.0 / 4503599627370496.0), (int)(x % 60) - 40);        // [2^-40, 2^20)
    unsigned long long y = x * 0xD6E8FEB86659FD93ull;
    y ^= y >> 32;
    double n = d * ((double)((y >> 11) + 1) * (1.0 / 9007199254740992.0));                                      // (0, d]
    if (n < 9.094947017729282e-13) n = (y & 1) ? 9.094947017729282e-13 : 0.0;
    const double a = cim::of_div(n, d, cim::of_recip(d));
    volatile double vn = n, vd = d;
    const double b = vn / vd;
    bad += __double_as_longlong(a) != __double_as_longlong(b);
  }
  if (bad) atomicAdd(mismatches, bad);
}

static int order_fast_device_veto() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return 0;   // (planning on a box without a GPU: nothing to check against)
  int* d = nullptr;
  int h = 0;
  if (hipMalloc((void**)&d, sizeof(int)) != hipSuccess) return 0;
  bool ran = hipMemset(d, 0, sizeof(int)) == hipSuccess;
  if (ran) {
    hipLaunchKernelGGL(mrx_k_cim_of_selfcheck, dim3(64), dim3(64), 0, 0, d);
    ran = hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
  }
  hipFree(d);
  if (ran && h) fprintf(stderr, "maro_amd: fp64 division self-check: %d of 32768 quotients differ from the shared-reciprocal form; the branch-free order generator is off\n", h);
  return ran && h ? 1 : 0;
}
static const bool g_order_fast_probe_set = (order_fast_probe() = order_fast_device_veto, true);

const char* mrx_last_error(void) { return g_err.c_str(); }
const char* mrx_version(void) { return "maro_amd 0.1 (gfx950)"; }

int64_t mrx_cim_workspace_bytes(const mrx_cim_topology* topo, const mrx_cim_config* cfg) {
  CimHostPlan pl;
  std::string err;
  int rc = cim_plan(topo, cfg, &pl, &err);
  if (rc != MRX_OK) { set_err(rc, err); return rc; }
  return pl.workspace_bytes;
}

int mrx_cim_create(const mrx_cim_topology* topo, const mrx_cim_config* cfg, void* d_workspace, int64_t workspace_bytes,
                   mrx_handle* out) {
  if (!out) return set_err(MRX_ERR_INVALID_ARG, "out handle is null");
  *out = nullptr;
  mrx_cim_engine* e = new (std::nothrow) mrx_cim_engine();
  if (!e) return set_err(MRX_ERR_INVALID_ARG, "out of host memory");
  std::string err;
  int rc = cim_plan(topo, cfg, &e->plan, &err);
  if (rc != MRX_OK) { delete e; return set_err(rc, err); }
  if (const char* ev = getenv("MRX_CIM_LPT")) e->lpt = atoi(ev) ? 1 : 0;  // experiments
  if (!d_workspace || workspace_bytes < e->plan.workspace_bytes || ((uintptr_t)d_workspace & 255)) {
    delete e;
    return set_err(MRX_ERR_WORKSPACE, "workspace is null, smaller than mrx_cim_workspace_bytes() or not 256-byte aligned");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { delete e; return set_err(MRX_ERR_NO_DEVICE, "no HIP device available"); }
  e->device = cfg->device;
  memset(&e->obs, 0, sizeof(e->obs));
  rc = use_device(e->device);
  if (rc != MRX_OK) { delete e; return rc; }
  cim_plan_bind(&e->plan, d_workspace);
  hipError_t he = hipMemcpy((uint8_t*)d_workspace + e->plan.const_off, e->plan.const_blob.data(), e->plan.const_blob.size(),
                            hipMemcpyHostToDevice);
  if (he != hipSuccess) { delete e; return set_err(MRX_ERR_HIP, std::string("upload of topology tables: ") + hipGetErrorString(he)); }
  const CimParams& K = e->plan.kp;
  if ((size_t)K.lds_words_reset * 4 > 64 * 1024) {
    hipFuncSetAttribute((const void*)mrx_k_cim_reset, hipFuncAttributeMaxDynamicSharedMemorySize, K.lds_words_reset * 4);
    hipFuncSetAttribute((const void*)mrx_k_cim_step, hipFuncAttributeMaxDynamicSharedMemorySize, K.lds_words * 4);
    hipFuncSetAttribute((const void*)mrx_k_cim_step_tab, hipFuncAttributeMaxDynamicSharedMemorySize, K.lds_words * 4);
    hipFuncSetAttribute((const void*)mrx_k_cim_step_obs, hipFuncAttributeMaxDynamicSharedMemorySize, K.lds_words * 4);
    hipFuncSetAttribute((const void*)mrx_k_cim_step_tab_obs, hipFuncAttributeMaxDynamicSharedMemorySize, K.lds_words * 4);
  }
  // Env.__init__ generates data with the topology's own seed (cim_data_generator.py:141-145)
  hipLaunchKernelGGL(mrx_k_cim_reset, dim3(K.n_envs), dim3(64), (size_t)K.lds_words_reset * 4, 0, K, nullptr, nullptr,
                     (long long)topo->seed);
  if (K.pregen && K.orders_stride)  // (real data files: the table is an input, uploaded with the constant tables)
    hipLaunchKernelGGL(mrx_k_cim_order_table, dim3(K.n_envs), dim3(64), (size_t)K.lds_words_gen * 4, 0, K, nullptr, nullptr,
                       (long long)topo->seed);
  he = hipDeviceSynchronize();
  if (he == hipSuccess) he = hipGetLastError();
  if (he != hipSuccess) { delete e; return set_err(MRX_ERR_HIP, std::string("initial reset kernel: ") + hipGetErrorString(he)); }
  *out = e;
  return MRX_OK;
}

int mrx_cim_destroy(mrx_handle h) {
  delete h;
  return MRX_OK;
}

int mrx_cim_get_layout(mrx_handle h, mrx_cim_layout* out) {
  if (!h || !out) return set_err(MRX_ERR_INVALID_ARG, "null handle/out");
  *out = h->plan.layout;
  return MRX_OK;
}

int mrx_cim_reset(mrx_handle h, const int64_t* d_seed_cmd, const uint8_t* d_env_mask, void* stream) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  h->order_ready = false;  // (reset rewrites hints)
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  const CimParams& K = h->plan.kp;
  const bool table = K.pregen && K.orders_stride && d_seed_cmd && !K.data_mode;  // envs that keep their seed keep their order table
  if (h->spec_module) {  // plan-specialised builds of the same two kernels
    CimParams Kc = K;
    const long long* cmd = (const long long*)d_seed_cmd;
    long long dflt = -1;
    void* params[] = {&Kc, &cmd, &d_env_mask, &dflt};
    HIP_TRY(hipModuleLaunchKernel(h->spec_reset, (unsigned)K.n_envs, 1, 1, 64, 1, 1, (unsigned)((size_t)K.lds_words_reset * 4), (hipStream_t)stream, params, nullptr));
    if (table)
      HIP_TRY(hipModuleLaunchKernel(h->spec_order_table, (unsigned)K.n_envs, 1, 1, 64, 1, 1, (unsigned)((size_t)K.lds_words_gen * 4), (hipStream_t)stream, params, nullptr));
    return MRX_OK;
  }
  hipLaunchKernelGGL(mrx_k_cim_reset, dim3(K.n_envs), dim3(64), (size_t)K.lds_words_reset * 4, (hipStream_t)stream, K,
                     (const long long*)d_seed_cmd, d_env_mask, (long long)-1);
  if (table)
    hipLaunchKernelGGL(mrx_k_cim_order_table, dim3(K.n_envs), dim3(64), (size_t)K.lds_words_gen * 4, (hipStream_t)stream, K,
                       (const long long*)d_seed_cmd, d_env_mask, (long long)-1);
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

static int launch_step(mrx_handle h, const int32_t* d_actions, const int32_t* d_n_actions, const int32_t* d_n_answered,
                       const uint8_t* d_env_mask, int32_t* d_decisions, int64_t* d_metrics, uint8_t* d_done, void* stream);

int mrx_cim_step(mrx_handle h, const int32_t* d_actions, const int32_t* d_n_actions, const uint8_t* d_env_mask,
                 int32_t* d_decisions, int64_t* d_metrics, uint8_t* d_done, void* stream) {
  if (h && h->plan.kp.decision_mode != 0) return set_err(MRX_ERR_INVALID_ARG, "engine was created in a Joint decision mode: use mrx_cim_step_joint");
  return launch_step(h, d_actions, d_n_actions, nullptr, d_env_mask, d_decisions, d_metrics, d_done, stream);
}

int mrx_cim_step_joint(mrx_handle h, const int32_t* d_actions, const int32_t* d_n_actions, const int32_t* d_n_answered,
                       const uint8_t* d_env_mask, int32_t* d_decisions, int64_t* d_metrics, uint8_t* d_done, void* stream) {
  if (h && h->plan.kp.decision_mode == 0) return set_err(MRX_ERR_INVALID_ARG, "engine was created in Sequential decision mode: use mrx_cim_step");
  return launch_step(h, d_actions, d_n_actions, d_n_answered, d_env_mask, d_decisions, d_metrics, d_done, stream);
}

// The launch form a step uses: 1 unsorted (workgroup b = env b), 2 sorted by mrx_k_cim_schedule, 4 split (lane-parallel
// fast-path kernel + looped full-path kernel).  0 / unset: the best one available (2).
static int effective_step_mode(mrx_handle h) {
  static const int env_mode = getenv("MRX_CIM_STEP_MODE") ? atoi(getenv("MRX_CIM_STEP_MODE")) : 0;  // experiments
  int m = h->step_mode ? h->step_mode : env_mode;
  if (m < 1 || m > 5) m = 2;  // measured (profiles/r02_*): the sorted launch is the fastest form at 16384 envs per GPU
  if (m == 3) m = 2;  // (3 was the persistent pipelined kernel of round 2: measured slower than the sorted launch, removed in round 3)
  if (m == 4 && h->spec_module && !(h->spec_fast && h->spec_loop)) m = 2;
  if (m == 5 && h->spec_module && !h->spec_fast) m = 2;
  return m;
}

// grid of the looped full-path kernel: the workgroups the device holds at once with this kernel's registers + LDS
static int resident_waves(const void* kern, size_t lds_bytes, int device) {
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64, lds_bytes) != hipSuccess || per_cu <= 0) per_cu = 8;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
  if (getenv("MRX_CIM_LOOP_WAVES_PER_CU")) per_cu = atoi(getenv("MRX_CIM_LOOP_WAVES_PER_CU"));  // experiments
  return per_cu * cus;
}

static int launch_step(mrx_handle h, const int32_t* d_actions, const int32_t* d_n_actions, const int32_t* d_n_answered,
                       const uint8_t* d_env_mask, int32_t* d_decisions, int64_t* d_metrics, uint8_t* d_done, void* stream) {
  if (!h || !d_decisions || !d_metrics || !d_done) return set_err(MRX_ERR_INVALID_ARG, "null handle/output pointer");
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  const CimParams& K = h->plan.kp;
  static const size_t lds_pad = getenv("MRX_DEBUG_LDS_PAD_BYTES") ? (size_t)atoi(getenv("MRX_DEBUG_LDS_PAD_BYTES")) : 0;  // occupancy experiments
  const bool obs = h->obs.np > 0 || h->obs.nv > 0;  // (the _obs kernels are only needed for the fused observation; the retention rows are written by every build)
  const int mode = effective_step_mode(h);
  cim::StepBatch B = {d_actions, d_n_actions, d_n_answered, d_decisions, (long long*)d_metrics, d_done};
  if (h->obs.agent_mode) {   // the device agent's key of THIS step (the launches below copy h->obs)
    h->obs.agent_key = h->agent_key;
    if (h->agent_key >= 0) h->agent_key++;
  }
  // built by mrx_cim_random_policy for exactly this step, on this very stream (a step issued on another stream has no ordering
  // against that launch: it rebuilds the list from the hints with its own schedule kernel)
  const bool have_order = h->order_ready && !d_env_mask && stream == h->order_stream;
  h->order_ready = false;
  if (mode >= 2 && !have_order) {
    const int per = ((K.n_envs + 1023) / 1024 + 15) / 16 * 16;  // envs per thread, whole 16-byte pieces
    hipLaunchKernelGGL(mrx_k_cim_schedule, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const uint8_t*)K.hint, d_env_mask,
                       ((uintptr_t)d_env_mask & 15) ? 0 : 1, K.n_envs, per, K.order, K.sched, h->lpt);
  }
  const int sorted = mode == 5 ? 2 : mode >= 2 ? 1 : 0;
  const size_t lds_bytes = (size_t)K.lds_words * 4 + lds_pad;
  if (mode == 4) {
    // the fast-hinted envs, one per lane, no LDS; then the full-path list on as many workgroups as are resident at once
    const unsigned fast_blocks = (unsigned)((K.n_envs + 63) / 64);
    if (h->spec_module) {
      CimParams Kc = K;
      CimObs Oc = h->obs;
      void* pf[] = {&Kc, &Oc, &B, &d_env_mask};
      HIP_TRY(hipModuleLaunchKernel(h->spec_fast, fast_blocks, 1, 1, 64, 1, 1, 0, (hipStream_t)stream, pf, nullptr));
      void* pl[] = {&Kc, &Oc, &B};
      const int W = h->loop_waves < K.n_envs ? h->loop_waves : K.n_envs;
      HIP_TRY(hipModuleLaunchKernel(h->spec_loop, (unsigned)W, 1, 1, 64, 1, 1, (unsigned)lds_bytes, (hipStream_t)stream, pl, nullptr));
      return MRX_OK;
    }
#define MRX_LAUNCH_SPLIT(SUFFIX)                                                                                                    \
    {                                                                                                                              \
      if (!h->loop_waves) h->loop_waves = resident_waves((const void*)mrx_k_cim_step_loop##SUFFIX, lds_bytes, h->device);                       \
      const int W = h->loop_waves < K.n_envs ? h->loop_waves : K.n_envs;                                                           \
      hipLaunchKernelGGL(mrx_k_cim_fast_lanes##SUFFIX, dim3(fast_blocks), dim3(64), 0, (hipStream_t)stream, K, h->obs, B, d_env_mask); \
      hipLaunchKernelGGL(mrx_k_cim_step_loop##SUFFIX, dim3(W), dim3(64), lds_bytes, (hipStream_t)stream, K, h->obs, B);            \
    }
    if (K.pregen) { if (obs) MRX_LAUNCH_SPLIT(_tab_obs) else MRX_LAUNCH_SPLIT(_tab) }
    else { if (obs) MRX_LAUNCH_SPLIT(_obs) else MRX_LAUNCH_SPLIT() }
#undef MRX_LAUNCH_SPLIT
    HIP_TRY(hipGetLastError());
    return MRX_OK;
  }
  if (mode == 5) {
    // launch form 5: the fast-hinted envs one per lane without LDS (as in form 4; what it cannot handle joins the full-path list),
    // then one workgroup per entry of the full-path list (as in form 2)
    const unsigned fast_blocks = (unsigned)((K.n_envs + 63) / 64);
    if (h->spec_module) {
      CimParams Kc = K;
      CimObs Oc = h->obs;
      void* pf[] = {&Kc, &Oc, &B, &d_env_mask};
      HIP_TRY(hipModuleLaunchKernel(h->spec_fast, fast_blocks, 1, 1, 64, 1, 1, 0, (hipStream_t)stream, pf, nullptr));
    } else if (K.pregen) {
      if (obs) hipLaunchKernelGGL(mrx_k_cim_fast_lanes_tab_obs, dim3(fast_blocks), dim3(64), 0, (hipStream_t)stream, K, h->obs, B, d_env_mask);
      else hipLaunchKernelGGL(mrx_k_cim_fast_lanes_tab, dim3(fast_blocks), dim3(64), 0, (hipStream_t)stream, K, h->obs, B, d_env_mask);
    } else {
      if (obs) hipLaunchKernelGGL(mrx_k_cim_fast_lanes_obs, dim3(fast_blocks), dim3(64), 0, (hipStream_t)stream, K, h->obs, B, d_env_mask);
      else hipLaunchKernelGGL(mrx_k_cim_fast_lanes, dim3(fast_blocks), dim3(64), 0, (hipStream_t)stream, K, h->obs, B, d_env_mask);
    }
  }
  if (h->spec_module) {
    CimParams Kc = K;
    CimObs Oc = h->obs;
    int srt = sorted;
    void* params[] = {&Kc, &Oc, &B, &d_env_mask, &srt};
    const int kw = K.wg_waves > 0 ? K.wg_waves : 1;   // envs per workgroup: one staged copy of the topology tables for all of them
    HIP_TRY(hipModuleLaunchKernel(h->spec_fn[(K.pregen ? 2 : 0) + (obs ? 1 : 0)], (unsigned)((K.n_envs + kw - 1) / kw), 1, 1, (unsigned)(64 * kw), 1, 1,
                                  (unsigned)((K.lean_ok ? ((size_t)kw * K.l_ctab + K.ctab_words + 3) / 4 * 16 : (size_t)K.lds_words * 4) + lds_pad), (hipStream_t)stream, params, nullptr));
    return MRX_OK;
  }
  auto kern = K.pregen ? (obs ? mrx_k_cim_step_tab_obs : mrx_k_cim_step_tab) : (obs ? mrx_k_cim_step_obs : mrx_k_cim_step);
  hipLaunchKernelGGL(kern, dim3(K.n_envs), dim3(64), (size_t)K.lds_words * 4 + lds_pad, (hipStream_t)stream, K, h->obs, B, d_env_mask, sorted);
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

int mrx_cim_set_device_agent(mrx_handle h, int mode, int32_t* d_actions, int32_t* d_n_actions, int32_t* d_counts, int64_t next_key) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  if (mode != 0 && mode != 1) return set_err(MRX_ERR_INVALID_ARG, "device agent mode must be 0 (off) or 1 (random legal)");
  if (mode && (!d_actions || !d_n_actions)) return set_err(MRX_ERR_INVALID_ARG, "null action buffers");
  if (mode && h->plan.kp.decision_mode != 0) return set_err(MRX_ERR_UNSUPPORTED, "the device agent answers Sequential decisions only");
  h->obs.agent_mode = mode;
  h->obs.agent_actions = mode ? d_actions : nullptr;
  h->obs.agent_n_actions = mode ? d_n_actions : nullptr;
  h->obs.agent_count = mode ? d_counts : nullptr;
  h->agent_key = next_key;
  return MRX_OK;
}

int mrx_cim_set_step_mode(mrx_handle h, int mode) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  if (mode < 0 || mode > 5) return set_err(MRX_ERR_INVALID_ARG, "step mode must be 0 (automatic), 1, 2, 4 or 5");
  h->step_mode = mode;
  return effective_step_mode(h);
}

static std::string plan_defines(const CimParams& K, const CimObs& O);
static int make_obs(const int32_t* port_attrs, int n_port_attrs, const int32_t* vessel_attrs, int n_vessel_attrs, CimObs* out);

int mrx_cim_set_observation(mrx_handle h, const int32_t* port_attrs, int n_port_attrs, const int32_t* vessel_attrs, int n_vessel_attrs,
                            double* d_obs_ports, double* d_obs_vessel) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  if (h->plan.kp.decision_mode != 0) return set_err(MRX_ERR_UNSUPPORTED, "the fused observation is defined for Sequential decision mode");
  if ((n_port_attrs > 0 && !d_obs_ports) || (n_vessel_attrs > 0 && !d_obs_vessel)) return set_err(MRX_ERR_INVALID_ARG, "null attribute list / output");
  CimObs o;
  const int rc = make_obs(port_attrs, n_port_attrs, vessel_attrs, n_vessel_attrs, &o);
  if (rc != MRX_OK) return rc;
  o.ports = d_obs_ports; o.vessel = d_obs_vessel;
  o.hist_n = h->obs.hist_n; o.hist_frames = h->obs.hist_frames; o.hist = h->obs.hist;  // (mrx_cim_set_port_history is independent of this call)
  o.agent_mode = h->obs.agent_mode; o.agent_actions = h->obs.agent_actions; o.agent_n_actions = h->obs.agent_n_actions; o.agent_count = h->obs.agent_count;   // (so is mrx_cim_set_device_agent)
  for (int i = 0; i < 4; i++) o.hist_attr[i] = h->obs.hist_attr[i];
  if (h->spec_module && plan_defines(h->plan.kp, o) != plan_defines(h->plan.kp, h->obs)) {
    // the loaded specialised kernels have the previous observation configuration compiled in: back to the generic ones until
    // mrx_cim_load_step_kernels is called with a code object for the new configuration
    h->unload_spec();
  }
  h->obs = o;
  // The fast path only patches the observation block (the cells its action changed): it relies on the env's previous
  // step having written the whole block.  A (re)configured observation is cold, so the next step of EVERY env is sent
  // down the full path, which writes the block in full, whatever the env's state.
  int rc2 = use_device(h->device);
  if (rc2 != MRX_OK) return rc2;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(h->plan.kp.hint, 1, (size_t)h->plan.kp.n_envs));
  h->order_ready = false;
  return MRX_OK;
}

int mrx_cim_set_port_history(mrx_handle h, const int32_t* port_attrs, int n_port_attrs, int32_t* d_hist, int64_t frames) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  if (n_port_attrs < 0 || n_port_attrs > 4) return set_err(MRX_ERR_INVALID_ARG, "at most 4 retained port attributes");
  if (n_port_attrs > 0 && (!port_attrs || !d_hist || frames <= 0 || frames > 0x7fffffff)) return set_err(MRX_ERR_INVALID_ARG, "null attribute list / buffer, or frames out of range");
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());  // (steps may be in flight with the previous configuration)
  h->obs.hist_n = 0; h->obs.hist = nullptr; h->obs.hist_frames = 0;
  for (int i = 0; i < n_port_attrs; i++) {
    if (port_attrs[i] < 0 || port_attrs[i] >= PA_COUNT || port_attrs[i] == PA_TRANSFER_COST)
      return set_err(MRX_ERR_INVALID_ARG, "retained attributes must be integer port attributes");
    h->obs.hist_attr[i] = port_attrs[i];
  }
  if (n_port_attrs > 0) { h->obs.hist_n = n_port_attrs; h->obs.hist = d_hist; h->obs.hist_frames = (int)frames; }
  return MRX_OK;
}

int mrx_cim_random_policy(mrx_handle h, const int32_t* d_decisions, int64_t step, int32_t* d_actions, int32_t* d_n_actions,
                          uint64_t* d_counter, void* stream) {
  if (!h || !d_decisions || !d_actions || !d_n_actions) return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  const CimParams& K = h->plan.kp;
  // with a sorted launch form the order list of the coming step is built by workgroup 0 of this launch (the step that
  // follows on the same stream then needs no schedule kernel of its own, unless it is given an env mask)
  static const bool fuse = !(getenv("MRX_CIM_FUSE_SCHEDULE") && atoi(getenv("MRX_CIM_FUSE_SCHEDULE")) == 0);
  // (256-thread workgroups: a 1024-thread one needs 16 free wave slots on ONE CU at once and waits for them behind the step kernels)
  const int sched_per = (fuse && effective_step_mode(h) >= 2) ? ((K.n_envs + 255) / 256 + 15) / 16 * 16 : 0;
  hipLaunchKernelGGL(mrx_k_cim_random_policy, dim3((K.n_envs + 255) / 256 + (sched_per > 0 ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, K,
                     d_decisions, (long long)step, d_actions, d_n_actions, (unsigned long long*)d_counter, sched_per > 0 ? (sched_per | (h->lpt << 24)) : 0);
  HIP_TRY(hipGetLastError());
  h->order_ready = sched_per > 0;
  h->order_stream = stream;
  return MRX_OK;
}

static int host_attr_slots(const CimParams& K, int node_type, int a) {
  if (node_type == 0) return (a >= 0 && a < PA_COUNT) ? 1 : -1;
  if (node_type == 1) {
    if (a < 0 || a >= VA_COUNT) return -1;
    if (a == VA_PAST_STOP_LIST || a == VA_PAST_STOP_TICK_LIST) return K.past_n;
    if (a == VA_FUTURE_STOP_LIST || a == VA_FUTURE_STOP_TICK_LIST) return K.future_n;
    return 1;
  }
  if (node_type == 2) {
    if (a == MA_FULL_ON_PORTS) return K.P * K.P;
    if (a == MA_FULL_ON_VESSELS || a == MA_VESSEL_PLANS) return K.V * K.P;
  }
  return -1;
}

int mrx_cim_attr_slots(mrx_handle h, int node_type, int attr_id) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  return host_attr_slots(h->plan.kp, node_type, attr_id);
}

int mrx_cim_attr_id(int node_type, const char* name) {
  static const char* P[] = {"capacity", "empty", "full", "on_shipper", "on_consignee", "shortage", "acc_shortage", "booking",
                            "acc_booking", "fulfillment", "acc_fulfillment", "transfer_cost"};
  static const char* V[] = {"capacity", "empty", "full", "remaining_space", "early_discharge", "is_parking", "loc_port_idx",
                            "route_idx", "last_loc_idx", "next_loc_idx", "past_stop_list", "past_stop_tick_list",
                            "future_stop_list", "future_stop_tick_list"};
  static const char* M[] = {"full_on_ports", "full_on_vessels", "vessel_plans"};
  const char** tab = node_type == 0 ? P : node_type == 1 ? V : node_type == 2 ? M : nullptr;
  const int n = node_type == 0 ? PA_COUNT : node_type == 1 ? VA_COUNT : node_type == 2 ? MA_COUNT : 0;
  if (!name) return -1;
  for (int i = 0; i < n; i++) if (!strcmp(tab[i], name)) return i;
  return -1;
}

int mrx_cim_query(mrx_handle h, int node_type, const int32_t* d_ticks, int nt, int ticks_per_env, const int32_t* d_nodes,
                  int nn, int nodes_per_env, const int32_t* attrs, int na, double* d_out, void* stream) {
  if (!h || !d_ticks || !d_nodes || !attrs || !d_out) return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  if (na > 16) return set_err(MRX_ERR_INVALID_ARG, "at most 16 attributes per query");
  if (nt <= 0 || nn <= 0 || na <= 0) return set_err(MRX_ERR_INVALID_ARG, "nt, nn and na must be positive");
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  const CimParams& K = h->plan.kp;
  int row_slots = 0;
  for (int i = 0; i < na; i++) {
    const int s = host_attr_slots(K, node_type, attrs[i]);
    if (s < 0) return set_err(MRX_ERR_INVALID_ARG, "unknown attribute id for this node type");
    row_slots += s;
  }
  const long long total = (long long)K.n_envs * nt * nn * row_slots;
  if (total == 0) return MRX_OK;
  const long long blocks = (long long)K.n_envs * nt;
  AttrList al;
  al.n = na;
  for (int i = 0; i < 16; i++) al.id[i] = i < na ? attrs[i] : 0;
  hipLaunchKernelGGL(mrx_k_cim_query, dim3((unsigned)blocks), dim3(nn * row_slots >= 192 ? 256 : (nn * row_slots >= 96 ? 128 : 64)), 0, (hipStream_t)stream, K, node_type, d_ticks, nt,
                     ticks_per_env, d_nodes, nn, nodes_per_env, al, row_slots, total, d_out);
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

// ---- plan-specialised step kernels (cim_spec.hip)
static std::string plan_defines(const CimParams& K, const CimObs& O) {
  std::string o;
#define X(f) o += std::string("#define MRXC_") + #f + " " + std::to_string((long long)K.f) + "\n";
  MRX_CIM_DIM_FIELDS(X)
#undef X
  // the fused observation's configuration (all zero: off)
  o += "#define MRXC_obs_np " + std::to_string(O.np) + "\n#define MRXC_obs_nv " + std::to_string(O.nv) + "\n#define MRXC_obs_pa_packed " +
       std::to_string(O.pa_packed) + "u\n#define MRXC_obs_i_empty " + std::to_string(O.i_empty) + "\n#define MRXC_obs_i_tc " + std::to_string(O.i_tc) + "\n";
  o += "#define MRXC_obs_va(i) (";
  for (int i = 0; i < 7; i++) o += "(i) == " + std::to_string(i) + " ? " + std::to_string(O.va[i]) + " : ";
  o += std::to_string(O.va[7]) + ")\n";
  return o;
}

// CimObs for a list of attribute ids (device pointers left null); shared by mrx_cim_set_observation and mrx_cim_plan_defines
static int make_obs(const int32_t* port_attrs, int n_port_attrs, const int32_t* vessel_attrs, int n_vessel_attrs, CimObs* out) {
  if (n_port_attrs < 0 || n_port_attrs > 8 || n_vessel_attrs < 0 || n_vessel_attrs > 8) return set_err(MRX_ERR_INVALID_ARG, "at most 8 port and 8 vessel attributes");
  if ((n_port_attrs && !port_attrs) || (n_vessel_attrs && !vessel_attrs)) return set_err(MRX_ERR_INVALID_ARG, "null attribute list");
  CimObs o;
  memset(&o, 0, sizeof(o));
  for (int i = 0; i < n_port_attrs; i++) {
    if (port_attrs[i] < 0 || port_attrs[i] >= PA_COUNT) return set_err(MRX_ERR_INVALID_ARG, "unknown port attribute id");
    o.pa[i] = port_attrs[i];
  }
  for (int i = 0; i < n_vessel_attrs; i++) {
    if (vessel_attrs[i] < 0 || vessel_attrs[i] >= VA_PAST_STOP_LIST) return set_err(MRX_ERR_INVALID_ARG, "vessel attribute must be a single-slot attribute");
    o.va[i] = vessel_attrs[i];
  }
  o.np = n_port_attrs; o.nv = n_vessel_attrs;
  o.i_empty = o.i_tc = -1;
  for (int i = 0; i < n_port_attrs; i++) {
    o.pa_packed |= (unsigned)o.pa[i] << (4 * i);
    if (o.pa[i] == PA_EMPTY) o.i_empty = i;
    if (o.pa[i] == PA_TRANSFER_COST) o.i_tc = i;
  }
  if (!o.np && !o.nv) o.i_empty = o.i_tc = 0;  // "off" is the all-zero struct
  *out = o;
  return MRX_OK;
}

int64_t mrx_cim_plan_defines(const mrx_cim_topology* topo, const mrx_cim_config* cfg, const int32_t* obs_port_attrs, int n_obs_port_attrs,
                             const int32_t* obs_vessel_attrs, int n_obs_vessel_attrs, char* buf, int64_t len) {
  CimHostPlan plan;
  std::string err;
  int rc = cim_plan(topo, cfg, &plan, &err);
  if (rc != MRX_OK) return set_err(rc, err);
  CimObs obs;
  rc = make_obs(obs_port_attrs, n_obs_port_attrs, obs_vessel_attrs, n_obs_vessel_attrs, &obs);
  if (rc != MRX_OK) return rc;
  const std::string d = plan_defines(plan.kp, obs);
  if (buf && len > 0) {
    if ((int64_t)d.size() + 1 > len) return set_err(MRX_ERR_INVALID_ARG, "buffer too small");
    memcpy(buf, d.c_str(), d.size() + 1);
  }
  return (int64_t)d.size() + 1;
}

int mrx_cim_load_step_kernels(mrx_handle h, const void* image, int64_t bytes, const char* defines) {
  if (!h || !image || bytes <= 0 || !defines) return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  if (plan_defines(h->plan.kp, h->obs) != defines)
    return set_err(MRX_ERR_INVALID_ARG, "the code object was built for a different plan or observation (defines differ)");
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  hipModule_t mod = nullptr;
  HIP_TRY(shared_module_acquire(image, (size_t)bytes, h->device, &mod));
  static const char* names[4] = {"mrx_k_cim_step", "mrx_k_cim_step_obs", "mrx_k_cim_step_tab", "mrx_k_cim_step_tab_obs"};
  hipFunction_t fn[4] = {nullptr, nullptr, nullptr, nullptr};
  const int want = (h->plan.kp.pregen ? 2 : 0) + ((h->obs.np > 0 || h->obs.nv > 0) ? 1 : 0);  // the one kernel this configuration launches
  for (int i = want; i <= want; i++) {
    if (hipModuleGetFunction(&fn[i], mod, names[i]) != hipSuccess) {
      shared_module_release(mod);
      return set_err(MRX_ERR_INVALID_ARG, std::string("code object lacks kernel ") + names[i]);
    }
    {
      const CimParams& Kp = h->plan.kp;
      const size_t wg_bytes = Kp.lean_ok ? ((size_t)(Kp.wg_waves > 0 ? Kp.wg_waves : 1) * Kp.l_ctab + Kp.ctab_words + 3) / 4 * 16 : (size_t)Kp.lds_words * 4;
      if (wg_bytes > 64 * 1024) hipFuncSetAttribute((const void*)fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)wg_bytes);
    }
  }
  hipFunction_t f_reset = nullptr, f_table = nullptr;
  if (hipModuleGetFunction(&f_reset, mod, "mrx_k_cim_reset") != hipSuccess || hipModuleGetFunction(&f_table, mod, "mrx_k_cim_order_table") != hipSuccess) {
    shared_module_release(mod);
    return set_err(MRX_ERR_INVALID_ARG, "code object lacks the reset kernels");
  }
  if ((size_t)h->plan.kp.lds_words_reset * 4 > 64 * 1024) hipFuncSetAttribute((const void*)f_reset, hipFuncAttributeMaxDynamicSharedMemorySize, h->plan.kp.lds_words_reset * 4);
  h->unload_spec();  // (drains the device first: kernels of the old module may still be running)
  h->spec_module = mod;
  for (int i = 0; i < 4; i++) h->spec_fn[i] = fn[i];
  h->spec_reset = f_reset;
  h->spec_order_table = f_table;
  {  // launch form 4: the two kernels of this configuration
    static const char* sfx[4] = {"", "_obs", "_tab", "_tab_obs"};
    const std::string nf = std::string("mrx_k_cim_fast_lanes") + sfx[want], nl = std::string("mrx_k_cim_step_loop") + sfx[want];
    hipFunction_t ff = nullptr, fl = nullptr;
    if (hipModuleGetFunction(&ff, mod, nf.c_str()) == hipSuccess && hipModuleGetFunction(&fl, mod, nl.c_str()) == hipSuccess && ff && fl) {
      if ((size_t)h->plan.kp.lds_words * 4 > 64 * 1024) hipFuncSetAttribute((const void*)fl, hipFuncAttributeMaxDynamicSharedMemorySize, h->plan.kp.lds_words * 4);
      int per_cu = 0, cus = 0;
      if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fl, 64, (size_t)h->plan.kp.lds_words * 4) != hipSuccess || per_cu <= 0) per_cu = 8;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus <= 0) cus = 256;
      if (getenv("MRX_CIM_LOOP_WAVES_PER_CU")) per_cu = atoi(getenv("MRX_CIM_LOOP_WAVES_PER_CU"));
      h->spec_fast = ff; h->spec_loop = fl; h->loop_waves = per_cu * cus;
    } else {
      (void)hipGetLastError();
    }
  }
  return MRX_OK;
}

int mrx_cim_read_kernel_global(mrx_handle h, const char* name, void* out, int64_t bytes, int reset) {
  if (!h || !name || !out || bytes <= 0) return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  if (!h->spec_module) return set_err(MRX_ERR_INVALID_ARG, "no plan-specialised code object is loaded");
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  hipDeviceptr_t p = nullptr;
  size_t n = 0;
  HIP_TRY(hipModuleGetGlobal(&p, &n, h->spec_module, name));
  if ((size_t)bytes > n) return set_err(MRX_ERR_INVALID_ARG, "the global is smaller than the requested size");
  HIP_TRY(hipMemcpy(out, (const void*)p, (size_t)bytes, hipMemcpyDeviceToHost));
  if (reset) HIP_TRY(hipMemset((void*)p, 0, (size_t)bytes));
  return MRX_OK;
}

int mrx_cim_sampler_record(int32_t n_envs, int32_t n_ports, int32_t state_dim, int32_t cap, int32_t max_actions, int32_t state_f64, int32_t first,
                           const int32_t* d_decisions, const float* d_state, const int32_t* d_choice, const int32_t* d_actions, int32_t* d_n_actions,
                           uint8_t* d_eoe, const uint8_t* d_done, int64_t* d_count, int64_t* d_last, int64_t* d_prev_j, uint8_t* d_prev_active,
                           int32_t* c_tick, int64_t* c_agent, void* c_state, int64_t* c_action, int32_t* c_env_action, uint8_t* c_terminal,
                           void* c_next_state, void* c_next_agent_state, int64_t* d_interactions, int32_t device, void* stream) {
  if (n_envs <= 0 || n_ports <= 0 || state_dim <= 0 || cap <= 0 || max_actions <= 0) return set_err(MRX_ERR_INVALID_ARG, "dimensions must be positive");
  if (cap & (cap - 1)) return set_err(MRX_ERR_INVALID_ARG, "cap (slots of an env's transition ring) must be a power of two");
  if (!d_decisions || !d_state || !d_choice || !d_actions || !d_n_actions || !d_eoe || !d_count || !d_last || !d_prev_j || !d_prev_active || !c_tick ||
      !c_agent || !c_state || !c_action || !c_env_action || !c_terminal || !c_next_state || !c_next_agent_state || !d_interactions)
    return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  int rc = use_device(device);
  if (rc != MRX_OK) return rc;
#define MRX_REC(T)                                                                                                                               \
  hipLaunchKernelGGL(mrx_k_cim_sampler_record<T>, dim3((unsigned)n_envs), dim3(64), 0, (hipStream_t)stream, (int)n_envs, (int)n_ports, (int)state_dim,  \
                     (int)cap, (int)max_actions, (int)first, d_decisions, d_state, d_choice, d_actions, d_n_actions, d_eoe, d_done, (long long*)d_count,            \
                     (long long*)d_last, (long long*)d_prev_j, d_prev_active, c_tick, (long long*)c_agent, (T*)c_state, (long long*)c_action, c_env_action, \
                     c_terminal, (T*)c_next_state, (T*)c_next_agent_state, (long long*)d_interactions)
  if (state_f64) MRX_REC(double); else MRX_REC(float);
#undef MRX_REC
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

int mrx_cim_sampler_emit(int32_t n_rows, int32_t n_ports, int32_t state_dim, int32_t cap, int32_t frames, int32_t window, int32_t state_f64,
                         double fulfillment_factor, double shortage_factor, const double* d_decay, const int64_t* d_rows, const int64_t* d_tail,
                         const int64_t* d_n_emit, const int64_t* d_out_offset, const int32_t* d_port_history, const int32_t* c_tick,
                         const int64_t* c_agent, const void* c_state, const int64_t* c_action, const int32_t* c_env_action, const uint8_t* c_terminal,
                         const void* c_next_state, const void* c_next_agent_state, void* o_state, int64_t* o_action, int32_t* o_env_action,
                         float* o_reward, void* o_next_state, void* o_next_agent_state, uint8_t* o_terminal, int32_t* o_env_id, int32_t* o_tick,
                         int32_t* o_agent, int32_t device, void* stream) {
  if (n_rows < 0 || n_ports <= 0 || state_dim <= 0 || cap <= 0 || frames <= 0 || window <= 0) return set_err(MRX_ERR_INVALID_ARG, "dimensions must be positive");
  if (cap & (cap - 1)) return set_err(MRX_ERR_INVALID_ARG, "cap (slots of an env's transition ring) must be a power of two");
  if (!d_decay || !d_tail || !d_n_emit || !d_out_offset || !d_port_history || !c_tick || !c_agent || !c_state || !c_action || !c_env_action || !c_terminal ||
      !c_next_state || !c_next_agent_state || !o_state || !o_action || !o_env_action || !o_reward || !o_next_state || !o_next_agent_state || !o_terminal ||
      !o_env_id || !o_tick || !o_agent)
    return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  if (n_rows == 0) return MRX_OK;
  int rc = use_device(device);
  if (rc != MRX_OK) return rc;
#define MRX_EMIT(T)                                                                                                                                   \
  hipLaunchKernelGGL(mrx_k_cim_sampler_emit<T>, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, (int)n_ports, (int)state_dim, (int)cap,      \
                     (int)frames, (int)window, fulfillment_factor, shortage_factor, d_decay, (const long long*)d_rows, (const long long*)d_tail,        \
                     (const long long*)d_n_emit, (const long long*)d_out_offset, d_port_history, c_tick, (const long long*)c_agent, (const T*)c_state,  \
                     (const long long*)c_action, c_env_action, c_terminal, (const T*)c_next_state, (const T*)c_next_agent_state, (T*)o_state,           \
                     (long long*)o_action, o_env_action, o_reward, (T*)o_next_state, (T*)o_next_agent_state, o_terminal, o_env_id, o_tick, o_agent)
  if (state_f64) MRX_EMIT(double); else MRX_EMIT(float);
#undef MRX_EMIT
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

// ---- DQN action selection (cim_dqn.h)
static int dqn_plan(const mrx_cim_dqn_model* m, cim::DqnParams* D) {
  using namespace cim;
  if (!m) return set_err(MRX_ERR_INVALID_ARG, "null model");
  if (m->n_layers < 1 || m->n_layers > MRX_DQN_MAX_LAYERS) return set_err(MRX_ERR_INVALID_ARG, "n_layers out of range");
  if (m->n_actions < 1 || m->n_actions > MRX_DQN_MAX_ACTIONS) return set_err(MRX_ERR_INVALID_ARG, "n_actions out of range");
  if (m->dims[m->n_layers] != m->n_actions + (m->dueling ? 1 : 0)) return set_err(MRX_ERR_INVALID_ARG, "last width must be n_actions (+ 1 when dueling)");
  memset(D, 0, sizeof(*D));
  D->n_layers = m->n_layers;
  D->dueling = m->dueling ? 1 : 0;
  D->state_dim = m->dims[0];
  D->n_actions = m->n_actions;
  D->slope = m->negative_slope;
  D->epsilon = m->epsilon;
  long long off = 0;
  for (int l = 0; l < m->n_layers; l++) {
    if (m->dims[l] < 1 || m->dims[l] > MRX_DQN_MAX_WIDTH || m->dims[l + 1] < 1 || m->dims[l + 1] > MRX_DQN_MAX_WIDTH)
      return set_err(MRX_ERR_UNSUPPORTED, "layer widths must be in 1..256");
    D->kpad[l] = l == 0 ? dq_kpad(m->dims[0]) : D->npad[l - 1];
    D->npad[l] = dq_npad(m->dims[l + 1]);
    D->n_out[l] = m->dims[l + 1];
  }
  // one net: the four waves' fragment streams (cim_dqn.h: each wave's layers one after another), the biases, then DQ_SLOTS
  // fragments of padding — the last two passes of a stream refill the ring from what follows it (nothing consumes that)
  for (int w = 0; w < 4; w++)
    for (int l = 0; l < m->n_layers; l++) {
      D->s_off[w][l] = off;
      off += 256LL * dq_frags(D->kpad[l], D->npad[l], w);
    }
  for (int l = 0; l < m->n_layers; l++) {
    D->b_off[l] = off;
    off += D->npad[l];
  }
  off += 256LL * DQ_SLOTS;
  D->net_floats = off;
  for (int a = 0; a < m->n_actions; a++) D->action_space[a] = m->action_space[a];
  return MRX_OK;
}

int64_t mrx_cim_dqn_net_floats(const mrx_cim_dqn_model* m) {
  cim::DqnParams D;
  const int rc = dqn_plan(m, &D);
  return rc != MRX_OK ? rc : D.net_floats;
}

int mrx_cim_dqn_pack_net(const mrx_cim_dqn_model* m, const float* const* weights, const float* const* biases, float* out) {
  cim::DqnParams D;
  const int rc = dqn_plan(m, &D);
  if (rc != MRX_OK) return rc;
  if (!weights || !biases || !out) return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  memset(out, 0, sizeof(float) * (size_t)D.net_floats);
  for (int l = 0; l < m->n_layers; l++) {
    const int K = m->dims[l], N = m->dims[l + 1];
    for (int k = 0; k < K; k++)
      for (int n = 0; n < N; n++) {
        int w, frag, within;
        cim::dq_w_slot(k, n, D.npad[l], &w, &frag, &within);
        out[D.s_off[w][l] + 256LL * frag + within] = weights[l][(size_t)k * N + n];
      }
    for (int n = 0; n < N; n++) out[D.b_off[l] + n] = biases[l][n];
  }
  return MRX_OK;
}

// rows of one forward workgroup's tile: 32 (two MFMA row tiles) or 16 (one: half the matrix work per workgroup and twice the
// workgroups, i.e. a finer balance over the CUs when a launch has about as many 32-row tiles as the chip has CUs).  MRX_DQN_TILE
// overrides the choice (experiments).
static int dqn_tile_rows(const CimParams& K) {
  static const int env_tile = getenv("MRX_DQN_TILE") ? atoi(getenv("MRX_DQN_TILE")) : 0;
  if (env_tile == 16 || env_tile == 32) return env_tile;
  return MRX_DQN_TILE_DEFAULT;
}
static long long dqn_max_tiles(const CimParams& K, int tile) { return (((long long)K.n_envs + tile - 1) / tile + K.P + 7) / 8 * 8 + 8; }  // (8 runs of ceil(tiles / 8) slots)

// scratch layout: int32 counters [128] (cnt[64], ticket), the per-port env lists [P][n_envs], then — 16-byte aligned — the state
// rows float32 [n_envs][kpad[0] <= 256] the prep launch's state waves write and the MLP kernel's tiles read
static long long dqn_rows_offset(const CimParams& K) { return ((128 + (long long)K.P * K.n_envs + 3) / 4) * 4; }   // in int32 words

int64_t mrx_cim_dqn_scratch_bytes(mrx_handle h) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  const CimParams& K = h->plan.kp;
  return 4 * (dqn_rows_offset(K) + (long long)K.n_envs * cim::DQ_MAX_WIDTH);
}

static int dqn_act(mrx_handle h, const mrx_cim_dqn_model* m, const int32_t* d_decisions, void* d_scratch, int32_t* d_actions,
                   int32_t* d_n_actions, float* d_q, float* d_state, int32_t* d_choice, uint64_t* d_counter, const cim::SamplerRec& R, void* stream) {
  using namespace cim;
  if (!h || !m || !d_decisions || !d_scratch || !d_actions || !d_n_actions || !m->d_weights) return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  const CimParams& K = h->plan.kp;
  if (K.decision_mode != 0) return set_err(MRX_ERR_INVALID_ARG, "mrx_cim_dqn_act answers Sequential-mode decisions");
  DqnParams D;
  int rc = dqn_plan(m, &D);
  if (rc != MRX_OK) return rc;
  if (m->n_nets != K.P) return set_err(MRX_ERR_INVALID_ARG, "one network per port: n_nets must equal n_ports");
  if (m->look_back < 2 || m->look_back - 1 > DQ_MAX_TICKS) return set_err(MRX_ERR_INVALID_ARG, "look_back out of range");
  if (m->n_port_attrs < 0 || m->n_port_attrs > 8 || m->n_vessel_attrs < 0 || m->n_vessel_attrs > 8) return set_err(MRX_ERR_INVALID_ARG, "at most 8 attributes each");
  if (1 + K.future_n > DQ_MAX_NODES - 1) return set_err(MRX_ERR_UNSUPPORTED, "future_stop_number too large");
  D.look_back = m->look_back;
  D.n_nodes = 1 + K.future_n;
  D.n_pa = m->n_port_attrs;
  D.n_va = m->n_vessel_attrs;
  for (int i = 0; i < D.n_pa; i++) {
    if (host_attr_slots(K, 0, m->port_attrs[i]) != 1) return set_err(MRX_ERR_INVALID_ARG, "unknown port attribute id");
    D.pa[i] = m->port_attrs[i];
  }
  for (int i = 0; i < D.n_va; i++) {
    if (host_attr_slots(K, 1, m->vessel_attrs[i]) != 1) return set_err(MRX_ERR_INVALID_ARG, "vessel attributes must be single-slot");
    D.va[i] = m->vessel_attrs[i];
  }
  if (D.state_dim != (D.look_back - 1) * D.n_nodes * D.n_pa + D.n_va)
    return set_err(MRX_ERR_INVALID_ARG, "dims[0] must be (look_back - 1) * (1 + future_stop_number) * n_port_attrs + n_vessel_attrs");
  D.weights = m->d_weights;
  rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  int32_t* cnt = (int32_t*)d_scratch;
  int32_t* lists = cnt + 128;
  float* xrows = (float*)(cnt + dqn_rows_offset(K));
  // as in mrx_cim_random_policy: with a sorted launch form one extra workgroup of the first launch builds the order list of the
  // coming step (it only depends on the previous step's hints), so the mrx_cim_step that follows on this stream needs no
  // schedule kernel of its own
  static const bool fuse = !(getenv("MRX_CIM_FUSE_SCHEDULE") && atoi(getenv("MRX_CIM_FUSE_SCHEDULE")) == 0);
  const int sched_per = (fuse && effective_step_mode(h) >= 2) ? ((K.n_envs + 255) / 256 + 15) / 16 * 16 : 0;
  const int n_bin = (K.n_envs + 255) / 256, n_state = (K.n_envs + 3) / 4;
  hipLaunchKernelGGL(mrx_k_cim_dqn_prep, dim3((unsigned)(n_bin + (sched_per > 0 ? 1 : 0) + n_state)), dim3(256), 0, (hipStream_t)stream, K, D, n_bin,
                     d_decisions, cnt, lists, d_n_actions, (unsigned long long*)d_counter, sched_per > 0 ? (sched_per | (h->lpt << 24)) : 0, xrows, d_state, R);
  h->order_ready = sched_per > 0;
  h->order_stream = stream;
  const int tile = dqn_tile_rows(K);
  static const bool xcd_runs = !(getenv("MRX_DQN_XCD") && atoi(getenv("MRX_DQN_XCD")) == 0);   // (0: block i takes tile i — experiments)
  D.xcd_runs = xcd_runs ? 1 : 0;
  if (tile == 16)
    hipLaunchKernelGGL(mrx_k_cim_dqn_mlp16, dim3((unsigned)dqn_max_tiles(K, 16)), dim3(256), 0, (hipStream_t)stream, K, D, d_decisions, cnt, lists, xrows,
                       d_actions, d_q, d_choice, R);
  else
    hipLaunchKernelGGL(mrx_k_cim_dqn_mlp32, dim3((unsigned)dqn_max_tiles(K, 32)), dim3(256), 0, (hipStream_t)stream, K, D, d_decisions, cnt, lists, xrows,
                       d_actions, d_q, d_choice, R);
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

int mrx_cim_dqn_act(mrx_handle h, const mrx_cim_dqn_model* m, const int32_t* d_decisions, void* d_scratch, int32_t* d_actions,
                    int32_t* d_n_actions, float* d_q, float* d_state, int32_t* d_choice, uint64_t* d_counter, void* stream) {
  cim::SamplerRec off = {};
  return dqn_act(h, m, d_decisions, d_scratch, d_actions, d_n_actions, d_q, d_state, d_choice, d_counter, off, stream);
}

// ---- the batched EnvSampler's cache as one argument block (mrx_cim_sampler_cache)
static int sampler_rec(mrx_handle h, const mrx_cim_sampler_cache* c, const uint8_t* d_done, cim::SamplerRec* R) {
  if (!h || !c) return set_err(MRX_ERR_INVALID_ARG, "null handle / cache");
  const CimParams& K = h->plan.kp;
  if (c->n_envs != K.n_envs || c->n_ports != K.P) return set_err(MRX_ERR_INVALID_ARG, "the cache was laid out for another engine (n_envs / n_ports differ)");
  if (c->cap <= 0 || (c->cap & (c->cap - 1))) return set_err(MRX_ERR_INVALID_ARG, "cap (slots of an env's transition ring) must be a power of two");
  if (c->state_dim <= 0 || !c->d_eoe || !c->d_head || !c->d_tail || !c->d_last || !c->d_prev_j || !c->d_prev_active || !c->d_interactions || !c->c_tick ||
      !c->c_agent || !c->c_state || !c->c_action || !c->c_env_action || !c->c_terminal || !c->c_next_state || !c->c_next_agent_state || !d_done)
    return set_err(MRX_ERR_INVALID_ARG, "null pointer in mrx_cim_sampler_cache");
  *R = cim::SamplerRec{1, c->cap, c->state_dim, c->state_f64 ? 1 : 0, K.max_actions, K.P, c->d_eoe, d_done, (long long*)c->d_head, (long long*)c->d_last,
                       (long long*)c->d_prev_j, c->d_prev_active, c->c_tick, (long long*)c->c_agent, c->c_state, (long long*)c->c_action, c->c_env_action,
                       c->c_terminal, c->c_next_state, c->c_next_agent_state, (long long*)c->d_interactions};
  return MRX_OK;
}

static int sampler_end(mrx_handle h, const mrx_cim_sampler_cache* c, const uint8_t* d_done, cim::SamplerEnd* E) {
  int rc = sampler_rec(h, c, d_done, &E->R);
  if (rc != MRX_OK) return rc;
  if (c->window <= 0 || c->frames <= 0 || !c->d_decay || !c->d_port_history) return set_err(MRX_ERR_INVALID_ARG, "reward window / port history missing");
  if (h->obs.hist != c->d_port_history || h->obs.hist_n != 2 || h->obs.hist_frames != c->frames || h->obs.hist_attr[0] != PA_FULFILLMENT || h->obs.hist_attr[1] != PA_SHORTAGE)
    return set_err(MRX_ERR_INVALID_ARG, "d_port_history must be the engine's mrx_cim_set_port_history(fulfillment, shortage) buffer");
  E->tail = (long long*)c->d_tail;
  E->window = c->window; E->frames = c->frames;
  E->ff = c->fulfillment_factor; E->sf = c->shortage_factor;
  E->decay = c->d_decay;
  E->hist = c->d_port_history;
  return MRX_OK;
}

int mrx_cim_collect_steps(mrx_handle h, const mrx_cim_dqn_model* m, void* d_scratch, const mrx_cim_sampler_cache* cache, int32_t* d_actions,
                          int32_t* d_n_actions, int32_t* d_decisions, int64_t* d_metrics, uint8_t* d_done, int32_t n_steps, void* stream) {
  cim::SamplerRec R;
  int rc = sampler_rec(h, cache, d_done, &R);
  if (rc != MRX_OK) return rc;
  if (n_steps < 0 || !d_decisions || !d_metrics || !d_actions || !d_n_actions) return set_err(MRX_ERR_INVALID_ARG, "null pointer / negative step count");
  // the forward kernel writes the model's state row (dims[0] values) into the cache's ring slots, whose row stride is
  // cache->state_dim: a mismatch would overwrite neighbouring slots / run past the arrays
  if (!m || cache->state_dim != m->dims[0]) return set_err(MRX_ERR_INVALID_ARG, "mrx_cim_collect_steps: cache->state_dim must equal the model's input width (dims[0])");
  for (int k = 0; k < n_steps; k++) {
    rc = dqn_act(h, m, d_decisions, d_scratch, d_actions, d_n_actions, nullptr, nullptr, nullptr, nullptr, R, stream);
    if (rc != MRX_OK) return rc;
    rc = launch_step(h, d_actions, d_n_actions, nullptr, nullptr, d_decisions, d_metrics, d_done, stream);   // (finished envs just report `done` again)
    if (rc != MRX_OK) return rc;
  }
  return MRX_OK;
}

int mrx_cim_sampler_finalize(mrx_handle h, const mrx_cim_sampler_cache* cache, const uint8_t* d_done, int64_t* d_n_emit, int64_t* d_out_offset,
                             int64_t* d_info, void* stream) {
  cim::SamplerEnd E;
  int rc = sampler_end(h, cache, d_done, &E);
  if (rc != MRX_OK) return rc;
  if (!d_n_emit || !d_out_offset || !d_info) return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  const CimParams& K = h->plan.kp;
  hipLaunchKernelGGL(mrx_k_cim_sampler_finalize, dim3((unsigned)K.n_envs), dim3(64), 0, (hipStream_t)stream, K, E, (long long*)d_n_emit);
  hipLaunchKernelGGL(mrx_k_cim_sampler_scan, dim3(1), dim3(1024), 0, (hipStream_t)stream, K.n_envs, (const long long*)d_n_emit, (const long long*)cache->d_head,
                     (const long long*)cache->d_tail, (const uint8_t*)cache->d_eoe, (const uint8_t*)cache->d_prev_active, (long long*)d_out_offset,
                     (long long*)d_info);
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

int mrx_cim_sampler_emit_all(mrx_handle h, const mrx_cim_sampler_cache* cache, const int64_t* d_n_emit, const int64_t* d_out_offset, void* o_state,
                             int64_t* o_action, int32_t* o_env_action, float* o_reward, void* o_next_state, void* o_next_agent_state,
                             uint8_t* o_terminal, int32_t* o_env_id, int32_t* o_tick, int32_t* o_agent, void* stream) {
  cim::SamplerEnd E;
  static const uint8_t dummy_done = 0;   // (the emission does not read `done`: sampler_rec only wants a non-null pointer)
  int rc = sampler_end(h, cache, &dummy_done, &E);
  if (rc != MRX_OK) return rc;
  if (!d_n_emit || !d_out_offset || !o_state || !o_action || !o_env_action || !o_reward || !o_next_state || !o_next_agent_state || !o_terminal || !o_env_id ||
      !o_tick || !o_agent)
    return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  const CimParams& K = h->plan.kp;
  // history rows staged per chunk: as many ticks as fit 48 KB (>= one reward window)
  int rows_cap = (48 * 1024) / (2 * K.P * 4);
  if (rows_cap < cache->window) return set_err(MRX_ERR_UNSUPPORTED, "reward window does not fit the emission kernel's LDS rows");
  const size_t lds = (size_t)rows_cap * 2 * K.P * 4;
#define MRX_EMIT_ALL(T)                                                                                                                              \
  hipLaunchKernelGGL(mrx_k_cim_sampler_emit_all<T>, dim3((unsigned)K.n_envs), dim3(256), lds, (hipStream_t)stream, E, rows_cap, (const long long*)d_n_emit, \
                     (const long long*)d_out_offset, (T*)o_state, (long long*)o_action, o_env_action, o_reward, (T*)o_next_state, (T*)o_next_agent_state,   \
                     o_terminal, o_env_id, o_tick, o_agent)
  if (cache->state_f64) MRX_EMIT_ALL(double); else MRX_EMIT_ALL(float);
#undef MRX_EMIT_ALL
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

}  // extern "C"
