// cb_engine.hip — gfx950 kernels + the C ABI of include/maro_amd_citi_bike.h.
//
// Launch geometry: ONE environment per LANE (cb_device.h explains why), 64 envs per wavefront, one wavefront per
// workgroup so that a batch spreads over as many CUs as it has waves.  State is struct-of-arrays [word][env]: the
// 64 lanes of a wave read/write 256 contiguous bytes whenever they agree on the word, which is the common case
// (every env replays the same trip table).  No LDS, no cross-lane traffic, no atomics.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>

#include "wave.h"
#include "cb_layout.h"
#include "cb_device.h"
#include "cb_wave.h"

struct CbAttrList { int n; int32_t id[16]; };

// The same query for wide station rows (hundreds of stations x several attributes per env), transposed through LDS: state is
// struct-of-arrays [word][env] (consecutive ENVS are contiguous) while the result is [env][tick][node][attr] (consecutive
// COLUMNS of one env are contiguous), so a thread-per-element kernel is uncoalesced on one side whatever its mapping — with the
// env-major mapping above every 8-byte result lands in its own 64-byte sector (measured on city.800s: 183 MB of results at
// 1.05 TB/s, 175 us per step).  Here a workgroup owns 64 envs x 32 columns of one tick: it gathers with consecutive threads on
// consecutive envs, and writes with consecutive threads on consecutive columns (256-byte runs per env).
extern "C" __global__ void __launch_bounds__(256)
mrx_k_cb_query_tiled(CbParams K, int node_type, const int32_t* __restrict__ ticks, int nt, int ticks_per_env, const int32_t* __restrict__ nodes,
                     int nn, int nodes_per_env, CbAttrList al, int row_slots, double* __restrict__ out) {
  __shared__ double tile[32][65];
  const int e0 = (int)blockIdx.x * 64, ti = (int)blockIdx.y, c0 = (int)blockIdx.z * 32;
  const int width = nn * row_slots;
  for (int idx = (int)threadIdx.x; idx < 64 * 32; idx += 256) {
    const int el = idx & 63, cl = idx >> 6;
    const int e = e0 + el, c = c0 + cl;
    if (e < K.n_envs && c < width) {
      const int ni = c / row_slots;
      tile[cl][el] = cb::query_elem(K, node_type, ticks, nt, ticks_per_env, nodes, nn, nodes_per_env, al.id, al.n, ((long long)e * nt + ti) * nn + ni, c - ni * row_slots);
    }
  }
  __syncthreads();
  for (int idx = (int)threadIdx.x; idx < 64 * 32; idx += 256) {
    const int cl = idx & 31, el = idx >> 5;
    const int e = e0 + el, c = c0 + cl;
    if (e < K.n_envs && c < width) out[((long long)e * nt + ti) * width + c] = tile[cl][el];
  }
}

// env-major layout (CbParams::aos): thread i computes result element i
extern "C" __global__ void __launch_bounds__(256)
mrx_k_cb_query_rows(CbParams K, int node_type, const int32_t* __restrict__ ticks, int nt, int ticks_per_env, const int32_t* __restrict__ nodes,
                    int nn, int nodes_per_env, CbAttrList al, int row_slots, long long total, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long row = i / row_slots;
  out[i] = cb::query_elem(K, node_type, ticks, nt, ticks_per_env, nodes, nn, nodes_per_env, al.id, al.n, row, (int)(i - row * row_slots));
}

int mrx_set_error_(int code, const std::string& m);  // cim_engine.hip (thread-local message behind mrx_last_error)

// ------------------------------------------------------------------------------------------ kernels
#include "cb_step_kernels.h"

extern "C" __global__ void __launch_bounds__(256)
mrx_k_cb_query(CbParams K, int node_type, const int32_t* __restrict__ ticks, int nt, int ticks_per_env, const int32_t* __restrict__ nodes,
               int nn, int nodes_per_env, CbAttrList al, int row_slots, long long total, double* __restrict__ out) {
  // consecutive threads take consecutive ENVS of one (tick, node, column) so the SoA state reads coalesce
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i % K.n_envs);
  long long r = i / K.n_envs;
  const int col = (int)(r % row_slots); r /= row_slots;
  const int ni = (int)(r % nn);
  const int ti = (int)(r / nn);
  const long long row = ((long long)e * nt + ti) * nn + ni;
  out[row * row_slots + col] = cb::query_elem(K, node_type, ticks, nt, ticks_per_env, nodes, nn, nodes_per_env, al.id, al.n, row, col);
}

extern "C" __global__ void __launch_bounds__(256)
mrx_k_cb_random_policy(CbParams K, const int32_t* __restrict__ decisions, const int32_t* __restrict__ scope, long long step,
                       int32_t* __restrict__ actions, int32_t* __restrict__ n_actions, unsigned long long* __restrict__ counter) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = false;
  if (e < K.n_envs)
    valid = cb::random_policy_env(K, e, decisions + (size_t)e * 8, scope + (size_t)e * K.scope_cap * 2, step,
                                  actions + (size_t)e * K.max_actions * 3, n_actions + e) != 0;
  if (counter) {
    const unsigned long long m = __ballot(valid);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(counter, (unsigned long long)__builtin_popcountll(m));
  }
}

// ------------------------------------------------------------------------------------------ C ABI
struct mrx_cb_engine {
  CbHostPlan plan;
  int device;
  int lanes = 64;  // envs per wave of the step kernel
  int step_budget = 0;  // mrx_cb_set_step_budget
  hipModule_t spec_module = nullptr;   // plan-specialised reset / step kernels (mrx_cb_load_step_kernels), else the generic ones
  hipFunction_t spec_reset = nullptr, spec_step = nullptr, spec_wave = nullptr, spec_replay = nullptr, spec_classify = nullptr;
  // mrx_cb_set_replay_overlap: the in-tick kernel and the replay kernel of one batch step side by side (a second stream, forked
  // from and joined back into the caller's stream by events inside mrx_cb_step)
  int replay_period = 1;   // mrx_cb_set_replay_period: the general (replay) kernel runs on every n-th step call
  long long step_calls = 0;
  bool overlap = false;
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int spec_lsh = -1;  // envs-per-wave shift compiled into the loaded step kernel (MRXC_lsh_plan of its plan text); -1: a kernel argument
  int wave_mode = 0;    // mrx_cb_set_wave_decisions: 0 automatic, 1 on, -1 off
  bool obs_wave = false;  // the row layout the fused observation's buffer was sized for: scope_cap rows (wave path) or S rows (lane path)
  // kernels of the module may still be queued or running on the caller's stream: drain the device before unloading
  void unload_spec() {
    if (!spec_module) return;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess) {
      if (cur != device) hipSetDevice(device);
      hipDeviceSynchronize();
      hipModuleUnload(spec_module);
      if (cur != device && cur >= 0) hipSetDevice(cur);
    }
    spec_module = nullptr;
    spec_wave = spec_replay = spec_classify = nullptr;
  }
  ~mrx_cb_engine() {
    unload_spec();
    if (side) {   // (unload_spec drained the device when kernels were loaded; the side stream only ever runs those)
      int cur = -1;
      if (hipGetDevice(&cur) == hipSuccess) {
        if (cur != device) hipSetDevice(device);
        hipStreamSynchronize(side);
        hipEventDestroy(ev_fork);
        hipEventDestroy(ev_join);
        hipStreamDestroy(side);
        if (cur != device && cur >= 0) hipSetDevice(cur);
      }
    }
  }
};

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

int64_t mrx_cb_workspace_bytes(const mrx_cb_topology* topo, const mrx_cb_config* cfg) {
  CbHostPlan pl;
  std::string err;
  int rc = cb_plan(topo, cfg, &pl, &err);
  if (rc != MRX_OK) { set_err(rc, err); return rc; }
  return pl.workspace_bytes;
}

int mrx_cb_create(const mrx_cb_topology* topo, const mrx_cb_config* cfg, void* d_workspace, int64_t workspace_bytes, mrx_cb_handle* out) {
  if (!out) return set_err(MRX_ERR_INVALID_ARG, "out handle is null");
  *out = nullptr;
  mrx_cb_engine* e = new (std::nothrow) mrx_cb_engine();
  if (!e) return set_err(MRX_ERR_INVALID_ARG, "out of host memory");
  std::string err;
  int rc = cb_plan(topo, cfg, &e->plan, &err);
  if (rc != MRX_OK) { delete e; return set_err(rc, err); }
  if (!d_workspace || workspace_bytes < e->plan.workspace_bytes || ((uintptr_t)d_workspace & 255)) {
    delete e;
    return set_err(MRX_ERR_WORKSPACE, "workspace is null, smaller than mrx_cb_workspace_bytes() or not 256-byte aligned");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { delete e; return set_err(MRX_ERR_NO_DEVICE, "no HIP device available"); }
  e->device = cfg->device;
  rc = use_device(e->device);
  if (rc != MRX_OK) { delete e; return rc; }
  cb_plan_bind(&e->plan, d_workspace);
  hipError_t he = hipMemcpy((uint8_t*)d_workspace + e->plan.const_off, e->plan.const_blob.data(), e->plan.const_blob.size(), hipMemcpyHostToDevice);
  if (he != hipSuccess) { delete e; return set_err(MRX_ERR_HIP, std::string("upload of the trip tables: ") + hipGetErrorString(he)); }
  const CbParams& K = e->plan.kp;
  // default transfer times: the distribution mean is not known here, so 1 tick until mrx_cb_reset supplies them
  he = hipMemsetD32((hipDeviceptr_t)K.tt, 1, (size_t)K.tt_cap * K.stride);
  if (he == hipSuccess) {
    hipLaunchKernelGGL(mrx_k_cb_reset, dim3((K.n_envs + 63) / 64), dim3(64), 0, 0, K, nullptr, 0, nullptr);
    he = hipDeviceSynchronize();
  }
  if (he == hipSuccess) he = hipGetLastError();
  if (he != hipSuccess) { delete e; return set_err(MRX_ERR_HIP, std::string("initial reset kernel: ") + hipGetErrorString(he)); }
  e->lanes = cb_auto_lanes(K.n_envs);
  if (const char* v = getenv("MRX_CB_LANES")) mrx_cb_set_lanes_per_wave(e, atoi(v));
  *out = e;
  return MRX_OK;
}

int mrx_cb_set_lanes_per_wave(mrx_cb_handle h, int lanes) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  if (lanes == 0) lanes = cb_auto_lanes(h->plan.kp.n_envs);
  if (lanes < 1 || lanes > 64 || (lanes & (lanes - 1))) return set_err(MRX_ERR_INVALID_ARG, "lanes per wave must be 1, 2, 4, ..., 64 (0 = automatic)");
  h->lanes = lanes;
  return MRX_OK;
}

// Whether Sequential-mode steps go through the wave-cooperative decision kernel first (cb_wave.h).  It needs aligned frames
// (start_tick a multiple of the snapshot resolution: otherwise every decision materialises a snapshot) and at most 2048 stations.
static bool cb_wave_applicable(const CbParams& K) {
  return K.decision_mode == 0 && K.start_tick % K.res == 0 && K.mask_words <= 64;
}
static bool cb_wave_on(mrx_cb_handle h) {
  const CbParams& K = h->plan.kp;
  if (!cb_wave_applicable(K) || h->wave_mode < 0) return false;
  // automatic: from ~100 stations on a decision tick raises dozens of decisions per env and the action scope dominates a step;
  // on the toys (3-5 stations) the extra launch costs more than it saves
  return h->wave_mode > 0 || K.S >= 96;
}

// Which kernels write the fused observation (mrx_cb_set_observation) on the path in effect: the wave kernels (rows = the action
// scope's stations, scope_cap per env) or the one-env-per-lane kernel (rows = every station).
static bool cb_replay_ok(mrx_cb_handle h, bool has_replay) { return has_replay && (int64_t)h->plan.kp.lds_words * 4 <= MRX_CB_LDS_BYTES; }
// (wave-stepped plans write the observation from BOTH wave kernels, which needs the plan-specialised replay kernel: mrx_cb_set_observation)
static bool cb_obs_consistent(mrx_cb_handle h, bool has_replay) {
  return cb_wave_on(h) == h->obs_wave && (!h->obs_wave || cb_replay_ok(h, has_replay));
}

int mrx_cb_set_wave_decisions(mrx_cb_handle h, int mode) {
  if (!h || mode < -1 || mode > 1) return set_err(MRX_ERR_INVALID_ARG, "null handle, or mode not in {-1 off, 0 automatic, 1 on}");
  if (mode > 0 && !cb_wave_applicable(h->plan.kp)) return set_err(MRX_ERR_UNSUPPORTED, "the wave-cooperative decision step needs Sequential mode, aligned frames and <= 2048 stations");
  const int before = h->wave_mode;
  h->wave_mode = mode;
  // a fused observation is sized for ONE row layout: a switch that would change who writes it (and how many rows) is refused —
  // the lane kernel writing S rows per env into a scope_cap-row buffer would run past it
  if (h->plan.kp.obs && !cb_obs_consistent(h, h->spec_replay != nullptr)) {
    h->wave_mode = before;
    return set_err(MRX_ERR_UNSUPPORTED, "a fused observation is configured for the other step path: switch it off (mrx_cb_set_observation with n_attrs = 0) before changing the wave mode");
  }
  return cb_wave_on(h) ? 1 : 0;
}

int mrx_cb_set_replay_overlap(mrx_cb_handle h, int on) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  h->overlap = on != 0;
  return MRX_OK;
}

int mrx_cb_set_replay_period(mrx_cb_handle h, int n, int phase) {
  if (!h || n < 1 || phase < 0) return set_err(MRX_ERR_INVALID_ARG, "null handle, period < 1 or negative phase");
  if (n > 1 && h->plan.kp.max_actions > CB_STASH_MAX) return set_err(MRX_ERR_UNSUPPORTED, "a replay period needs max_actions <= 4 (the deferred answer is kept per env)");
  h->replay_period = n;
  h->step_calls = phase % n;   // the replay kernel runs on the calls c with (phase + c) % n == 0, c = 1, 2, ...
  return MRX_OK;
}

int mrx_cb_observation_rows(mrx_cb_handle h) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  return cb_wave_on(h) ? h->plan.layout.scope_cap : h->plan.kp.S;
}

int mrx_cb_set_step_budget(mrx_cb_handle h, int max_records) {
  if (!h || max_records < 0) return set_err(MRX_ERR_INVALID_ARG, "null handle or negative budget");
  h->step_budget = max_records;
  return MRX_OK;
}

int mrx_cb_set_observation(mrx_cb_handle h, const int32_t* station_attrs, int n_attrs, double* d_obs) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  CbParams& K = h->plan.kp;
  if (n_attrs < 0 || n_attrs > 8 || (n_attrs > 0 && (!station_attrs || !d_obs))) return set_err(MRX_ERR_INVALID_ARG, "at most 8 station attributes, with an output buffer");
  if (n_attrs > 0 && K.decision_mode != 0) return set_err(MRX_ERR_UNSUPPORTED, "the fused observation is defined for Sequential decision mode");
  // (plans stepped one env per lane: rows = every station; plans stepped by the wave kernels: rows = the stations of the action scope)
  if (n_attrs > 0 && cb_wave_on(h) && !(h->spec_replay && (int64_t)K.lds_words * 4 <= MRX_CB_LDS_BYTES))
    return set_err(MRX_ERR_UNSUPPORTED, "on wave-stepped plans the fused observation needs the plan-specialised kernels (mrx_cb_load_step_kernels)");
  for (int i = 0; i < n_attrs; i++) {
    if (station_attrs[i] < 0 || station_attrs[i] >= SA_COUNT) return set_err(MRX_ERR_INVALID_ARG, "unknown station attribute id");
    K.obs_attr[i] = station_attrs[i];
  }
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());   // (steps may be in flight with the previous configuration)
  K.obs_n = n_attrs;
  K.obs = n_attrs > 0 ? d_obs : nullptr;
  h->obs_wave = cb_wave_on(h);   // d_obs holds mrx_cb_observation_rows(h) rows per env
  return MRX_OK;
}

int mrx_cb_destroy(mrx_cb_handle h) {
  delete h;
  return MRX_OK;
}

int mrx_cb_get_layout(mrx_cb_handle h, mrx_cb_layout* out) {
  if (!h || !out) return set_err(MRX_ERR_INVALID_ARG, "null handle/out");
  *out = h->plan.layout;
  return MRX_OK;
}

int mrx_cb_reset(mrx_cb_handle h, const int32_t* d_transfer_times, int32_t n_times, const uint8_t* d_env_mask, void* stream) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  if (d_transfer_times && n_times <= 0) return set_err(MRX_ERR_INVALID_ARG, "n_times must be positive when transfer times are given");
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  const CbParams& K = h->plan.kp;
  if (h->spec_module) {
    CbParams Kc = K;
    int nt = (int)n_times;
    void* params[] = {&Kc, &d_transfer_times, &nt, &d_env_mask};
    HIP_TRY(hipModuleLaunchKernel(h->spec_reset, (unsigned)((K.n_envs + 63) / 64), 1, 1, 64, 1, 1, 0, (hipStream_t)stream, params, nullptr));
    return MRX_OK;
  }
  hipLaunchKernelGGL(mrx_k_cb_reset, dim3((K.n_envs + 63) / 64), dim3(64), 0, (hipStream_t)stream, K, d_transfer_times, (int)n_times, d_env_mask);
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

static int cb_launch_step(mrx_cb_handle h, const int32_t* d_actions, const int32_t* d_n_actions, const int32_t* d_n_answered, const uint8_t* d_env_mask,
                          int32_t* d_decisions, int32_t* d_scope, int64_t* d_metrics, uint8_t* d_done, void* stream) {
  if (!d_decisions || !d_scope || !d_metrics || !d_done) return set_err(MRX_ERR_INVALID_ARG, "null output pointer");
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  const CbParams& K = h->plan.kp;
  if (K.obs && !cb_obs_consistent(h, h->spec_replay != nullptr))   // (cannot happen through the setters; never write past the buffer)
    return set_err(MRX_ERR_INVALID_ARG, "the fused observation was configured for the other step path: call mrx_cb_set_observation again");
  CbParams Kc = K;
  Kc.step_budget = h->step_budget;
  if (cb_wave_on(h)) {
    long long* metw = (long long*)d_metrics;
    const bool replay = h->spec_replay && (int64_t)K.lds_words * 4 <= MRX_CB_LDS_BYTES;
    if (replay && h->spec_classify && h->overlap) {
      // The two wave kernels side by side.  Which envs leave their tick is read off the state first (mrx_k_cb_classify -> K.todo);
      // then the replay kernel (few envs, a long sequential chain each) runs on the side stream while the in-tick kernel (all the
      // others) runs on the caller's: a batch step costs the longer of the two instead of their sum.  Disjoint envs, no shared
      // words; the caller's stream continues after both.
      if (!h->side) {
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi));   // (the long pole of the step: first in line for CUs)
        HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
      }
      hipStream_t main_s = (hipStream_t)stream;
      void* pc[] = {&Kc, &d_actions, &d_n_actions, &d_env_mask};
      HIP_TRY(hipModuleLaunchKernel(h->spec_classify, (unsigned)K.n_envs, 1, 1, 64, 1, 1, 0, main_s, pc, nullptr));
      HIP_TRY(hipEventRecord(h->ev_fork, main_s));
      HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
      CbParams Kr = Kc;
      Kr.lsh = 0;
      Kr.pool_stage = CB_POOL_STAGE;
      const uint8_t* todo = K.todo;
      void* pr[] = {&Kr, &d_actions, &d_n_actions, &todo, &d_decisions, &d_scope, &metw, &d_done};
      HIP_TRY(hipModuleLaunchKernel(h->spec_replay, (unsigned)K.n_envs, 1, 1, 64, 1, 1, (unsigned)(K.lds_words * 4), h->side, pr, nullptr));
      HIP_TRY(hipEventRecord(h->ev_join, h->side));
      int classified = 1;
      void* pw[] = {&Kc, &d_actions, &d_n_actions, &d_env_mask, &d_decisions, &d_scope, &metw, &d_done, &classified};
      HIP_TRY(hipModuleLaunchKernel(h->spec_wave, (unsigned)K.n_envs, 1, 1, 64, 1, 1, 0, main_s, pw, nullptr));
      HIP_TRY(hipStreamWaitEvent(main_s, h->ev_join, 0));
      return MRX_OK;
    }
    // mrx_cb_set_replay_period: on the calls in between the replay kernel sits out; envs that leave their tick put their answer aside
    // and report "no decision yet" (cb::defer_env_wave)
    const bool defer = replay && h->spec_wave && h->replay_period > 1 && (++h->step_calls % h->replay_period) != 0;
    Kc.defer = defer ? 1 : 0;
    // one env per wave: the steps that stay inside their tick; everything else is flagged in K.todo for the general kernel below
    int classified = 0;
    if (h->spec_wave) {
      void* pw[] = {&Kc, &d_actions, &d_n_actions, &d_env_mask, &d_decisions, &d_scope, &metw, &d_done, &classified};
      HIP_TRY(hipModuleLaunchKernel(h->spec_wave, (unsigned)K.n_envs, 1, 1, 64, 1, 1, 0, (hipStream_t)stream, pw, nullptr));
    } else {
      hipLaunchKernelGGL(mrx_k_cb_step_wave, dim3(K.n_envs), dim3(64), 0, (hipStream_t)stream, Kc, d_actions, d_n_actions, d_env_mask, d_decisions, d_scope,
                         metw, d_done, classified);
    }
    if (defer) return MRX_OK;
    d_env_mask = K.todo;
    if (replay) {
      // ... and the general step for the flagged envs, also one env per wave: state in the wave's LDS column (cb::step_env_wave)
      CbParams Kr = Kc;
      Kr.lsh = 0;
      Kr.pool_stage = CB_POOL_STAGE;
      const uint8_t* todo = K.todo;
      void* pr[] = {&Kr, &d_actions, &d_n_actions, &todo, &d_decisions, &d_scope, &metw, &d_done};
      HIP_TRY(hipModuleLaunchKernel(h->spec_replay, (unsigned)K.n_envs, 1, 1, 64, 1, 1, (unsigned)(K.lds_words * 4), (hipStream_t)stream, pr, nullptr));
      return MRX_OK;
    }
  }
  if (h->spec_module) {
    long long* met = (long long*)d_metrics;
    // the specialised kernel keeps each env's working state in an LDS column of lds_words words (when one column fits at
    // all): as many envs per wave as asked for and as fit
    int lanes = h->lanes;
    const bool lds_frame = (int64_t)K.lds_words * 4 <= MRX_CB_LDS_BYTES;
    while (lds_frame && lanes > 1 && (int64_t)K.lds_words * 4 * lanes > MRX_CB_LDS_BYTES) lanes /= 2;
    int lsh = 0;
    while ((1 << lsh) < lanes) lsh++;
    Kc.lsh = lsh;
    // (a plan whose specialised step kernel has the envs-per-wave shift compiled in runs it with that shift only: another choice
    // of mrx_cb_set_lanes_per_wave / MRX_CB_LANES goes to the generic kernel below)
    if (h->spec_lsh < 0 || !lds_frame || lsh == h->spec_lsh) {
      const unsigned lds_bytes = lds_frame ? (unsigned)(K.lds_words * 4 * lanes) : 0u;
      void* params[] = {&Kc, &d_actions, &d_n_actions, &d_env_mask, &d_decisions, &d_scope, &met, &d_done, &lanes, &d_n_answered};
      HIP_TRY(hipModuleLaunchKernel(h->spec_step, (unsigned)((K.n_envs + lanes - 1) / lanes), 1, 1, 64, 1, 1, lds_bytes, (hipStream_t)stream, params, nullptr));
      return MRX_OK;
    }
  }
  hipLaunchKernelGGL(mrx_k_cb_step, dim3((K.n_envs + h->lanes - 1) / h->lanes), dim3(64), 0, (hipStream_t)stream, Kc, d_actions, d_n_actions, d_env_mask,
                     d_decisions, d_scope, (long long*)d_metrics, d_done, h->lanes, d_n_answered);
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

int mrx_cb_step(mrx_cb_handle h, const int32_t* d_actions, const int32_t* d_n_actions, const uint8_t* d_env_mask, int32_t* d_decisions,
                int32_t* d_scope, int64_t* d_metrics, uint8_t* d_done, void* stream) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  if (h->plan.kp.decision_mode != 0) return set_err(MRX_ERR_INVALID_ARG, "engine was created in a Joint decision mode: use mrx_cb_step_joint");
  return cb_launch_step(h, d_actions, d_n_actions, nullptr, d_env_mask, d_decisions, d_scope, d_metrics, d_done, stream);
}

int mrx_cb_step_joint(mrx_cb_handle h, const int32_t* d_actions, const int32_t* d_n_actions, const int32_t* d_n_answered, const uint8_t* d_env_mask,
                      int32_t* d_decisions, int32_t* d_scope, int64_t* d_metrics, uint8_t* d_done, void* stream) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  if (h->plan.kp.decision_mode == 0) return set_err(MRX_ERR_INVALID_ARG, "engine was created in Sequential decision mode: use mrx_cb_step");
  return cb_launch_step(h, d_actions, d_n_actions, d_n_answered, d_env_mask, d_decisions, d_scope, d_metrics, d_done, stream);
}

// ---- plan-specialised kernels (cb_spec.hip)
static std::string cb_plan_defines(const CbParams& K) {
  std::string o;
#define X(f) o += std::string("#define MRXC_") + #f + " " + std::to_string((long long)K.f) + "\n";
  MRX_CB_DIM_FIELDS(X)
#undef X
#define X(f)                                                                                                    \
  o += std::string("#define MRXC_") + #f + "(i) ((i) == 0 ? " + std::to_string(K.f[0]) + " : (i) == 1 ? " +     \
       std::to_string(K.f[1]) + " : (i) == 2 ? " + std::to_string(K.f[2]) + " : " + std::to_string(K.f[3]) + ")\n";
  MRX_CB_DIM_ARRAYS(X)
#undef X
  return o;
}

int64_t mrx_cb_plan_defines(const mrx_cb_topology* topo, const mrx_cb_config* cfg, char* buf, int64_t len) {
  CbHostPlan plan;
  std::string err;
  const int rc = cb_plan(topo, cfg, &plan, &err);
  if (rc != MRX_OK) return set_err(rc, err);
  const std::string d = cb_plan_defines(plan.kp);
  if (buf && len > 0) {
    if ((int64_t)d.size() + 1 > len) return set_err(MRX_ERR_INVALID_ARG, "buffer too small");
    memcpy(buf, d.c_str(), d.size() + 1);
  }
  return (int64_t)d.size() + 1;
}

int mrx_cb_load_step_kernels(mrx_cb_handle h, const void* image, int64_t bytes, const char* defines) {
  if (!h || !image || bytes <= 0 || !defines) return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  // The plan text must be this plan's — except for MRXC_lsh_plan, which may also be -1: a build that takes the envs-per-wave shift
  // as a kernel argument (what the host side loads when mrx_cb_set_lanes_per_wave asks for another split than the automatic one).
  int spec_lsh = h->plan.kp.lsh_plan;
  {
    CbParams alt = h->plan.kp;
    alt.lsh_plan = -1;
    if (cb_plan_defines(h->plan.kp) == defines) spec_lsh = h->plan.kp.lsh_plan;
    else if (cb_plan_defines(alt) == defines) spec_lsh = -1;
    else return set_err(MRX_ERR_INVALID_ARG, "the code object was built for a different plan (defines differ)");
  }
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  hipModule_t mod = nullptr;
  HIP_TRY(hipModuleLoadData(&mod, image));
  hipFunction_t f_reset = nullptr, f_step = nullptr;
  if (hipModuleGetFunction(&f_reset, mod, "mrx_k_cb_reset") != hipSuccess || hipModuleGetFunction(&f_step, mod, "mrx_k_cb_step") != hipSuccess) {
    hipModuleUnload(mod);
    return set_err(MRX_ERR_INVALID_ARG, "code object lacks mrx_k_cb_reset / mrx_k_cb_step");
  }
  if (h->plan.kp.obs) {  // would the new kernels change who writes the fused observation?
    hipFunction_t probe = nullptr;
    const bool has_replay = hipModuleGetFunction(&probe, mod, "mrx_k_cb_replay_wave") == hipSuccess && probe;
    if (!has_replay) (void)hipGetLastError();
    if (!cb_obs_consistent(h, has_replay)) {
      hipModuleUnload(mod);
      return set_err(MRX_ERR_UNSUPPORTED, "a fused observation is configured for the other step path: switch it off before loading step kernels, then set it again");
    }
  }
  h->unload_spec();
  h->spec_module = mod;
  h->spec_reset = f_reset;
  h->spec_step = f_step;
  h->spec_lsh = spec_lsh;
  hipFunction_t f_wave = nullptr;
  if (hipModuleGetFunction(&f_wave, mod, "mrx_k_cb_step_wave") == hipSuccess && f_wave) h->spec_wave = f_wave;
  else (void)hipGetLastError();
  hipFunction_t f_replay = nullptr;
  if (hipModuleGetFunction(&f_replay, mod, "mrx_k_cb_replay_wave") == hipSuccess && f_replay) h->spec_replay = f_replay;
  else (void)hipGetLastError();
  hipFunction_t f_classify = nullptr;
  if (hipModuleGetFunction(&f_classify, mod, "mrx_k_cb_classify") == hipSuccess && f_classify) h->spec_classify = f_classify;
  else (void)hipGetLastError();
  return MRX_OK;
}

int mrx_cb_random_policy(mrx_cb_handle h, const int32_t* d_decisions, const int32_t* d_scope, int64_t step, int32_t* d_actions,
                         int32_t* d_n_actions, uint64_t* d_counter, void* stream) {
  if (!h || !d_decisions || !d_scope || !d_actions || !d_n_actions) return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  const CbParams& K = h->plan.kp;
  if (K.decision_mode != 0) return set_err(MRX_ERR_INVALID_ARG, "mrx_cb_random_policy answers Sequential-mode decisions");
  hipLaunchKernelGGL(mrx_k_cb_random_policy, dim3((K.n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, K, d_decisions, d_scope,
                     (long long)step, d_actions, d_n_actions, (unsigned long long*)d_counter);
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

int mrx_cb_attr_slots(mrx_cb_handle h, int node_type, int attr_id) {
  if (!h) return set_err(MRX_ERR_INVALID_ARG, "null handle");
  const CbParams& K = h->plan.kp;
  if (node_type == 0) return (attr_id >= 0 && attr_id < SA_COUNT) ? 1 : -1;
  if (node_type == 1) return attr_id == CB_MA_TRIPS_ADJ ? K.S * K.S : -1;
  return -1;
}

int mrx_cb_attr_id(int node_type, const char* name) {
  static const char* ST[] = {"bikes", "shortage", "trip_requirement", "fulfillment", "capacity", "id", "weekday", "temperature",
                             "weather", "holiday", "extra_cost", "transfer_cost", "failed_return", "min_bikes"};
  static const char* M[] = {"trips_adj"};
  const char** tab = node_type == 0 ? ST : node_type == 1 ? M : nullptr;
  const int n = node_type == 0 ? SA_COUNT : node_type == 1 ? CB_MA_COUNT : 0;
  if (!name) return -1;
  for (int i = 0; i < n; i++) if (!strcmp(tab[i], name)) return i;
  return -1;
}

int mrx_cb_query(mrx_cb_handle h, int node_type, const int32_t* d_ticks, int nt, int ticks_per_env, const int32_t* d_nodes, int nn,
                 int nodes_per_env, const int32_t* attrs, int na, double* d_out, void* stream) {
  if (!h || !d_ticks || !d_nodes || !attrs || !d_out) return set_err(MRX_ERR_INVALID_ARG, "null pointer");
  if (na > 16) return set_err(MRX_ERR_INVALID_ARG, "at most 16 attributes per query");
  if (nt <= 0 || nn <= 0 || na <= 0) return set_err(MRX_ERR_INVALID_ARG, "nt, nn and na must be positive");
  int rc = use_device(h->device);
  if (rc != MRX_OK) return rc;
  const CbParams& K = h->plan.kp;
  int row_slots = 0;
  for (int i = 0; i < na; i++) {
    const int s = mrx_cb_attr_slots(h, node_type, attrs[i]);
    if (s < 0) return set_err(MRX_ERR_INVALID_ARG, "unknown attribute id for this node type");
    row_slots += s;
  }
  const long long total = (long long)K.n_envs * nt * nn * row_slots;
  CbAttrList al;
  al.n = na;
  for (int i = 0; i < 16; i++) al.id[i] = i < na ? attrs[i] : 0;
  const long long width = (long long)nn * row_slots;
  if (K.aos) {
    // env-major state: an env's rows are contiguous, and so is its result row — consecutive threads take consecutive result elements
    hipLaunchKernelGGL(mrx_k_cb_query_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K, node_type, d_ticks, nt,
                       ticks_per_env, d_nodes, nn, nodes_per_env, al, row_slots, total, d_out);
  } else if (node_type == 0 && width >= 64 && nt <= 65535 && (width + 31) / 32 <= 65535) {
    hipLaunchKernelGGL(mrx_k_cb_query_tiled, dim3((unsigned)((K.n_envs + 63) / 64), (unsigned)nt, (unsigned)((width + 31) / 32)), dim3(256), 0, (hipStream_t)stream, K,
                       node_type, d_ticks, nt, ticks_per_env, d_nodes, nn, nodes_per_env, al, row_slots, d_out);
  } else {
    hipLaunchKernelGGL(mrx_k_cb_query, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K, node_type, d_ticks, nt,
                       ticks_per_env, d_nodes, nn, nodes_per_env, al, row_slots, total, d_out);
  }
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

}  // extern "C"
