// cim_dqn.h — on-device action selection for the CIM RL example: for every env that
// pauses at a decision, build the CIMEnvSampler state from the snapshot ring (examples/cim/rl/env_sampler.py:15-31),
// evaluate the deciding port's dueling DQN (examples/cim/rl/algorithms/dqn.py:13-84, maro/rl/model/fc_block.py:72-133),
// take the greedy action and translate it into an env Action (env_sampler.py:33-64) — two launches, no host round trip.
//
// This is the one dense-contraction piece next to the simulator, so it runs on the matrix cores: exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate — the reference networks are f32), envs binned by port so that a
// workgroup multiplies a tile of envs by ONE port's weights:
//   mrx_k_cim_dqn_bin      the deciding envs appended to their port's list (LDS histogram + one global atomic per port and workgroup)
//   mrx_k_cim_dqn_forward  one workgroup (4 waves) per 32-env tile of one port's list: state gather -> LDS, the dense chain layer by layer with the
//                          activations kept in one LDS buffer (written back in place between barriers), weights streamed
//                          from L2 as coalesced 16-B loads in the packed layout below, then argmax + action translation.
// HIP only (not part of the CPU wave emulator build).
#pragma once
#include <hip/hip_runtime.h>

#include "cim_device.h"

namespace cim {

enum { DQ_MAX_LAYERS = 8, DQ_MAX_WIDTH = 256, DQ_TILE_MAX = 32, DQ_LD = DQ_MAX_WIDTH + 4, DQ_MAX_TICKS = 16, DQ_MAX_NODES = 8, DQ_PF = 4 };

struct DqnParams {
  int n_layers, dueling, state_dim, look_back, n_nodes, n_pa, n_va, n_actions;
  int xcd_runs;  // 1: XCD-aware tile order (launch configuration, set by the host)
  int pa[8], va[8];
  int kpad[DQ_MAX_LAYERS], npad[DQ_MAX_LAYERS], n_out[DQ_MAX_LAYERS];
  long long w_off[DQ_MAX_LAYERS], b_off[DQ_MAX_LAYERS], net_floats;
  float slope, epsilon;
  const float* weights;
  double action_space[32];
};

// The batched EnvSampler's transition cache as the policy launches see it (mrx_cim_collect_steps): when `on`, the binning launch
// retires the envs whose episode is over and the forward kernel APPENDS each deciding env's transition — its state row is in LDS
// anyway — instead of a separate cache-update launch reading the state back from HBM (mrx_cim_sampler_record, whose semantics
// these two pieces reproduce exactly; maro/rl/rollout/env_sampler.py:404-410, 484-511).  An env's cache is a ring of `cap` slots
// (a power of two): element number q lives in slot q & (cap - 1); `last` holds element numbers (-1: none), `prev_j` the SLOT the
// env's previous step wrote, `prev_active` whether that element still waits for its next_state.
struct SamplerRec {
  int on, cap, D, f64, A, P;
  uint8_t* eoe;
  const uint8_t* done;
  long long *head, *last, *prev_j;
  uint8_t* prev_active;
  int32_t* c_tick;
  long long* c_agent;
  void* c_state;
  long long* c_action;
  int32_t* c_env_action;
  uint8_t* c_terminal;
  void *c_next_state, *c_nas;
  long long* steps_env;
};

__device__ __forceinline__ void rec_store(void* base, size_t i, float v, int f64) {
  if (f64) ((double*)base)[i] = (double)v; else ((float*)base)[i] = v;
}

// padded widths: a layer's outputs are split over the 4 waves in 16-column MFMA tiles
__host__ __device__ inline int dq_npad(int n) { return n <= 16 ? 16 : n <= 32 ? 32 : (n + 63) / 64 * 64; }
__host__ __device__ inline int dq_kpad(int k) { return (k + 15) / 16 * 16; }

// Packed weight layout of one layer (K = kpad inputs, N = npad outputs, zero padded): 16 inputs x 1 output column are
// stored as 4 groups g of 4 consecutive floats, Wp[((kb * N + n) * 4 + g) * 4 + s] = W[kb * 16 + g * 4 + s][n], so that
// MFMA lane (n & 15, g) fetches its B operands of four consecutive k-steps with one 16-byte load and a wave reads
// 1 KB contiguous.  The A operand uses the same k permutation (one ds_read_b128 per lane from the activation row).
__host__ __device__ inline long long dq_w_index(int k, int n, int N) {
  return (((long long)(k >> 4) * N + n) * 4 + ((k >> 2) & 3)) * 4 + (k & 3);
}

typedef float dq_f4 __attribute__((ext_vector_type(4)));

// DQ_PFL k-blocks of weights are in flight per wave (the rotating buffer below).  A layer's time is (its k-blocks / DQ_PFL)
// memory latencies — a workgroup is alone on its CU at the batch sizes of config 5, so nothing else hides them — and the
// buffer costs DQ_PFL x NT x 4 registers: narrow layers (NT = 1, 2) therefore fetch deeper at the same register count
// (DQ_PFL x NT = 16 for one MFMA row tile), e.g. all 16 k-blocks of the 256 -> 32 head at once instead of four rounds of four.
template <int MT, int NT, int DQ_PFL>
__device__ __forceinline__ void dq_dense(float* X, const float* __restrict__ Wp, const float* __restrict__ bias, int Kpad, int Npad,
                                         int m0, int nt0, int nt_step, bool idle, bool act, float slope) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  // the bias is requested with the first weights, not after the k loop (one memory latency per layer less)
  float bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) bv[nt] = idle ? 0.f : bias[(nt0 + nt * nt_step) * 16 + r];
  dq_f4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) acc[mt][nt] = dq_f4{0.f, 0.f, 0.f, 0.f};
  const int nkb = idle ? 0 : Kpad >> 4;
  // B operands (weights, L2) are fetched DQ_PFL k-blocks ahead into a rotating register buffer: a narrow layer gives a wave
  // only a few MFMAs per k-block, far less than the L2 latency
  dq_f4 bq[DQ_PFL][NT];
  const float* wlane = Wp + ((size_t)nt0 * 16 + r) * 16 + g * 4;
  const float* xlane = X + (m0 * 16 + r) * DQ_LD + g * 4;
  const int nfull = nkb / DQ_PFL, rem = nkb - nfull * DQ_PFL;
  if (nkb > 0) {
#pragma unroll
    for (int p = 0; p < DQ_PFL; p++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) bq[p][nt] = *(const dq_f4*)(wlane + ((size_t)min(p, nkb - 1) * Npad + (size_t)nt * nt_step * 16) * 16);
  }
  // Static register slots (a rotating buffer's moves would wait for the loads they move) and no control flow inside a pass
  // (a branch merge makes the compiler drain every outstanding load): slot p is refilled with the block DQ_PFL ahead right
  // after it is consumed.  The LAST full pass refills only when tail blocks follow it (then clamped: the last block may be
  // fetched twice); a layer whose k-blocks are a whole number of passes issues no load it does not use — the barrier that
  // ends the layer drains every load in flight, so a useless re-fetch late in the k loop cost the layer a memory latency.
#define DQ_PASS(I, REFILL)                                                                                                  \
  _Pragma("unroll") for (int p = 0; p < DQ_PFL; p++) {                                                                      \
    const int kb = (I) * DQ_PFL + p;                                                                                        \
    dq_f4 a[MT], b[NT];                                                                                                     \
    _Pragma("unroll") for (int nt = 0; nt < NT; nt++) {                                                                     \
      b[nt] = bq[p][nt];                                                                                                    \
      if (REFILL) bq[p][nt] = *(const dq_f4*)(wlane + ((size_t)min(kb + DQ_PFL, nkb - 1) * Npad + (size_t)nt * nt_step * 16) * 16); \
    }                                                                                                                       \
    _Pragma("unroll") for (int mt = 0; mt < MT; mt++) a[mt] = *(const dq_f4*)(xlane + mt * 16 * DQ_LD + kb * 16);           \
    _Pragma("unroll") for (int s = 0; s < 4; s++)                                                                           \
      _Pragma("unroll") for (int mt = 0; mt < MT; mt++)                                                                     \
        _Pragma("unroll") for (int nt = 0; nt < NT; nt++)                                                                   \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);                     \
  }
  const int n_refill = rem > 0 ? nfull : nfull - 1;
  for (int i = 0; i < n_refill; i++) { DQ_PASS(i, true) }
  if (rem == 0 && nfull > 0) { DQ_PASS(nfull - 1, false) }
#undef DQ_PASS
#pragma unroll
  for (int p = 0; p + 1 < DQ_PFL; p++)
    if (p < rem) {  // the tail blocks are already in their slots
      const int kb = nfull * DQ_PFL + p;
      dq_f4 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; mt++) a[mt] = *(const dq_f4*)(xlane + mt * 16 * DQ_LD + kb * 16);
#pragma unroll
      for (int s = 0; s < 4; s++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][s], bq[p][nt][s], acc[mt][nt], 0, 0, 0);
    }
  __syncthreads();  // every wave has read the layer's input: the buffer may be overwritten with its output
  if (!idle) {
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int col = (nt0 + nt * nt_step) * 16 + r;
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float v = acc[mt][nt][i] + bv[nt];  // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + i
          if (act) v = v > 0.f ? v : v * slope;
          X[((m0 + mt) * 16 + g * 4 + i) * DQ_LD + col] = v;
        }
    }
  }
  __syncthreads();
}

// A layer for the TILE-row tile (TILE / 16 MFMA row tiles): the output columns are dealt to the 4 waves in 16-column tiles.
template <int TILE>
__device__ __forceinline__ void dq_layer(float* X, const float* Wp, const float* bias, int Kpad, int Npad, bool act, float slope) {
  const int w = threadIdx.x >> 6;
  if constexpr (TILE == 32) {
    switch (Npad) {
      case 16: dq_dense<1, 1, 8>(X, Wp, bias, Kpad, Npad, w & 1, 0, 1, w >= 2, act, slope); break;   // 2 tiles: waves 2, 3 idle
      case 32: dq_dense<1, 1, 8>(X, Wp, bias, Kpad, Npad, w >> 1, w & 1, 1, false, act, slope); break;
      case 64: dq_dense<2, 1, 8>(X, Wp, bias, Kpad, Npad, 0, w, 4, false, act, slope); break;
      case 128: dq_dense<2, 2, DQ_PF>(X, Wp, bias, Kpad, Npad, 0, w, 4, false, act, slope); break;
      case 192: dq_dense<2, 3, DQ_PF>(X, Wp, bias, Kpad, Npad, 0, w, 4, false, act, slope); break;
      default: dq_dense<2, 4, DQ_PF>(X, Wp, bias, Kpad, Npad, 0, w, 4, false, act, slope); break;
    }
  } else {   // 16 rows: one MFMA row tile — half the matrix work per workgroup, twice the workgroups (finer balance over the CUs)
    switch (Npad) {
      case 16: dq_dense<1, 1, 16>(X, Wp, bias, Kpad, Npad, 0, 0, 1, w >= 1, act, slope); break;
      case 32: dq_dense<1, 1, 16>(X, Wp, bias, Kpad, Npad, 0, w & 1, 1, w >= 2, act, slope); break;
      case 64: dq_dense<1, 1, 16>(X, Wp, bias, Kpad, Npad, 0, w, 4, false, act, slope); break;
      case 128: dq_dense<1, 2, 8>(X, Wp, bias, Kpad, Npad, 0, w, 4, false, act, slope); break;
      case 192: dq_dense<1, 3, 5>(X, Wp, bias, Kpad, Npad, 0, w, 4, false, act, slope); break;
      default: dq_dense<1, 4, DQ_PF>(X, Wp, bias, Kpad, Npad, 0, w, 4, false, act, slope); break;
    }
  }
}

__device__ __forceinline__ unsigned long long dq_mix64(unsigned long long seed, unsigned long long key) {
  unsigned long long x = seed * 0x9E3779B97F4A7C15ull + key * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

}  // namespace cim

// Bins the envs with a pending decision by deciding port: lists int32 [P][n_envs] (list p = the envs deciding for port p,
// any order), cnt int32 [64] = list lengths (zero on entry: the forward kernel's last workgroup resets them).  Also writes
// n_actions (1 for a deciding env, else 0) and adds the number of deciding envs to *counter (may be NULL).
extern "C" __global__ void __launch_bounds__(256)
mrx_k_cim_dqn_bin(int n_envs, int P, const int32_t* __restrict__ decisions, int32_t* __restrict__ cnt, int32_t* __restrict__ lists,
                  int32_t* __restrict__ n_actions, unsigned long long* __restrict__ counter, const uint8_t* __restrict__ hint,
                  int32_t* __restrict__ order, int32_t* __restrict__ sched, int sched_per, cim::SamplerRec R) {
  // sched_per > 0: the LAST workgroup builds the order list of the coming step instead (mrx_schedule_block, cim_engine.hip)
  if (sched_per > 0 && blockIdx.x == gridDim.x - 1) {
    mrx_schedule_block(hint, nullptr, 0, n_envs, sched_per & 0xffffff, order, sched, sched_per >> 24);
    return;
  }
  __shared__ int lcnt[64], base[64];
  const int t = threadIdx.x, e = blockIdx.x * blockDim.x + t;
  if (t < 64) lcnt[t] = 0;
  __syncthreads();
  int port = -1, rank = 0;
  if (e < n_envs) {
    const int32_t* d = decisions + (size_t)e * 8;
    bool over = false;
    if (R.on) {
      // the sampler's end-of-episode bookkeeping (eoe |= done of the previous step), and the last element of an env whose episode
      // just ended: its next state is its own state (AbsEnvSampler keeps `_state` unchanged by a final step).  One thread copies
      // the row: it happens once per env and episode.
      const bool was_over = R.eoe[e] != 0;
      over = was_over || R.done[e] != 0;
      if (over) {
        if (!was_over) R.eoe[e] = 1;
        if (R.prev_active[e]) {
          const size_t row = ((size_t)e * R.cap + (size_t)R.prev_j[e]) * R.D;
          if (R.f64) for (int k = 0; k < R.D; k++) ((double*)R.c_next_state)[row + k] = ((const double*)R.c_state)[row + k];
          else for (int k = 0; k < R.D; k++) ((float*)R.c_next_state)[row + k] = ((const float*)R.c_state)[row + k];
          R.prev_active[e] = 0;
        }
      }
    }
    if (!over && d[7] == 1 && (unsigned)d[1] < (unsigned)P) {
      port = d[1];
      rank = atomicAdd(&lcnt[port], 1);
    }
    n_actions[e] = port >= 0 ? 1 : 0;
  }
  __syncthreads();
  if (t < P && lcnt[t] > 0) {
    base[t] = atomicAdd(&cnt[t], lcnt[t]);
    if (counter) atomicAdd(counter, (unsigned long long)lcnt[t]);
  }
  __syncthreads();
  if (port >= 0) lists[(size_t)port * n_envs + base[port] + rank] = e;
}

template <int DQ_TILE>
__device__ __forceinline__ void mrx_dqn_forward_body(const CimParams& K, const cim::DqnParams& M, const int32_t* __restrict__ decisions, int32_t* __restrict__ cnt,
                                                     const int32_t* __restrict__ lists, int32_t* __restrict__ actions, float* __restrict__ q_out,
                                                     float* __restrict__ state_out, int32_t* __restrict__ choice_out, const cim::SamplerRec& R) {
  using namespace cim;
  __shared__ __attribute__((aligned(16))) float X[DQ_TILE * DQ_LD];
  __shared__ int r_env[DQ_TILE], r_node[DQ_TILE][DQ_MAX_NODES], c_info[DQ_MAX_WIDTH], s_tile[3];
  __shared__ int r_slot[DQ_TILE][3];   // transition cache (R.on): the row's new slot, the slot waiting for its next_state, the agent's previous slot
  __shared__ const int32_t* r_frame[DQ_TILE][DQ_MAX_TICKS];
  const int t = threadIdx.x;

  // ---- which (port, tile) is this workgroup: prefix over the ports' tile counts (P <= 64: one wave)
  if (t < 64) {
    const int c = t < K.P ? cnt[t] : 0;
    const int nt = (c + DQ_TILE - 1) / DQ_TILE;
    int incl = nt;
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o);
      if (t >= o) incl += up;
    }
    if (t == 0) s_tile[0] = -1;
    // XCD-aware tile order: the dispatcher places block i on XCD i % 8 (observed, MI355X_MICROARCH.md: a speed assumption only),
    // and each XCD has its own 4 MB L2 while the 22 networks are 8 MB.  The tile list (port 0's tiles, port 1's ...) is cut into
    // 8 equal runs and XCD x works through run x: an XCD then streams about three ports' weights (1 MB) instead of all of them,
    // and every XCD gets the same number of tiles.  (M.xcd_runs = 0: block i takes tile i, the round-3 order.)
    const int total = __shfl(incl, 63);
    int tile_idx = (int)blockIdx.x;
    if (M.xcd_runs) {
      const int run = (total + 7) >> 3, x = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
      tile_idx = slot < run ? x * run + slot : total;   // (past the list: no tile)
    }
    const int b = tile_idx - (incl - nt);
    if (tile_idx < total && b >= 0 && b < nt) {  // exactly one lane matches (or none: more workgroups than tiles)
      s_tile[0] = t;
      s_tile[1] = b * DQ_TILE;
      s_tile[2] = min(DQ_TILE, c - b * DQ_TILE);
    }
    // every workgroup has now read the counters: the last one to say so clears them for the next call
    if (t == 0) {
      int32_t* done = cnt + 64;
      if (atomicAdd(done, 1) == (int)gridDim.x - 1) {
        for (int p = 0; p < 64; p++) cnt[p] = 0;
        *done = 0;
      }
    }
  }
  __syncthreads();
  const int port = s_tile[0];
  if (port < 0) return;
  const int rows = s_tile[2];
  const int32_t* list = lists + (size_t)port * K.n_envs + s_tile[1];
  const int n_ticks = M.look_back - 1;
#ifdef MRX_DQN_PROFILE
  long long tm[12]; int tmi = 0;
#define DQ_MARK() do { __syncthreads(); tm[tmi++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define DQ_MARK() do {} while (0)
#endif
  DQ_MARK();

  // ---- per-row lookups: env, the node list [port] + future_stop_list, the frame of each look-back tick;
  //      per-column descriptors: which (tick, node, frame word) a state column reads
  // what the LAST phase of row t needs (the decision row, two words of the live frame, the cache head, the env's seed) is
  // requested here, by the thread that will use it: those were two more dependent memory latencies after the dense chain
  int32_t dq_d[5] = {0, 0, 0, 0, 0};
  int32_t dq_space = 0, dq_early = 0;
  long long dq_head = 0, dq_seed = 0;
  if (t < DQ_TILE) {
    const int env = t < rows ? list[t] : -1;
    r_env[t] = env;
    if (env >= 0) {
      const int32_t* d = decisions + (size_t)env * 8;
#pragma unroll
      for (int i = 0; i < 5; i++) dq_d[i] = d[i];
      {
        const int32_t* live = K.live + (size_t)env * K.FW;  // the decision's frame is the live frame
        const int v = (unsigned)dq_d[2] < (unsigned)K.V ? dq_d[2] : 0;
        dq_space = live[frame_word(K, 1, VA_REMAINING_SPACE, v, 0)];
        dq_early = live[frame_word(K, 1, VA_EARLY_DISCHARGE, v, 0)];
        if (R.on) dq_head = R.head[env];
        if (M.epsilon > 0.f) dq_seed = K.seed[env];
      }
      // snapshots[tick : vessel : future_stop_list] (cim::stop_list_value, slot by slot): the frame's last_loc_idx is requested
      // from the live frame — which the decision's frame is — TOGETHER with frame_of's own loads and the route words
      const int32_t* livef = K.live + (size_t)env * K.FW;
      const int vv = (unsigned)dq_d[2] < (unsigned)K.V ? dq_d[2] : 0;
      const int kw = frame_word(K, 1, VA_LAST_LOC_IDX, vv, 0);
      const int k_live = livef[kw];
      const int Lr = K.v_route_len[vv], rb = K.v_route_base[vv], start = K.v_start[vv];
      const int32_t* now = frame_of(K, env, dq_d[0]);
      r_node[t][0] = d[1];
      r_node[t][DQ_MAX_NODES - 1] = d[2];
      if (R.on) {
        const long long ring = (long long)R.cap - 1;
        const int agent = d[1] < 0 ? 0 : (d[1] >= R.P ? R.P - 1 : d[1]);
        const long long prev = R.last[(size_t)env * R.P + agent];
        r_slot[t][0] = (int)(dq_head & ring);
        r_slot[t][1] = R.prev_active[env] ? (int)R.prev_j[env] : -1;
        r_slot[t][2] = prev >= 0 ? (int)(prev & ring) : -1;
      }
      {
        const int k = now == livef ? k_live : now ? now[kw] : 0;
        int x = (start + k) % Lr;
        for (int j = 1; j < M.n_nodes; j++) {
          x = (x + 1 == Lr) ? 0 : x + 1;
          r_node[t][j] = now ? (int)K.route_port[rb + x] : 0;
        }
      }
    }
  }
  for (int i = t; i < DQ_TILE * n_ticks; i += blockDim.x) {
    const int r = i / n_ticks, ti = i - r * n_ticks;
    const int env = r < rows ? list[r] : -1;
    const int32_t* f = nullptr;
    if (env >= 0) {
      const int tick = decisions[(size_t)env * 8];
      f = frame_of(K, env, max(0, tick - ti));  // ticks = [max(0, tick - rt) for rt in range(look_back - 1)]
    }
    r_frame[r][ti] = f;
  }
  const int kp0 = M.kpad[0], per_tick = M.n_nodes * M.n_pa, n_port_feats = n_ticks * per_tick;
  if (t < kp0) {
    int info = -1;  // padding column
    if (t < n_port_feats) {
      const int ti = t / per_tick, rem = t - ti * per_tick, ni = rem / M.n_pa, a = M.pa[rem - ni * M.n_pa];
      info = ti | (ni << 8) | ((a == PA_TRANSFER_COST ? 1 : 0) << 15) | ((K.f_ports + a * K.P) << 16);
    } else if (t < M.state_dim) {
      info = 0 | (0x7f << 8) | (M.va[t - n_port_feats] << 16);  // vessel attribute of the decision's frame
    }
    c_info[t] = info;
  }
  __syncthreads();
  DQ_MARK();

  // ---- state rows (float32, as FullyConnected.forward's x.float()): ports[ticks : nodes : port_attrs] then
  //      vessels[tick : vessel : vessel_attrs]; thread = column, independent loads in flight per thread
  if (t < kp0) {
    const int info = c_info[t];
    const int ti = info & 0xff, ni = (info >> 8) & 0x7f, word = info >> 16;
    const bool is_f32 = (info >> 15) & 1, vessel_col = ni == 0x7f;
    int32_t raw[DQ_TILE];
    unsigned okm = 0;
#pragma unroll
    for (int r = 0; r < DQ_TILE; r++) {  // all the tile's loads of this column in flight at once
      const int32_t* f = r_frame[r][ti];
      const int node = vessel_col ? r_node[r][DQ_MAX_NODES - 1] : r_node[r][ni];  // last slot: the deciding vessel
      const bool ok = r_env[r] >= 0 && info >= 0 && f != nullptr && (unsigned)node < (unsigned)(vessel_col ? K.V : K.P);
      const int w = vessel_col ? frame_word(K, 1, word, node, 0) : word + node;
      raw[r] = *(ok ? f + w : K.live);  // always a valid address: no branch around the load
      okm |= (unsigned)ok << r;
    }
#pragma unroll
    for (int r = 0; r < DQ_TILE; r++) {
      const float v = !((okm >> r) & 1) ? 0.f : is_f32 ? bits_f(raw[r]) : (float)raw[r];
      if (state_out && r_env[r] >= 0 && info >= 0) state_out[(size_t)r_env[r] * M.state_dim + t] = v;
      if (R.on && r_env[r] >= 0 && info >= 0) {
        // the transition's state row, the previous element's next state, the agent's previous element's next agent state:
        // consecutive threads write consecutive words of each row
        const size_t base = (size_t)r_env[r] * R.cap;
        rec_store(R.c_state, (base + r_slot[r][0]) * R.D + t, v, R.f64);
        if (r_slot[r][1] >= 0) rec_store(R.c_next_state, (base + r_slot[r][1]) * R.D + t, v, R.f64);
        if (r_slot[r][2] >= 0) rec_store(R.c_nas, (base + r_slot[r][2]) * R.D + t, v, R.f64);
      }
      X[r * DQ_LD + t] = v;
    }
  }
  __syncthreads();

  // ---- the dense chain of this port's network
  const float* net = M.weights + (size_t)port * M.net_floats;
  DQ_MARK();
  for (int l = 0; l < M.n_layers; l++) {
    dq_layer<DQ_TILE>(X, net + M.w_off[l], net + M.b_off[l], M.kpad[l], M.npad[l], l + 1 < M.n_layers, M.slope);
    DQ_MARK();
  }

  // ---- q = adv - mean(adv) + v (dqn.py:48-52), greedy action, env_sampler.py:33-64 translation
  if (t < rows) {
    const int env = r_env[t];
    const float* y = X + t * DQ_LD;
    const int A = M.n_actions;
    float mean = 0.f;
    if (M.dueling) {
      for (int a = 0; a < A; a++) mean += y[a];
      mean = mean / (float)A - y[A];
    }
    int best = 0;
    float bq = y[0] - mean;
    for (int a = 0; a < A; a++) {
      const float q = y[a] - mean;
      if (q_out) q_out[(size_t)env * A + a] = q;
      if (q > bq) { bq = q; best = a; }
    }
    const int32_t* d = dq_d;  // (requested with the row lookups)
    if (M.epsilon > 0.f) {  // counter-based epsilon-greedy keyed on (env seed, tick, vessel)
      const unsigned long long x = dq_mix64((unsigned long long)dq_seed, (((unsigned long long)(unsigned)d[0] << 8) | (unsigned)d[2]) + 0x200000000ull);
      if ((double)(x >> 11) * (1.0 / 9007199254740992.0) < (double)M.epsilon) best = (int)(dq_mix64(x, 1) % (unsigned long long)A);
    }
    if (choice_out) choice_out[env] = best;
    const double percent = fabs(M.action_space[best]);
    const double load = (double)d[3], discharge = (double)d[4];
    const bool is_load = 2 * best < A;  // model_action < len(action_space) / 2
    double qty;
    if (is_load) {
      qty = fmin(rint(percent * load), (double)dq_space);
    } else {
      const double early = (double)dq_early;
      const double plan = percent * (discharge + early) - early;
      qty = plan > 0 ? rint(plan) : rint(percent * discharge);
    }
    int32_t* a = actions + (size_t)env * K.max_actions * 4;
    a[0] = d[2];
    a[1] = d[1];
    a[2] = (int32_t)qty;
    a[3] = is_load ? MRX_ACTION_LOAD : MRX_ACTION_DISCHARGE;
    if (R.on) {   // the scalar half of the transition (mrx_k_cim_sampler_record's lane 0)
      const int agent = d[1] < 0 ? 0 : (d[1] >= R.P ? R.P - 1 : d[1]);
      const long long q = dq_head;
      const size_t base = (size_t)env * R.cap, ci = base + r_slot[t][0];
      R.c_tick[ci] = d[0];
      R.c_agent[ci] = agent;
      R.c_action[ci] = best;
      R.c_terminal[ci] = 0;
      if (r_slot[t][2] >= 0) R.c_terminal[base + r_slot[t][2]] = 0;
      R.c_env_action[ci * 4 + 0] = a[0]; R.c_env_action[ci * 4 + 1] = a[1]; R.c_env_action[ci * 4 + 2] = a[2]; R.c_env_action[ci * 4 + 3] = a[3];
      R.last[(size_t)env * R.P + agent] = q;
      R.head[env] = q + 1;
      R.prev_j[env] = r_slot[t][0];
      R.prev_active[env] = 1;
      R.steps_env[env] += 1;
    }
  }
#ifdef MRX_DQN_PROFILE
  DQ_MARK();
  if (t == 0 && q_out) for (int i = 0; i + 1 < tmi; i++) q_out[(size_t)blockIdx.x * 16 + i] = (float)(tm[i + 1] - tm[i]);  // overwrites q rows: profiling build only
#endif
#undef DQ_MARK
}

extern "C" __global__ void __launch_bounds__(256)
mrx_k_cim_dqn_forward(CimParams K, cim::DqnParams M, const int32_t* __restrict__ decisions, int32_t* __restrict__ cnt,
                      const int32_t* __restrict__ lists, int32_t* __restrict__ actions, float* __restrict__ q_out,
                      float* __restrict__ state_out, int32_t* __restrict__ choice_out, cim::SamplerRec R) {
  mrx_dqn_forward_body<32>(K, M, decisions, cnt, lists, actions, q_out, state_out, choice_out, R);
}

// 16-env tiles: the same kernel with one MFMA row tile per workgroup
extern "C" __global__ void __launch_bounds__(256)
mrx_k_cim_dqn_forward16(CimParams K, cim::DqnParams M, const int32_t* __restrict__ decisions, int32_t* __restrict__ cnt,
                        const int32_t* __restrict__ lists, int32_t* __restrict__ actions, float* __restrict__ q_out,
                        float* __restrict__ state_out, int32_t* __restrict__ choice_out, cim::SamplerRec R) {
  mrx_dqn_forward_body<16>(K, M, decisions, cnt, lists, actions, q_out, state_out, choice_out, R);
}
