// cim_dqn.h — on-device action selection for the CIM RL example: for every env that
// pauses at a decision, build the CIMEnvSampler state from the snapshot ring (examples/cim/rl/env_sampler.py:15-31),
// evaluate the deciding port's dueling DQN (examples/cim/rl/algorithms/dqn.py:13-84, maro/rl/model/fc_block.py:72-133),
// take the greedy action and translate it into an env Action (env_sampler.py:33-64) — two launches, no host round trip.
//
// This is the one dense-contraction piece next to the simulator, so it runs on the matrix cores: exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate — the reference networks are f32), envs binned by port so that a
// workgroup multiplies a tile of envs by ONE port's weights.  Round 6 split the work by what bounds it:
//   mrx_k_cim_dqn_prep     ONE launch, three kinds of workgroup:
//                            bin     256 envs each: the deciding envs appended to their port's list (LDS histogram + one global
//                                    atomic per port and workgroup), the sampler's end-of-episode bookkeeping
//                            (sched) one workgroup builds the order list of the coming step's sorted launch (cim_engine.hip)
//                            state   4 envs each, ONE WAVE PER ENV: the env's state row gathered from the snapshot ring — a chain
//                                    of four dependent memory round trips (decision -> frames / route -> stops -> cells) that
//                                    thousands of resident waves hide — written as a contiguous zero-padded f32 row, plus the
//                                    transition cache's three copies of it
//   mrx_k_cim_dqn_mlp      one workgroup (4 waves) per 16- or 32-env tile of one port's list: the rows arrive as 16-byte loads,
//                          the dense chain runs layer by layer between two LDS activation buffers (one LDS-only barrier per
//                          layer); each wave reads ITS share of the net's weights as one contiguous stream of 1 KB fragments
//                          (packed in consumption order, below) through a register ring two passes deep that never stops at
//                          a layer boundary; then argmax + action translation + the scalar half of the transition.
// HIP only (not part of the CPU wave emulator build).
#pragma once
#include <hip/hip_runtime.h>

#include "cim_device.h"

namespace cim {

enum { DQ_MAX_LAYERS = 8, DQ_MAX_WIDTH = 256, DQ_TILE_MAX = 32, DQ_LD = DQ_MAX_WIDTH + 4, DQ_MAX_TICKS = 16, DQ_MAX_NODES = 8,
       DQ_PASS = 8 /* weight fragments (16 bytes per lane, 1 KB per wave) a pass of the k loop consumes */,
       DQ_SLOTS = 2 * DQ_PASS /* fragments a wave keeps in flight: 16 KB, two passes ahead of the MFMAs */ };

struct DqnParams {
  int n_layers, dueling, state_dim, look_back, n_nodes, n_pa, n_va, n_actions;
  int xcd_runs;  // 1: XCD-aware tile order (launch configuration, set by the host)
  int pa[8], va[8];
  int kpad[DQ_MAX_LAYERS], npad[DQ_MAX_LAYERS], n_out[DQ_MAX_LAYERS];
  long long s_off[4][DQ_MAX_LAYERS];   // where wave w's fragments of layer l start (floats from the net's first)
  long long b_off[DQ_MAX_LAYERS], net_floats;
  float slope, epsilon;
  const float* weights;
  double action_space[32];
};

// The batched EnvSampler's transition cache as the policy launches see it (mrx_cim_collect_steps): when `on`, the binning
// workgroups retire the envs whose episode is over, the state waves write each deciding env's state row into the cache — they
// hold it in registers anyway — and the MLP kernel's last phase APPENDS the scalar half of the transition, instead of a separate
// cache-update launch reading the state back from HBM (mrx_cim_sampler_record, whose semantics these pieces reproduce exactly;
// maro/rl/rollout/env_sampler.py:404-410, 484-511).  An env's cache is a ring of `cap` slots (a power of two): element number q
// lives in slot q & (cap - 1); `last` holds element numbers (-1: none), `prev_j` the SLOT the env's previous step wrote,
// `prev_active` whether that element still waits for its next_state.
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

// padded widths: a layer's outputs are split over the 4 waves in 16-column MFMA tiles — wave w takes tiles w, w + 4, ... (NT = 1, 2
// or 4 of them); a 32-wide layer keeps waves 0 and 1 busy, a 16-wide one wave 0
__host__ __device__ inline int dq_npad(int n) { return n <= 16 ? 16 : n <= 32 ? 32 : n <= 64 ? 64 : n <= 128 ? 128 : 256; }
__host__ __device__ inline int dq_kpad(int k) { return (k + 15) / 16 * 16; }
__host__ __device__ inline int dq_sh(int Npad) { return Npad <= 64 ? 0 : Npad == 128 ? 1 : 2; }   // log2(NT)
__host__ __device__ inline bool dq_idle(int Npad, int w) { return Npad == 16 ? w >= 1 : Npad == 32 ? w >= 2 : false; }
// fragments of a layer in wave w's stream: k-blocks x NT, rounded up to whole passes (the padding is never consumed)
__host__ __device__ inline int dq_frags(int Kpad, int Npad, int w) {
  return dq_idle(Npad, w) ? 0 : ((((Kpad >> 4) << dq_sh(Npad)) + DQ_PASS - 1) / DQ_PASS) * DQ_PASS;
}

// Packed weight layout: a FRAGMENT is 16 inputs (k-block kb) x the 16 output columns of one tile, 256 floats, float
// [(k >> 2) & 3][col & 15][k & 3] — MFMA lane (col & 15, g) fetches its operands of four consecutive k-steps with one 16-byte
// load and a wave reads 1 KB contiguous (the activations use the same k permutation: one ds_read_b128 per lane).  A wave's
// fragments of a layer are stored in the order it consumes them (k-block major, its column tiles minor), the layers one after
// another, the four waves' streams one after another: a wave reads ONE contiguous stream from the first layer's first fragment
// to the last layer's last, whatever the layer boundaries.  dq_w_slot: (wave, fragment within the layer, float within the fragment).
__host__ __device__ inline void dq_w_slot(int k, int n, int Npad, int* w, int* frag, int* within) {
  const int tile = n >> 4, sh = dq_sh(Npad);
  int nt = 0;
  if (Npad == 16) *w = 0; else if (Npad == 32) *w = tile; else { *w = tile & 3; nt = tile >> 2; }
  *frag = ((k >> 4) << sh) + nt;
  *within = ((k >> 2) & 3) * 64 + (n & 15) * 4 + (k & 3);   // = lane (g * 16 + col) * 4 + s: a wave's load is lane-contiguous
}

typedef float dq_f4 __attribute__((ext_vector_type(4)));

// A 16-byte load per lane from a WAVE-UNIFORM byte offset into the net + a per-lane byte offset, as a buffer load
// (buffer_load_dwordx4 v, v_lane_off, s[rsrc], s_off offen): the descriptor and the stream offset stay in scalar registers, so a
// fragment costs one scalar add and one load — hipcc turns the same thing written with pointers into a 64-bit vector address per load.
typedef int dq_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dq_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);   // (raw buffer, no swizzle; gfx9 word 3)
}
__device__ __forceinline__ dq_f4 dq_ldw(__amdgpu_buffer_rsrc_t rs, int soff_bytes, unsigned lane_off) {
  return __builtin_bit_cast(dq_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_off, soff_bytes, 0));
}

// A workgroup barrier that orders LDS traffic only: __syncthreads() also drains every global load in flight (vmcnt(0)), which
// would throw away the weights requested ahead for the next layer.
__device__ __forceinline__ void dq_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the dense chain of one wave.  The wave's share of the network's weights is ONE contiguous stream of fragments (layout
// above), consumed in PASSES of DQ_PASS fragments: 8 / NT k-blocks of one layer (a layer's last pass may be partial: its padding
// fragments are fetched and skipped).  The fragments live in a ring of 2 x DQ_PASS static register slots; consuming slot p of a pass
// refills it with the fragment 2 passes further down the stream — the same layer's next k-blocks or the next layer's first:
// weights do not depend on activations, so the stream never stops at a layer boundary, a layer starts with its weights in
// registers, and a refill is one 64-bit pointer bump per pass + one load per fragment (immediate offsets).  Every load of a pass
// is unconditional, so the compiler waits for a slot with vmcnt(number of younger loads), never vmcnt(0).
//
// One pass (template: MT row tiles, NT column tiles, H = which half of the ring): k-blocks of a layer whose input rows start at
// `xk` (the lane's row, k group and the pass's first k-block), `nvalid` fragments to consume, `refill` = the stream 2 passes ahead.
// The MFMA computes the TRANSPOSED tile (weights as the A operand, activations as B): a lane then holds four CONSECUTIVE output
// columns of one row — one 16-byte LDS store per tile in the epilogue.  `acur` holds the activations of the pass's first k-block
// on entry and of the next pass's first block on exit (requested from LDS one block ahead; the block after a layer's last is
// read and dropped).
template <int MT, int NT, int H, bool FULL>
__device__ __forceinline__ void dq_pass(dq_f4 (&ring)[DQ_SLOTS], dq_f4 (&acc)[DQ_TILE_MAX / 16][4], dq_f4 (&acur)[DQ_TILE_MAX / 16], const float* xk, int nvalid,
                                        __amdgpu_buffer_rsrc_t rs, int refill, unsigned lane_off) {
  dq_f4 an_[MT];
#pragma unroll
  for (int p = 0; p < DQ_PASS; p++) {
    const int kbl = p / NT, nt = p % NT;
    if (nt == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++) an_[mt] = *(const dq_f4*)(xk + mt * 16 * DQ_LD + (kbl + 1) * 16);
    }
    const dq_f4 b_ = ring[H * DQ_PASS + p];
    ring[H * DQ_PASS + p] = dq_ldw(rs, refill + p * 1024, lane_off);
    if (FULL || p < nvalid) {   // (FULL: a pass of 8 valid fragments is straight-line code; the partial form branches per fragment)
#pragma unroll
      for (int s = 0; s < 4; s++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b_[s], acur[mt][s], acc[mt][nt], 0, 0, 0);
    }
    if (nt == NT - 1) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++) acur[mt] = an_[mt];
    }
  }
}

template <int MT, int NT, int H>
__device__ __forceinline__ void dq_pass_any(dq_f4 (&ring)[DQ_SLOTS], dq_f4 (&acc)[DQ_TILE_MAX / 16][4], dq_f4 (&acur)[DQ_TILE_MAX / 16], const float* xk, int nvalid,
                                            __amdgpu_buffer_rsrc_t rs, int refill, unsigned lane_off) {
  if (nvalid >= DQ_PASS) dq_pass<MT, NT, H, true>(ring, acc, acur, xk, nvalid, rs, refill, lane_off);
  else dq_pass<MT, NT, H, false>(ring, acc, acur, xk, nvalid, rs, refill, lane_off);
}

// bias + LeakyReLU, the tile rows written to the other activation buffer; D[i][j]: i = (lane >> 4) * 4 + e = the output column
// within the tile, j = lane & 15 = the row.  Leaves the accumulators zero for the next layer.
template <int MT, int NT>
__device__ __forceinline__ void dq_epilogue(dq_f4 (&acc)[DQ_TILE_MAX / 16][4], const dq_f4 (&bv)[4], float* __restrict__ Xout, int col0, int colstep, bool act, float slope) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    const int col = col0 + nt * colstep + g * 4;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      dq_f4 v = acc[mt][nt] + bv[nt];
      if (act) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
      }
      *(dq_f4*)(Xout + (mt * 16 + r) * DQ_LD + col) = v;
      acc[mt][nt] = dq_f4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// the bias values lane (r, g) adds in a layer's epilogue: four consecutive columns of each of the wave's NT tiles
__device__ __forceinline__ void dq_bias(dq_f4 (&bv)[4], const float* bias, int Npad, int w, int g) {
  const int sh = dq_sh(Npad), col0 = (Npad == 16 ? 0 : Npad == 32 ? (w & 1) : w) * 16, colstep = Npad <= 32 ? 16 : 64;
#pragma unroll
  for (int nt = 0; nt < 4; nt++) bv[nt] = *(const dq_f4*)(bias + col0 + min(nt, (1 << sh) - 1) * colstep + g * 4);
}

__device__ __forceinline__ unsigned long long dq_mix64(unsigned long long seed, unsigned long long key) {
  unsigned long long x = seed * 0x9E3779B97F4A7C15ull + key * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

__device__ __forceinline__ const int32_t* dq_shfl_ptr(const int32_t* p, int src) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = (unsigned)__shfl((int)(unsigned)u, src), hi = (unsigned)__shfl((int)(unsigned)(u >> 32), src);
  return (const int32_t*)(((unsigned long long)hi << 32) | lo);
}

// ---- the state row of ONE env, built by one wave (mrx_k_cim_dqn_prep's state workgroups).  c_info[c] describes state column c
// (built once per workgroup): tick index | node index << 8 (0x7f: the deciding vessel) | f32 flag << 15 | frame word << 16, -1 = padding.
// Writes xrows[env][0 .. kp0) (float32, zero padded) and, for the real columns, state_out / the transition cache's rows.
__device__ __forceinline__ void dq_state_wave(const CimParams& K, const DqnParams& M, int env, const int32_t* __restrict__ decisions,
                                              const int* c_info, float* __restrict__ xrows, float* __restrict__ state_out, const SamplerRec& R) {
  const int lane = threadIdx.x & 63;
  // round trip 1: the decision row, the env's header words, the cache's cursors
  const int dl = lane < 8 ? decisions[(size_t)env * 8 + lane] : 0;
  const int32_t* hdr = K.priv + (size_t)env * K.PW;
  const int flags = hdr[PH_FLAGS], htick = hdr[PH_TICK];
  int over = 0, pact = 0, pj = 0;
  long long head = 0;
  if (R.on) {
    over = (int)R.eoe[env] | (int)R.done[env];
    head = R.head[env];
    pact = R.prev_active[env];
    pj = (int)R.prev_j[env];
  }
  const int tick = __builtin_amdgcn_readlane(dl, 0), port = __builtin_amdgcn_readlane(dl, 1), vessel = __builtin_amdgcn_readlane(dl, 2);
  if (over || __builtin_amdgcn_readlane(dl, 7) != 1 || (unsigned)port >= (unsigned)K.P) return;   // (as the binning workgroups decide)
  // round trip 2: which frame holds each look-back tick (cim::frame_of, lane = tick index), the vessel's route position
  const int32_t* livef = K.live + (size_t)env * K.FW;
  const int vv = (unsigned)vessel < (unsigned)K.V ? vessel : 0;
  const int kw = K.f_vessels + VA_LAST_LOC_IDX * K.V + vv;
  const int k_live = livef[kw];
  const int Lr = K.v_route_len[vv], rb = K.v_route_base[vv], start = K.v_start[vv];
  long long last = -1;
  if (R.on) last = R.last[(size_t)env * R.P + port];
  const int n_ticks = M.look_back - 1;
  const int fi = max(0, tick - lane);   // ticks = [max(0, tick - rt) for rt in range(look_back - 1)]
  const int slot = fi % K.S;
  const int held = lane < n_ticks ? K.ring_fi[(size_t)env * K.S + slot] : -1;
  const bool paused = (flags & (FL_FRESH | FL_FINISHED)) == 0;
  const int cur_fi = (htick - K.start_tick) / K.resolution;
  const int32_t* fptr = nullptr;
  if (held == fi) fptr = K.ring + ((size_t)env * K.S + slot) * K.FW;
  if (paused && slot == cur_fi % K.S) fptr = fi == cur_fi ? livef : nullptr;
  if (lane >= n_ticks) fptr = nullptr;
  // round trip 3: snapshots[tick : vessel : future_stop_list] (cim::stop_list_value), lane j = node j of [port] + future stops
  const int32_t* now = dq_shfl_ptr(fptr, 0);
  now = (const int32_t*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)now >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)now));
  const int k = now == livef ? k_live : now ? now[kw] : 0;
  const int x0 = (start + k) % Lr;
  int node = port;
  if (lane >= 1 && lane < M.n_nodes) node = now ? (int)K.route_port[rb + (x0 + lane) % Lr] : 0;
  // round trip 4: the cells.  Every load of the row is issued before the first conversion.
  const int kp0 = M.kpad[0];
  const size_t cbase = (size_t)env * R.cap;
  const int s0 = (int)(head & ((long long)R.cap - 1)), s1 = pact ? pj : -1, s2 = last >= 0 ? (int)(last & ((long long)R.cap - 1)) : -1;
  int32_t raw[4];
  int inf[4];
  unsigned okm = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = lane + 64 * i;
    const int info = c < kp0 ? c_info[c] : -1;
    inf[i] = info;
    const int ti = info & 0xff, ni = (info >> 8) & 0x7f, word = info >> 16;
    const bool vessel_col = ni == 0x7f;
    const int32_t* f = dq_shfl_ptr(fptr, ti & 63);
    const int nd = vessel_col ? vessel : __shfl(node, ni & 63);
    const bool ok = info >= 0 && f != nullptr && (unsigned)nd < (unsigned)(vessel_col ? K.V : K.P);
    const int w = vessel_col ? K.f_vessels + word * K.V + nd : word + nd;
    raw[i] = *(ok ? f + w : K.live);  // always a valid address: no branch around the load
    okm |= (unsigned)ok << i;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = lane + 64 * i;
    if (c < kp0) {
      const float v = !((okm >> i) & 1) ? 0.f : ((inf[i] >> 15) & 1) ? bits_f(raw[i]) : (float)raw[i];
      xrows[(size_t)env * kp0 + c] = v;
      if (inf[i] >= 0) {
        if (state_out) state_out[(size_t)env * M.state_dim + c] = v;
        if (R.on) {   // the transition's state row, the previous element's next state, the agent's previous element's next agent state
          rec_store(R.c_state, (cbase + s0) * R.D + c, v, R.f64);
          if (s1 >= 0) rec_store(R.c_next_state, (cbase + s1) * R.D + c, v, R.f64);
          if (s2 >= 0) rec_store(R.c_nas, (cbase + s2) * R.D + c, v, R.f64);
        }
      }
    }
  }
}

}  // namespace cim

// The policy's first launch.  Workgroups [0, n_bin): bin the envs with a pending decision by deciding port — lists int32 [P][n_envs]
// (list p = the envs deciding for port p, any order), cnt int32 [64] = list lengths (zero on entry: the MLP kernel's last
// workgroup resets them); also n_actions (1 for a deciding env, else 0) and the number of deciding envs added to *counter (may be
// NULL).  sched_per > 0: workgroup n_bin builds the order list of the coming step (mrx_schedule_block, cim_engine.hip).  The
// remaining workgroups build the state rows, one wave per env (cim::dq_state_wave).
extern "C" __global__ void __launch_bounds__(256)
mrx_k_cim_dqn_prep(CimParams K, cim::DqnParams M, int n_bin, const int32_t* __restrict__ decisions, int32_t* __restrict__ cnt,
                   int32_t* __restrict__ lists, int32_t* __restrict__ n_actions, unsigned long long* __restrict__ counter, int sched_per,
                   float* __restrict__ xrows, float* __restrict__ state_out, cim::SamplerRec R) {
  using namespace cim;
  const int n_envs = K.n_envs, P = K.P;
  const int first_state = n_bin + (sched_per > 0 ? 1 : 0);
  if ((int)blockIdx.x >= first_state) {
    __shared__ int c_info[DQ_MAX_WIDTH];
    const int t = threadIdx.x, kp0 = M.kpad[0];
    const int per_tick = M.n_nodes * M.n_pa, n_port_feats = (M.look_back - 1) * per_tick;
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
    const int env = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - first_state) * 4 + (t >> 6));
    if (env < n_envs) dq_state_wave(K, M, env, decisions, c_info, xrows, state_out, R);
    return;
  }
  if (sched_per > 0 && (int)blockIdx.x == n_bin) {
    mrx_schedule_block((const uint8_t*)K.hint, nullptr, 0, n_envs, sched_per & 0xffffff, K.order, K.sched, sched_per >> 24);
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
      // the row: it happens once per env and episode.  (The state waves of this launch read eoe | done, never eoe alone.)
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
__device__ __forceinline__ void mrx_dqn_mlp_body(const CimParams& K, const cim::DqnParams& M, const int32_t* __restrict__ decisions, int32_t* __restrict__ cnt,
                                                 const int32_t* __restrict__ lists, const float* __restrict__ xrows, int32_t* __restrict__ actions,
                                                 float* __restrict__ q_out, int32_t* __restrict__ choice_out, const cim::SamplerRec& R) {
  using namespace cim;
  __shared__ __attribute__((aligned(16))) float X[2][DQ_TILE * DQ_LD + 16];   // (+16: the k loop reads one block past a row's last)
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform: the weight addresses stay in scalar registers

  // ---- which (port, tile) is this workgroup: prefix over the ports' tile counts (P <= 64: one wave).  EVERY wave works it out for
  //      itself (one load + a wave scan): no LDS round trip and no barrier stand between the kernel's start and its first weights.
  int port, rows, first;
  {
    const int c = lane < K.P ? cnt[lane] : 0;
    const int nt = (c + DQ_TILE - 1) / DQ_TILE;
    int incl = nt;
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    // XCD-aware tile order: the dispatcher places block i on XCD i % 8 (observed, MI355X_MICROARCH.md: a speed assumption only),
    // and each XCD has its own 4 MB L2 while the 22 networks are 8 MB.  The tile list (port 0's tiles, port 1's ...) is cut into
    // 8 equal runs and XCD x works through run x: an XCD then streams about three ports' weights (1 MB) instead of all of them,
    // and every XCD gets the same number of tiles.  (M.xcd_runs = 0: block i takes tile i.)
    const int total = __builtin_amdgcn_readlane(incl, 63);
    int tile_idx = (int)blockIdx.x;
    if (M.xcd_runs) {
      const int run = (total + 7) >> 3, x = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
      tile_idx = slot < run ? x * run + slot : total;   // (past the list: no tile)
    }
    const int b = tile_idx - (incl - nt);
    const unsigned long long hit = __ballot(tile_idx < total && b >= 0 && b < nt);  // exactly one lane matches (or none: more workgroups than tiles)
    port = hit ? __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(hit)) : -1;
    const int src = port < 0 ? 0 : port;
    first = __builtin_amdgcn_readfirstlane(__shfl(b * DQ_TILE, src));
    rows = __builtin_amdgcn_readfirstlane(__shfl(min(DQ_TILE, c - b * DQ_TILE), src));
  }
  if (port < 0) {
    // every wave of every workgroup has read the counters before the last workgroup to say so clears them for the next call
    dq_barrier();
    if (t == 0) {
      int32_t* done = cnt + 64;
      if (atomicAdd(done, 1) == (int)gridDim.x - 1) {
        for (int p = 0; p < 64; p++) cnt[p] = 0;
        *done = 0;
      }
    }
    return;
  }
  const int32_t* list = lists + (size_t)port * K.n_envs + first;
  const float* net = M.weights + (size_t)port * M.net_floats;
  // ---- the weight stream starts before anything else does: the wave's first two passes
  constexpr int MT = DQ_TILE / 16;
  const int r16 = lane & 15, g4 = lane >> 4;
  const unsigned lane_off = (unsigned)lane * 16;
  dq_f4 ring[DQ_SLOTS], bv[4], bvn[4];
  const __amdgpu_buffer_rsrc_t rs = dq_rsrc(net);
  int soff = (int)M.s_off[w][0] * 4;   // the stream (byte offset into the net) at the fragment that slot 0 of the coming pass holds
#pragma unroll
  for (int p = 0; p < DQ_SLOTS; p++) ring[p] = dq_ldw(rs, soff + p * 1024, lane_off);
  int lw = 0;   // the layer the wave computes next (n_layers: none left)
  while (lw < M.n_layers && dq_idle(M.npad[lw], w)) lw++;
  dq_bias(bv, net + M.b_off[lw < M.n_layers ? lw : 0], M.npad[lw < M.n_layers ? lw : 0], w, g4);
#ifdef MRX_DQN_PROFILE
  long long tm[12]; int tmi = 0;
  const long long rt0 = (long long)__builtin_amdgcn_s_memrealtime();   // 100 MHz: what a s_memtime tick is worth under THIS launch
#define DQ_MARK() do { dq_barrier(); tm[tmi++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
  __shared__ long long dq_tl[4][64];   // per-wave timeline: lane 0 stamps s_memtime at the points DQ_T marks (-DMRX_DQN_PROFILE_T)
  int tli = 0;
#ifdef MRX_DQN_PROFILE_T
#define DQ_T() do { if (lane == 0 && tli < 64) dq_tl[w][tli] = (long long)__builtin_amdgcn_s_memtime(); tli++; } while (0)
#else
#define DQ_T() do {} while (0)
#endif
#else
#define DQ_MARK() do {} while (0)
#define DQ_T() do {} while (0)
#endif
  DQ_MARK();

  // ---- what the LAST phase of row t needs (the decision row, two words of the live frame, the cache cursors, the env's seed) is
  //      requested here, by the thread that will use it: nothing after the dense chain waits for memory
  int32_t dq_d[5] = {0, 0, 0, 0, 0};
  int32_t dq_space = 0, dq_early = 0, dq_env = -1;
  long long dq_head = 0, dq_seed = 0, dq_last = -1;
  if (t < rows) {
    const int env = list[t];
    dq_env = env;
    const int32_t* d = decisions + (size_t)env * 8;
#pragma unroll
    for (int i = 0; i < 5; i++) dq_d[i] = d[i];
    const int32_t* live = K.live + (size_t)env * K.FW;  // the decision's frame is the live frame
    const int v = (unsigned)dq_d[2] < (unsigned)K.V ? dq_d[2] : 0;
    dq_space = live[K.f_vessels + VA_REMAINING_SPACE * K.V + v];
    dq_early = live[K.f_vessels + VA_EARLY_DISCHARGE * K.V + v];
    if (R.on) {
      dq_head = R.head[env];
      const int agent = dq_d[1] < 0 ? 0 : (dq_d[1] >= R.P ? R.P - 1 : dq_d[1]);
      dq_last = R.last[(size_t)env * R.P + agent];
    }
    if (M.epsilon > 0.f) dq_seed = K.seed[env];
  }
  // ---- the tile's state rows (written by the state waves of mrx_k_cim_dqn_prep): 16-byte loads, rows past the list zero
  {
    const int kp0 = M.kpad[0], q4 = kp0 >> 2;
    for (int i = t; i < DQ_TILE * q4; i += 256) {
      const int r = i / q4, c4 = i - r * q4;
      dq_f4 v = dq_f4{0.f, 0.f, 0.f, 0.f};
      if (r < rows) v = *(const dq_f4*)(xrows + (size_t)list[r] * kp0 + c4 * 4);
      *(dq_f4*)(&X[0][r * DQ_LD + c4 * 4]) = v;
    }
  }
  dq_barrier();
  DQ_MARK();

  // ---- the dense chain of this port's network, ping-pong between the two activation buffers, one barrier per layer
  int cur = 0, half = 0;
  dq_f4 acc[DQ_TILE_MAX / 16][4], acur[DQ_TILE_MAX / 16];
#pragma unroll
  for (int mt = 0; mt < DQ_TILE_MAX / 16; mt++) {
    acur[mt] = dq_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 4; nt++) acc[mt][nt] = dq_f4{0.f, 0.f, 0.f, 0.f};
  }
  for (int l = 0; l < M.n_layers; l++) {
    DQ_T();
    if (l == lw) {
      const int Npad = M.npad[l], sh = dq_sh(Npad), F = (M.kpad[l] >> 4) << sh;   // fragments to consume
      // the layer after this one (for this wave): its bias is requested NOW — it arrives under this layer's passes and moves into
      // place after the epilogue without a wait
      int ln = l + 1;
      while (ln < M.n_layers && dq_idle(M.npad[ln], w)) ln++;
      const int lb = ln < M.n_layers ? ln : l;
      dq_bias(bvn, net + M.b_off[lb], M.npad[lb], w, g4);
      const float* xlane = X[cur] + r16 * DQ_LD + g4 * 4;
#pragma unroll
      for (int mt = 0; mt < MT; mt++) acur[mt] = *(const dq_f4*)(xlane + mt * 16 * DQ_LD);
      DQ_T();
      for (int f0 = 0; f0 < F; f0 += DQ_PASS) {
        const int nvalid = F - f0;
        const float* xk = xlane + (f0 >> sh) * 16;
        const int refill = soff + DQ_SLOTS * 1024;
        if (half == 0) {
          if (sh == 0) dq_pass_any<MT, 1, 0>(ring, acc, acur, xk, nvalid, rs, refill, lane_off);
          else if (sh == 1) dq_pass_any<MT, 2, 0>(ring, acc, acur, xk, nvalid, rs, refill, lane_off);
          else dq_pass_any<MT, 4, 0>(ring, acc, acur, xk, nvalid, rs, refill, lane_off);
        } else {
          if (sh == 0) dq_pass_any<MT, 1, 1>(ring, acc, acur, xk, nvalid, rs, refill, lane_off);
          else if (sh == 1) dq_pass_any<MT, 2, 1>(ring, acc, acur, xk, nvalid, rs, refill, lane_off);
          else dq_pass_any<MT, 4, 1>(ring, acc, acur, xk, nvalid, rs, refill, lane_off);
        }
        half ^= 1;
        soff += DQ_PASS * 1024;
        DQ_T();
      }
      const bool act = l + 1 < M.n_layers;
      const int col0 = (Npad == 16 ? 0 : Npad == 32 ? (w & 1) : w) * 16, colstep = Npad <= 32 ? 16 : 64;
      if (sh == 0) dq_epilogue<MT, 1>(acc, bv, X[cur ^ 1], col0, colstep, act, M.slope);
      else if (sh == 1) dq_epilogue<MT, 2>(acc, bv, X[cur ^ 1], col0, colstep, act, M.slope);
      else dq_epilogue<MT, 4>(acc, bv, X[cur ^ 1], col0, colstep, act, M.slope);
      DQ_T();
      lw = ln;
#pragma unroll
      for (int nt = 0; nt < 4; nt++) bv[nt] = bvn[nt];
    }
    dq_barrier();  // the layer's output is complete (and every wave has read its input: the buffer may be overwritten by the next layer)
    DQ_T();
    cur ^= 1;
    DQ_MARK();
  }
  // every wave of this workgroup is past its read of the counters: the last workgroup to say so clears them for the next call
  if (t == 0) {
    int32_t* done = cnt + 64;
    if (atomicAdd(done, 1) == (int)gridDim.x - 1) {
      for (int p = 0; p < 64; p++) cnt[p] = 0;
      *done = 0;
    }
  }

  // ---- q = adv - mean(adv) + v (dqn.py:48-52), greedy action, env_sampler.py:33-64 translation
  if (t < rows) {
    const int env = dq_env;
    const float* y = X[cur] + t * DQ_LD;
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
    const int32_t* d = dq_d;
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
    if (R.on) {   // the scalar half of the transition (mrx_k_cim_sampler_record's lane 0); its rows were written by the state wave
      const long long ring = (long long)R.cap - 1;
      const int agent = d[1] < 0 ? 0 : (d[1] >= R.P ? R.P - 1 : d[1]);
      const long long q = dq_head;
      const int s0 = (int)(q & ring), s2 = dq_last >= 0 ? (int)(dq_last & ring) : -1;
      const size_t base = (size_t)env * R.cap, ci = base + s0;
      R.c_tick[ci] = d[0];
      R.c_agent[ci] = agent;
      R.c_action[ci] = best;
      R.c_terminal[ci] = 0;
      if (s2 >= 0) R.c_terminal[base + s2] = 0;
      R.c_env_action[ci * 4 + 0] = a[0]; R.c_env_action[ci * 4 + 1] = a[1]; R.c_env_action[ci * 4 + 2] = a[2]; R.c_env_action[ci * 4 + 3] = a[3];
      R.last[(size_t)env * R.P + agent] = q;
      R.head[env] = q + 1;
      R.prev_j[env] = s0;
      R.prev_active[env] = 1;
      R.steps_env[env] += 1;
    }
  }
#ifdef MRX_DQN_PROFILE
  DQ_MARK();
  if (t == 0 && q_out) {   // profiling build only: the caller's q buffer has room for 16 floats per workgroup behind the n_envs rows
    float* qp = q_out + (size_t)K.n_envs * M.n_actions;
    for (int i = 0; i + 1 < tmi; i++) qp[(size_t)blockIdx.x * 16 + i] = (float)(tm[i + 1] - tm[i]);
    qp[(size_t)blockIdx.x * 16 + 10] = (float)(port * 64 + rows);
    qp[(size_t)blockIdx.x * 16 + 12] = (float)(rt0 & 0xfffff);
    qp[(size_t)blockIdx.x * 16 + 13] = (float)(tm[0] & 0xfffff);
    qp[(size_t)blockIdx.x * 16 + 14] = (float)(tm[tmi - 1] - tm[0]);
    qp[(size_t)blockIdx.x * 16 + 15] = (float)((long long)__builtin_amdgcn_s_memrealtime() - rt0);
    // the per-wave timelines behind the per-workgroup rows: [workgroup][wave][64], ticks since the first mark
    float* qw = qp + (size_t)gridDim.x * 16 + (size_t)blockIdx.x * 256;
    for (int i = 0; i < 256; i++) qw[i] = (i & 63) < tli ? (float)(dq_tl[i >> 6][i & 63] - tm[0]) : -1.f;
  }
#endif
#undef DQ_MARK
#undef DQ_T
}

extern "C" __global__ void __launch_bounds__(256)
mrx_k_cim_dqn_mlp32(CimParams K, cim::DqnParams M, const int32_t* __restrict__ decisions, int32_t* __restrict__ cnt,
                    const int32_t* __restrict__ lists, const float* __restrict__ xrows, int32_t* __restrict__ actions, float* __restrict__ q_out,
                    int32_t* __restrict__ choice_out, cim::SamplerRec R) {
  mrx_dqn_mlp_body<32>(K, M, decisions, cnt, lists, xrows, actions, q_out, choice_out, R);
}

// 16-env tiles: the same kernel with one MFMA row tile per workgroup
extern "C" __global__ void __launch_bounds__(256)
mrx_k_cim_dqn_mlp16(CimParams K, cim::DqnParams M, const int32_t* __restrict__ decisions, int32_t* __restrict__ cnt,
                    const int32_t* __restrict__ lists, const float* __restrict__ xrows, int32_t* __restrict__ actions, float* __restrict__ q_out,
                    int32_t* __restrict__ choice_out, cim::SamplerRec R) {
  mrx_dqn_mlp_body<16>(K, M, decisions, cnt, lists, xrows, actions, q_out, choice_out, R);
}
