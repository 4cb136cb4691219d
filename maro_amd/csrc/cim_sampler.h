// cim_sampler.h — the END of a batched EnvSampler call on the device (mrx_cim_sampler_finalize / mrx_cim_sampler_emit_all):
// what `AbsEnvSampler.sample` does after its interaction loop (maro/rl/rollout/env_sampler.py:512-530) for every env of a batch —
// `_append_cache_element(None)`, the emission bound `tick <= env.tick - reward_eval_delay`, the delayed rewards
// (examples/cim/rl/env_sampler.py:65-80), popping the emitted prefix — in three launches and one 32-byte read-back, instead of a
// sequence of masked tensor ops with several host synchronisations (maro_amd/cim/sampler.py::_finalize_and_emit stays as the
// specification and serves the rare mid-call roll-overs).  HIP only.
#pragma once
#include <hip/hip_runtime.h>

#include "cim_dqn.h"

namespace cim {

struct SamplerEnd {
  SamplerRec R;          // the cache
  long long* tail;       // [n] number of the oldest element still cached
  int window, frames;    // reward window = reward_eval_delay (ticks); frames per env of the port history
  double ff, sf;         // fulfillment / shortage factors
  const double* decay;   // [window]
  int32_t* hist;         // int32 [n][frames][2][P]: (fulfillment, shortage) rows, written by the step kernel at every snapshot
};

}  // namespace cim

// One wave per env.  (1) eoe |= done.  (2) An env whose episode is over and whose last element still waits for its next state gets
// its own state (the binning launch of the NEXT step would do it, but the call ends here).  (3) An env that is paused at a decision
// has its current tick's retention row filled in from the live frame (the row is only written when the tick completes; the
// pre-decision snapshot of core.py:345 is the live frame).  (4) n_emit[e] = the env's cached elements, oldest first, whose tick is
// <= env.tick - window (ticks are non-decreasing: a prefix).
extern "C" __global__ void __launch_bounds__(64)
mrx_k_cim_sampler_finalize(CimParams K, cim::SamplerEnd E, long long* __restrict__ n_emit) {
  const cim::SamplerRec& R = E.R;
  const int e = (int)blockIdx.x, lane = (int)threadIdx.x;
  const bool was_over = R.eoe[e] != 0;
  const bool over = was_over || R.done[e] != 0;
  if (over && !was_over && lane == 0) R.eoe[e] = 1;
  const int tick_now = K.tick[e];
  if (over && R.prev_active[e]) {
    const size_t row = ((size_t)e * R.cap + (size_t)R.prev_j[e]) * R.D;
    if (R.f64) for (int k = lane; k < R.D; k += 64) ((double*)R.c_next_state)[row + k] = ((const double*)R.c_state)[row + k];
    else for (int k = lane; k < R.D; k += 64) ((float*)R.c_next_state)[row + k] = ((const float*)R.c_state)[row + k];
  }
  if (!over && tick_now >= 0 && tick_now < E.frames) {
    const int32_t* live = K.live + (size_t)e * K.FW + K.f_ports;
    int32_t* row = E.hist + ((size_t)e * E.frames + (size_t)tick_now) * 2 * K.P;
    if (lane < K.P) { row[lane] = live[PA_FULFILLMENT * K.P + lane]; row[K.P + lane] = live[PA_SHORTAGE * K.P + lane]; }
  }
  // prefix of elements old enough: lanes test 64 consecutive elements at a time
  const long long head = R.head[e], tail = E.tail[e], ring = (long long)R.cap - 1;
  const long long bound = (long long)tick_now - E.window;
  long long cnt = 0;
  for (long long q0 = tail; q0 < head; q0 += 64) {
    const long long q = q0 + lane;
    const bool ok = q < head && (long long)R.c_tick[(size_t)e * R.cap + (size_t)(q & ring)] <= bound;
    const unsigned long long m = __ballot(ok);
    if (m == ~0ull) { cnt += 64; continue; }
    cnt += __builtin_ctzll(~m);   // first element that is too young (or past the head)
    break;
  }
  __syncthreads();   // (every lane has read prev_active)
  if (lane == 0) {
    n_emit[e] = cnt;
    if (over) R.prev_active[e] = 0;
  }
}

// Exclusive scan of n_emit over the envs (one workgroup; n_envs up to a few hundred thousand) and the call's read-back:
// info[0] = experiences emitted, info[1] = most elements any env keeps cached afterwards, info[2] = envs at the end of their
// episode, info[3] = running envs whose NEWEST element is being emitted although it still waits for its next state (the gap to the
// pending decision is >= the reward window: the reference sets next_state right after _step, env_sampler.py:494-497, so the host
// completes those elements — one state evaluation, CimBatchSampler._fill_next_state — before it launches the emission).
extern "C" __global__ void __launch_bounds__(1024)
mrx_k_cim_sampler_scan(int n, const long long* __restrict__ n_emit, const long long* __restrict__ head, const long long* __restrict__ tail,
                       const uint8_t* __restrict__ eoe, const uint8_t* __restrict__ prev_active, long long* __restrict__ out_off,
                       long long* __restrict__ info) {
  __shared__ long long part[1024], pmax[1024], pover[1024], ppend[1024];
  const int t = (int)threadIdx.x, per = (n + 1023) / 1024, lo = t * per, hi = min(n, lo + per);
  long long s = 0, mx = 0, ov = 0, pe = 0;
  for (int e = lo; e < hi; e++) {
    s += n_emit[e];
    const long long left = head[e] - tail[e] - n_emit[e];
    mx = left > mx ? left : mx;
    ov += eoe[e] != 0;
    pe += (n_emit[e] > 0 && left == 0 && eoe[e] == 0 && prev_active[e] != 0) ? 1 : 0;
  }
  part[t] = s; pmax[t] = mx; pover[t] = ov; ppend[t] = pe;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {   // inclusive scan of the partial sums (Hillis-Steele)
    const long long v = t >= o ? part[t - o] : 0;
    const long long m2 = t >= o ? pmax[t - o] : 0, o2 = t >= o ? pover[t - o] : 0, p2 = t >= o ? ppend[t - o] : 0;
    __syncthreads();
    part[t] += v; pmax[t] = m2 > pmax[t] ? m2 : pmax[t]; pover[t] += o2; ppend[t] += p2;
    __syncthreads();
  }
  long long run = part[t] - s;
  for (int e = lo; e < hi; e++) { out_off[e] = run; run += n_emit[e]; }
  if (t == 1023) { info[0] = part[t]; info[1] = pmax[t]; info[2] = pover[t]; info[3] = ppend[t]; }
}

// Emission: one 256-thread workgroup per env.  The env's n_emit oldest elements occupy CONSECUTIVE ring slots (two pieces when the
// ring wraps), and their experience rows are consecutive in the outputs — so the three state arrays move as plain contiguous word
// copies by all 256 threads.  The delayed reward of an element sums `window` history rows of its port; an env's elements read
// overlapping ranges of ITS history, which is contiguous per env: the range [first tick + 1, last tick + window] is staged in LDS
// once (in chunks of `rows_cap` ticks) and every element's sum comes out of LDS (float64, lanes over the window, one reduction per
// element and wave).  `_append_cache_element(None)` is applied lazily: an emitted element that is still its agent's LAST element
// gets terminal = end_of_episode and next_agent_state = its own state here (env_sampler.py:404-410).  Finally the emitted prefix
// is popped (tail += n_emit; an agent whose last element went out has none).
template <class T>
__global__ void __launch_bounds__(256)
mrx_k_cim_sampler_emit_all(cim::SamplerEnd E, int rows_cap, const long long* __restrict__ n_emit, const long long* __restrict__ out_off,
                           T* __restrict__ o_state, long long* __restrict__ o_action, int32_t* __restrict__ o_env_action,
                           float* __restrict__ o_reward, T* __restrict__ o_next_state, T* __restrict__ o_nas, uint8_t* __restrict__ o_terminal,
                           int32_t* __restrict__ o_env_id, int32_t* __restrict__ o_tick, int32_t* __restrict__ o_agent) {
  extern __shared__ int32_t hrows[];   // [rows_cap][2 P]
  const cim::SamplerRec& R = E.R;
  const int e = (int)blockIdx.x, t = (int)threadIdx.x, w = t >> 6, lane = t & 63;
  const long long ne = n_emit[e];
  if (ne <= 0) return;
  const long long t0 = E.tail[e], o0 = out_off[e], ring = (long long)R.cap - 1;
  const int P = R.P, D = R.D;
  const size_t base = (size_t)e * R.cap;
  const T* cs = (const T*)R.c_state;
  const T* cn = (const T*)R.c_next_state;
  const T* ca = (const T*)R.c_nas;
  const bool eoe = R.eoe[e] != 0;
  // ---- the three row blocks: piece 1 = slots [s0, min(cap, s0 + ne)), piece 2 = the wrapped rest
  const long long s0 = t0 & ring;
  const long long n1 = (s0 + ne <= (long long)R.cap) ? ne : (long long)R.cap - s0;
  for (int piece = 0; piece < 2; piece++) {
    const long long cnt = piece == 0 ? n1 : ne - n1;
    if (cnt <= 0) continue;
    const size_t src = (base + (size_t)(piece == 0 ? s0 : 0)) * D, dst = (size_t)(o0 + (piece == 0 ? 0 : n1)) * D;
    const size_t words = (size_t)cnt * D;
    for (size_t i = t; i < words; i += 256) {
      o_state[dst + i] = cs[src + i];
      o_next_state[dst + i] = cn[src + i];
      o_nas[dst + i] = ca[src + i];
    }
  }
  // ---- scalars, the lazy last-element rule, rewards
  long long j = 0;
  while (j < ne) {   // chunks of elements whose reward windows fit the LDS rows
    const int tick_first = R.c_tick[base + (size_t)((t0 + j) & ring)];
    long long j1 = j + 1;
    // (uniform) the longest run of elements whose ticks stay within rows_cap - window of the chunk's first tick
    while (j1 < ne && R.c_tick[base + (size_t)((t0 + j1) & ring)] - tick_first + E.window <= rows_cap) j1++;
    const int row0 = tick_first + 1;
    const int tick_last = R.c_tick[base + (size_t)((t0 + j1 - 1) & ring)];
    int nrows = tick_last + E.window - tick_first;
    nrows = nrows > rows_cap ? rows_cap : nrows;      // (a single element always fits: rows_cap >= window, checked by the host)
    __syncthreads();
    for (int i = t; i < nrows * 2 * P; i += 256) {
      const int r = row0 + i / (2 * P);
      hrows[i] = r < E.frames ? E.hist[((size_t)e * E.frames + (size_t)r) * 2 * P + (i - (i / (2 * P)) * 2 * P)] : 0;   // beyond the episode: zeros
    }
    __syncthreads();
    for (long long k = j + w; k < j1; k += 4) {
      const size_t ci = base + (size_t)((t0 + k) & ring), oi = (size_t)(o0 + k);
      const int tick = R.c_tick[ci];
      int agent = (int)R.c_agent[ci];
      agent = agent < 0 ? 0 : (agent >= P ? P - 1 : agent);
      double af = 0.0, as = 0.0;
      for (int x = lane; x < E.window; x += 64) {
        const int r = tick - tick_first + x;     // row of tick + 1 + x
        const double dk = E.decay[x];
        af += dk * (double)hrows[r * 2 * P + agent];
        as += dk * (double)hrows[r * 2 * P + P + agent];
      }
      for (int off = 32; off; off >>= 1) { af += __shfl_down(af, off, 64); as += __shfl_down(as, off, 64); }
      const bool is_last = R.last[(size_t)e * P + agent] == t0 + k;
      if (is_last) {   // next agent state = the element's own state (the block copy above wrote the stored one)
        for (int d = lane; d < D; d += 64) o_nas[oi * D + d] = cs[ci * D + d];
      }
      if (lane < 4) o_env_action[oi * 4 + lane] = R.c_env_action[ci * 4 + lane];
      if (lane == 0) {
        o_reward[oi] = (float)(E.ff * af - E.sf * as);
        o_action[oi] = R.c_action[ci];
        o_terminal[oi] = is_last ? (uint8_t)eoe : R.c_terminal[ci];
        o_env_id[oi] = e;
        o_tick[oi] = tick;
        o_agent[oi] = agent;
      }
    }
    j = j1;
  }
  // ---- pop the emitted prefix
  __syncthreads();
  const long long new_tail = t0 + ne;
  if (t < P) { const long long li = R.last[(size_t)e * P + t]; if (li < new_tail) R.last[(size_t)e * P + t] = -1; }
  if (t == 0) E.tail[e] = new_tail;
}
