// cb_step_kernels.h — the citi_bike reset and step kernels, shared by the generic build (cb_engine.hip) and the plan-specialised
// build (cb_spec.hip, MRX_SPECIALIZED: CD() dimensions are constants).  No include guard on purpose: it only instantiates kernels.
extern "C" __global__ void __launch_bounds__(64)
mrx_k_cb_reset(CbParams K, const int32_t* __restrict__ tt, int n_times, const uint8_t* __restrict__ mask) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= K.n_envs || (mask && !mask[e])) return;
  if (tt)
    for (int i = 0; i < CD(tt_cap); i++) K.tt[CB_IX(CD(aos), CD(stride), CD(tt_cap), i, e)] = i < n_times ? tt[(size_t)e * n_times + i] : 1;
  cb::reset_env(K, e);
}

extern "C" __global__ void __launch_bounds__(64)
mrx_k_cb_step(CbParams K, const int32_t* __restrict__ actions, const int32_t* __restrict__ n_actions, const uint8_t* __restrict__ mask,
              int32_t* __restrict__ decisions, int32_t* __restrict__ scope, long long* __restrict__ metrics, uint8_t* __restrict__ done, int lanes,
              const int32_t* __restrict__ n_answered) {
  // `lanes` envs per wave (mrx_cb_set_lanes_per_wave): lane l < lanes owns env blockIdx.x * lanes + l.  A wave runs the UNION
  // of its lanes' control flow, and steps differ a lot in length, so a small batch is faster spread thin over many waves.
  const int lane = (int)threadIdx.x;
#if defined(MRX_CB_LDSFRAME) && MRXC_lsh_plan >= 0
  lanes = 1 << MRXC_lsh_plan;  // (the host launches this build with the plan's envs per wave only)
#endif
  const int e0 = blockIdx.x * lanes;
  const int e = e0 + lane;
  const bool active = lane < lanes && e < K.n_envs && !(mask && !mask[e]);
#ifdef MRX_CB_LDSFRAME
  // All 64 lanes move the wave's envs' state between HBM and the LDS columns (row w of the `lanes` envs = `lanes` consecutive
  // words): with a few envs per wave the owning lanes alone would issue one narrow load per word of the frame.
  const int cl = lane & (lanes - 1), r0 = lane >> CB_LSH, rstep = 64 >> CB_LSH;
  const bool cok = e0 + cl < K.n_envs;
  const size_t cbase = (size_t)e0 + cl;
#define MRX_CB_LFX(w) cb::mrx_cb_lds[((CB_EV_BLOCK * 4 + (w)) << CB_LSH) + cl]
  if (cok) {
#pragma unroll 4
    for (int w = r0; w < MRXC_FW; w += rstep) MRX_CB_LFX(w) = K.live[CB_IX(CD(aos), CD(stride), CD(FW), w, cbase)];
    for (int w = r0; w < MRXC_S; w += rstep) MRX_CB_LFX(LDS_CAP + w) = K.capacity[w];
    for (int w = r0; w < CH_WORDS; w += rstep) MRX_CB_LFX(LDS_HDR + w) = K.hdr[CB_IX(CD(aos), CD(stride), CH_WORDS, w, cbase)];
#ifdef MRX_CB_TWC_LDS
    for (int w = r0; w < MRXC_ring_slots; w += rstep) {
      MRX_CB_LFX(LDS_TWC + w) = K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, cbase)];
      MRX_CB_LFX(LDS_TWC + MRXC_ring_slots + w) = K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, cbase)];
    }
#endif
    for (int w = r0; w < MRXC_w_words; w += rstep) MRX_CB_LFX(LDS_FUL + w) = (int32_t)K.fulfilled[CB_IX(CD(aos), CD(stride), CD(w_words), w, cbase)];
    for (int w = r0; w < 2 * MRXC_mask_words; w += rstep) MRX_CB_LFX(LDS_DMK + w) = (int32_t)K.decmask[CB_IX(CD(aos), CD(stride), (2 * CD(mask_words)), w, cbase)];
  }
  __syncthreads();
#endif
  if (active) {
    if (CD(decision_mode) == 0) {
      int na = (actions && n_actions) ? n_actions[e] : 0;
      if (na > CD(max_actions)) na = CD(max_actions);
      cb::step_env(K, e, actions ? actions + (size_t)e * CD(max_actions) * 3 : nullptr, na, nullptr, decisions + (size_t)e * 8,
                   scope + (size_t)e * CD(scope_cap) * 2, (int64_t*)metrics + (size_t)e * 3, done + e);
      if (K.obs) cb::write_observation(K, e, decisions + (size_t)e * 8);
    } else {  // Joint modes: S rows per env
      const size_t S = (size_t)CD(S);
      int nans = (actions && n_answered) ? n_answered[e] : 0;
      if (nans < 0) nans = 0;
      cb::step_env(K, e, actions ? actions + (size_t)e * S * CD(max_actions) * 3 : nullptr, nans, n_actions ? n_actions + (size_t)e * S : nullptr,
                   decisions + (size_t)e * S * 8, scope + (size_t)e * S * CD(scope_cap) * 2, (int64_t*)metrics + (size_t)e * 3, done + e);
    }
  }
#ifdef MRX_CB_LDSFRAME
  __syncthreads();
  if (cok) {
#pragma unroll 4
    for (int w = r0; w < MRXC_FW; w += rstep) K.live[CB_IX(CD(aos), CD(stride), CD(FW), w, cbase)] = MRX_CB_LFX(w);
    for (int w = r0; w < CH_WORDS; w += rstep) K.hdr[CB_IX(CD(aos), CD(stride), CH_WORDS, w, cbase)] = MRX_CB_LFX(LDS_HDR + w);
#ifdef MRX_CB_TWC_LDS
    for (int w = r0; w < MRXC_ring_slots; w += rstep) {
      K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, cbase)] = MRX_CB_LFX(LDS_TWC + w);
      K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, cbase)] = MRX_CB_LFX(LDS_TWC + MRXC_ring_slots + w);
    }
#endif
    for (int w = r0; w < MRXC_w_words; w += rstep) K.fulfilled[CB_IX(CD(aos), CD(stride), CD(w_words), w, cbase)] = (uint32_t)MRX_CB_LFX(LDS_FUL + w);
    for (int w = r0; w < 2 * MRXC_mask_words; w += rstep) K.decmask[CB_IX(CD(aos), CD(stride), (2 * CD(mask_words)), w, cbase)] = (uint32_t)MRX_CB_LFX(LDS_DMK + w);
  }
#undef MRX_CB_LFX
#endif
}

// Which envs of this batch step leave their tick (todo[e] = 1: the general kernel's) and which stay inside it (0: the in-tick
// kernel's; masked-out envs too).  One wave per env, reads only: with the answer in K.todo BEFORE either kernel runs, the two
// work on disjoint envs and mrx_cb_step launches them side by side on two streams.
extern "C" __global__ void __launch_bounds__(64)
mrx_k_cb_classify(CbParams K, const int32_t* __restrict__ actions, const int32_t* __restrict__ n_actions, const uint8_t* __restrict__ mask) {
  const int e = (int)blockIdx.x;
  if (mask && !mask[e]) {
    if (threadIdx.x == 0) K.todo[e] = 0;
    return;
  }
  int na = (actions && n_actions) ? n_actions[e] : 0;
  if (na > CD(max_actions)) na = CD(max_actions);
  cb::WavePre P;
  const bool ok = cb::decision_step_wave_pre(K, e, na, P);
  if (threadIdx.x == 0) K.todo[e] = ok ? 0 : 1;
}

// The wave-cooperative decision step (cb_wave.h): ONE env per wave.  Handles the env-steps that stay inside their tick (apply the
// action, next pending station, its action scope across the lanes).  classified = 0: writes todo[e] = 1 for every env it had to
// leave alone, and the general kernel runs AFTER it with `todo` as its env mask.  classified = 1: mrx_k_cb_classify has filled
// K.todo already; this kernel steps the envs it marked 0 and the general kernel the others, at the same time.
extern "C" __global__ void __launch_bounds__(64)
mrx_k_cb_step_wave(CbParams K, const int32_t* __restrict__ actions, const int32_t* __restrict__ n_actions, const uint8_t* __restrict__ mask,
                   int32_t* __restrict__ decisions, int32_t* __restrict__ scope, long long* __restrict__ metrics, uint8_t* __restrict__ done,
                   int classified) {
  __shared__ int32_t scr[2 * cb::CBW_MAX];
  const int e = (int)blockIdx.x;
  if (mask && !mask[e]) {
    if (threadIdx.x == 0 && !classified) K.todo[e] = 0;
    return;
  }
  if (classified && K.todo[e]) return;
  int na = (actions && n_actions) ? n_actions[e] : 0;
  if (na > CD(max_actions)) na = CD(max_actions);
  const bool ok = cb::decision_step_wave(K, e, actions ? actions + (size_t)e * CD(max_actions) * 3 : nullptr, na, decisions + (size_t)e * 8,
                                         scope + (size_t)e * CD(scope_cap) * 2, (int64_t*)metrics + (size_t)e * 3, done + e, scr);
  if (classified) {
    if (!ok) __builtin_trap();  // (the same question on the same state, asked twice: an env nobody steps must not pass silently)
  } else if (!ok && K.defer) {   // mrx_cb_set_replay_period: the general kernel sits this call out
    cb::defer_env_wave(K, e, actions ? actions + (size_t)e * CD(max_actions) * 3 : nullptr, na, decisions + (size_t)e * 8,
                       scope + (size_t)e * CD(scope_cap) * 2, (int64_t*)metrics + (size_t)e * 3, done + e);
    if (threadIdx.x == 0) K.todo[e] = 0;
  } else if (threadIdx.x == 0) {
    K.todo[e] = ok ? 0 : 1;
  }
}

#if defined(MRX_CB_LDSFRAME) && MRXC_lsh_plan < 0   /* (the wave REPLAY kernel runs with lsh 0: a build with the envs-per-wave shift compiled in has the speculative wave kernel only — forcing wave mode on loads the runtime-shift build, citi_bike/engine.py) */
// The GENERAL step on one wave per env (cb::step_env_wave): for the envs mrx_k_cb_step_wave flagged in K.todo.  The env's state is
// moved HBM <-> the wave's LDS column by all 64 lanes (launched with K.lsh = 0: one column), the sequential parts run on lane 0
// out of LDS, the station sweeps / snapshot / action scope across the lanes.
extern "C" __global__ void __launch_bounds__(64)
mrx_k_cb_replay_wave(CbParams K, const int32_t* __restrict__ actions, const int32_t* __restrict__ n_actions, const uint8_t* __restrict__ todo,
                     int32_t* __restrict__ decisions, int32_t* __restrict__ scope, long long* __restrict__ metrics, uint8_t* __restrict__ done) {
  __shared__ int32_t scr[2 * cb::CBW_MAX + 8];
  const int e = (int)blockIdx.x, lane = (int)threadIdx.x;
  if (!todo[e]) return;
  cb::WProf WP;
  WP.count(K, e, 12);
#define MRX_CB_LW(w) cb::mrx_cb_lds[CB_EV_BLOCK * 4 + (w)]
#ifdef MRX_CB_POOL_LDS
  const int pool_head = K.hdr[CB_IX(CD(aos), CD(stride), CH_WORDS, CH_POOL_HEAD, e)], pool_tail = K.hdr[CB_IX(CD(aos), CD(stride), CH_WORDS, CH_POOL_TAIL, e)];
  const int ev_pos = K.hdr[CB_IX(CD(aos), CD(stride), CH_WORDS, CH_EV_POS, e)];
#endif
  // Every load of the column in flight before the first LDS write: the loops below have compile-time trip counts and no load sits
  // behind a branch (a lane past the end reads word 0 again and drops it).  As `for (w = lane; w < N; w += 64) lds = hbm` each
  // trip waited for its own load: thirteen round trips for the capacities alone.
#define MRX_CB_FILL_LOAD(v, N, expr) \
  _Pragma("unroll") for (int i_ = 0; i_ < ((N) + 63) / 64; i_++) { const int w_ = i_ * 64 + lane < (N) ? i_ * 64 + lane : 0; (v)[i_] = (int32_t)(expr); }
#define MRX_CB_FILL_STORE(v, N, at) \
  _Pragma("unroll") for (int i_ = 0; i_ < ((N) + 63) / 64; i_++) if (i_ * 64 + lane < (N)) MRX_CB_LW((at) + i_ * 64 + lane) = (v)[i_];
#if MRXC_aos && MRXC_FW % 4 == 0
  constexpr int NQ = MRXC_FW / 4, NI = (NQ + 63) / 64;  // env-major: the frame is one contiguous run — 16 bytes per lane and load
  int4 q[NI];
  {
    const int4* src = (const int4*)(K.live + (size_t)e * MRXC_FW);
#pragma unroll
    for (int i = 0; i < NI; i++) q[i] = src[(i + 1) * 64 <= NQ || i * 64 + lane < NQ ? i * 64 + lane : 0];
  }
#else
  int32_t q[(MRXC_FW + 63) / 64];
  MRX_CB_FILL_LOAD(q, MRXC_FW, K.live[CB_IX(CD(aos), CD(stride), CD(FW), w_, e)])
#endif
  int32_t vcap[(MRXC_S + 63) / 64], vhdr[1], vful[(MRXC_w_words + 63) / 64], vdmk[(2 * MRXC_mask_words + 63) / 64];
  MRX_CB_FILL_LOAD(vcap, MRXC_S, K.capacity[w_])
  MRX_CB_FILL_LOAD(vhdr, CH_WORDS, K.hdr[CB_IX(CD(aos), CD(stride), CH_WORDS, w_, e)])
  MRX_CB_FILL_LOAD(vful, MRXC_w_words, K.fulfilled[CB_IX(CD(aos), CD(stride), CD(w_words), w_, e)])
  MRX_CB_FILL_LOAD(vdmk, 2 * MRXC_mask_words, K.decmask[CB_IX(CD(aos), CD(stride), (2 * CD(mask_words)), w_, e)])
#ifdef MRX_CB_TWC_LDS
  int32_t vtf[(MRXC_ring_slots + 63) / 64], vtt[(MRXC_ring_slots + 63) / 64];
  MRX_CB_FILL_LOAD(vtf, MRXC_ring_slots, K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), w_, e)])
  MRX_CB_FILL_LOAD(vtt, MRXC_ring_slots, K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), w_, e)])
#endif
#ifdef MRX_CB_POOL_LDS
  // (the loads that need a header word go out last — everything above is in flight while that word arrives — and land last)
  cb::EvWin::Rec ev_rec0;
  cb::PoolStage PS;
  if (K.pool_stage) { ev_rec0 = cb::evw_fetch(K, ev_pos); cb::pool_stage_fetch(K, e, pool_head, pool_tail, PS); }
#endif
#if MRXC_aos && MRXC_FW % 4 == 0
  {
    int4* dst = (int4*)&MRX_CB_LW(0);
#pragma unroll
    for (int i = 0; i < NI; i++)
      if ((i + 1) * 64 <= NQ || i * 64 + lane < NQ) dst[i * 64 + lane] = q[i];
  }
#else
  MRX_CB_FILL_STORE(q, MRXC_FW, 0)
#endif
  MRX_CB_FILL_STORE(vcap, MRXC_S, LDS_CAP)
  MRX_CB_FILL_STORE(vhdr, CH_WORDS, LDS_HDR)
  MRX_CB_FILL_STORE(vful, MRXC_w_words, LDS_FUL)
  MRX_CB_FILL_STORE(vdmk, 2 * MRXC_mask_words, LDS_DMK)
#ifdef MRX_CB_TWC_LDS
  MRX_CB_FILL_STORE(vtf, MRXC_ring_slots, LDS_TWC)
  MRX_CB_FILL_STORE(vtt, MRXC_ring_slots, LDS_TWC + MRXC_ring_slots)
#endif
#ifdef MRX_CB_POOL_LDS
  if (K.pool_stage) { cb::pool_stage_put(K, PS); cb::evw_put(K, ev_pos, ev_rec0); }
#endif
  __syncthreads();
  WP.mark(K, e, 0);
  int na = (actions && n_actions) ? n_actions[e] : 0;
  if (na > CD(max_actions)) na = CD(max_actions);
  cb::step_env_wave(K, e, actions ? actions + (size_t)e * CD(max_actions) * 3 : nullptr, na, decisions + (size_t)e * 8, scope + (size_t)e * CD(scope_cap) * 2,
                    (int64_t*)metrics + (size_t)e * 3, done + e, scr);
  __syncthreads();
  WP.mark(K, e, 15);  // (step_env_wave keeps its own clock: this stretch is the sum of its phases 1-6)
#if MRXC_aos && MRXC_FW % 4 == 0
  {  // (all the LDS reads first, then the 16-byte stores: a read per trip made the write-back a chain of LDS round trips)
    const int4* srcl = (const int4*)&MRX_CB_LW(0);
    int4* dstg = (int4*)(K.live + (size_t)e * MRXC_FW);
#pragma unroll
    for (int i = 0; i < NI; i++) q[i] = srcl[(i + 1) * 64 <= NQ || i * 64 + lane < NQ ? i * 64 + lane : 0];
#pragma unroll
    for (int i = 0; i < NI; i++)
      if ((i + 1) * 64 <= NQ || i * 64 + lane < NQ) dstg[i * 64 + lane] = q[i];
  }
#else
#pragma unroll 8
  for (int w = lane; w < MRXC_FW; w += 64) K.live[CB_IX(CD(aos), CD(stride), CD(FW), w, e)] = MRX_CB_LW(w);
#endif
  for (int w = lane; w < CH_WORDS; w += 64) K.hdr[CB_IX(CD(aos), CD(stride), CH_WORDS, w, e)] = MRX_CB_LW(LDS_HDR + w);
#ifdef MRX_CB_TWC_LDS
  for (int w = lane; w < MRXC_ring_slots; w += 64) {
    K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, e)] = MRX_CB_LW(LDS_TWC + w);
    K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, e)] = MRX_CB_LW(LDS_TWC + MRXC_ring_slots + w);
  }
#endif
#pragma unroll
  for (int i = 0; i < (MRXC_w_words + 63) / 64; i++)
    if (i * 64 + lane < MRXC_w_words) K.fulfilled[CB_IX(CD(aos), CD(stride), CD(w_words), i * 64 + lane, e)] = (uint32_t)MRX_CB_LW(LDS_FUL + i * 64 + lane);
  for (int w = lane; w < 2 * MRXC_mask_words; w += 64) K.decmask[CB_IX(CD(aos), CD(stride), (2 * CD(mask_words)), w, e)] = (uint32_t)MRX_CB_LW(LDS_DMK + w);
  WP.mark(K, e, 7);
#undef MRX_CB_FILL_LOAD
#undef MRX_CB_FILL_STORE
#undef MRX_CB_LW
}
#endif
