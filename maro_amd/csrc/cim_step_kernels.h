// cim_step_kernels.h — the four step-kernel entry points, shared by the generic build (cim_engine.hip, plan dimensions read
// from the kernel arguments) and the plan-specialised build (cim_spec.hip, MRX_SPECIALIZED: dimensions are constants).
// No include guard on purpose: it only instantiates kernels (reset, order table, and the four step kernels).
extern "C" __global__ void __launch_bounds__(64)
mrx_k_cim_reset(CimParams K, const long long* __restrict__ seed_cmd, const uint8_t* __restrict__ mask, long long default_cmd) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const int env = blockIdx.x;
  if (mask && !mask[env]) return;
  cim::reset_env(K, env, lds, seed_cmd ? seed_cmd[env] : default_cmd);
}

extern "C" __global__ void __launch_bounds__(64)
mrx_k_cim_order_table(CimParams K, const long long* __restrict__ seed_cmd, const uint8_t* __restrict__ mask, long long default_cmd) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const int env = blockIdx.x;
  if (mask && !mask[env]) return;
  if ((seed_cmd ? seed_cmd[env] : default_cmd) == -1) return;  // reset(keep_seed=True): same seed, same table
  cim::gen_order_table(K, env, lds);
}

#ifndef MRX_STEP_WAVES
#define MRX_STEP_WAVES 2  // generic build: ~197 VGPRs, 2 waves/SIMD
#endif
// One env per WAVE, MRX_WG_WAVES waves per workgroup (1 in the generic build; the plan's wg_waves in a specialised one: the
// waves share the staged topology tables and nothing else).  sorted = 0: wave s steps env s (the fast rows and the env's hint
// are read first); sorted = 1: wave s steps entry s of the order list of this step (mrx_k_cim_schedule: full-path envs first,
// so the long waves start first and the short ones fill the tail; a full-path entry skips the header round trip); sorted = 2
// (launch form 5): only the full-path entries [0, sched[0]) — the fast-hinted envs were stepped by mrx_k_cim_fast_lanes.
#ifdef MRX_SPECIALIZED
#define MRX_WG_WAVES MRXC_wg_waves
#else
#define MRX_WG_WAVES 1
#endif
#define MRX_STEP_KERNEL(NAME, PG, OBS, WAVES)                                                                            \
  extern "C" __global__ void __launch_bounds__(64 * MRX_WG_WAVES, WAVES)                                                 \
  NAME(CimParams K, CimObs O, cim::StepBatch B, const uint8_t* __restrict__ mask, int sorted) {                          \
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];                                                       \
    const int w = MRX_WG_WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;                        \
    const int slot = (int)blockIdx.x * MRX_WG_WAVES + w;                                                                \
    if (MRX_WG_WAVES > 1 && slot >= K.n_envs) return;                                                                   \
    int env = slot, path = cim::PATH_PROBE;                                                                             \
    if (sorted) {                                                                                                       \
      if (sorted == 2 && slot >= K.sched[0]) return; /* launch form 5: only the full-path list (the fast kernel ran) */  \
      const int e = K.order[slot];                                                                                      \
      if (e < 0) return;                                                                                                \
      env = e & (MRX_ORDER_TICK - 1);                                                                                   \
      path = (e & MRX_ORDER_TICK) ? cim::PATH_FULL : cim::PATH_FAST;                                                    \
    } else if (mask && !mask[env]) {                                                                                    \
      return;                                                                                                           \
    }                                                                                                                   \
    cim::step_env<PG, OBS>(K, O, env, lds + w * KD(l_ctab), cim::step_io(K, B, env), path,                               \
                           lds + MRX_WG_WAVES * KD(l_ctab));                                                            \
  }
// a specialised build only needs the one kernel that matches its plan's order mode (CimParams::pregen) and whether a fused
// observation is configured (mrx_cim_set_observation reloads the code object when that changes)
#ifdef MRX_SPECIALIZED
#define MRX_WANT(PG, OBS) ((PG) == (MRXC_pregen != 0) && (OBS) == ((MRXC_obs_np | MRXC_obs_nv) != 0))
#else
#define MRX_WANT(PG, OBS) 1
#endif
// launch form 4: the lane-parallel fast-path kernel (no LDS) and the looped full-path kernel
#define MRX_SPLIT_KERNELS(SUFFIX, PG, OBS, WAVES)                                                                       \
  extern "C" __global__ void __launch_bounds__(64)                                                                      \
  mrx_k_cim_fast_lanes##SUFFIX(CimParams K, CimObs O, cim::StepBatch B, const uint8_t* __restrict__ mask) {               \
    cim::fast_lanes_env<OBS>(K, O, B, mask, (int)(blockIdx.x * 64 + threadIdx.x));                                      \
  }                                                                                                                     \
  extern "C" __global__ void __launch_bounds__(64, WAVES)                                                               \
  mrx_k_cim_step_loop##SUFFIX(CimParams K, CimObs O, cim::StepBatch B) {                                                 \
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];                                                       \
    cim::step_loop<PG, OBS>(K, O, lds, (int)blockIdx.x, (int)gridDim.x, B);                                             \
  }
#if MRX_WANT(0, 0)
MRX_STEP_KERNEL(mrx_k_cim_step, false, false, MRX_STEP_WAVES)
MRX_SPLIT_KERNELS(, false, false, MRX_STEP_WAVES)
#endif
#if MRX_WANT(0, 1)
MRX_STEP_KERNEL(mrx_k_cim_step_obs, false, true, MRX_STEP_WAVES)      // + fused observation (mrx_cim_set_observation)
MRX_SPLIT_KERNELS(_obs, false, true, MRX_STEP_WAVES)
#endif
#if MRX_WANT(1, 0)
MRX_STEP_KERNEL(mrx_k_cim_step_tab, true, false, MRX_STEP_WAVES)
MRX_SPLIT_KERNELS(_tab, true, false, MRX_STEP_WAVES)
#endif
#if MRX_WANT(1, 1)
MRX_STEP_KERNEL(mrx_k_cim_step_tab_obs, true, true, MRX_STEP_WAVES)
MRX_SPLIT_KERNELS(_tab_obs, true, true, MRX_STEP_WAVES)
#endif
#undef MRX_WANT
#undef MRX_STEP_KERNEL
#undef MRX_SPLIT_KERNELS
