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
#define MRX_STEP_KERNEL(NAME, PG, OBS, WAVES)                                                                                  \
  extern "C" __global__ void __launch_bounds__(64, WAVES)                                                           \
  NAME(CimParams K, CimObs O, const int32_t* __restrict__ actions, const int32_t* __restrict__ n_actions,           \
       const int32_t* __restrict__ n_answered, const uint8_t* __restrict__ mask, int32_t* __restrict__ decisions,   \
       long long* __restrict__ metrics, uint8_t* __restrict__ done) {                                               \
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];                                                   \
    const int env = blockIdx.x;                                                                                     \
    if (mask && !mask[env]) return;                                                                                 \
    const int32_t* a = actions ? actions + (size_t)env * KD(max_actions) * 4 : nullptr;                               \
    const int na = (actions && n_actions) ? n_actions[env] : 0;                                                     \
    const size_t drow = KD(decision_mode) ? (size_t)KD(V) * 8 : 8; /* Joint modes: one row per vessel */                \
    cim::step_env<PG, OBS>(K, O, env, lds, a, na, n_answered ? n_answered[env] : -1, decisions + (size_t)env * drow,     \
                      metrics + (size_t)env * 3, done + env);                                                       \
  }
// a specialised build only needs the one kernel that matches its plan's order mode (CimParams::pregen) and whether a fused
// observation is configured (mrx_cim_set_observation reloads the code object when that changes)
#ifdef MRX_SPECIALIZED
#define MRX_WANT(PG, OBS) ((PG) == (MRXC_pregen != 0) && (OBS) == ((MRXC_obs_np | MRXC_obs_nv) != 0))
#else
#define MRX_WANT(PG, OBS) 1
#endif
#if MRX_WANT(0, 0)
MRX_STEP_KERNEL(mrx_k_cim_step, false, false, MRX_STEP_WAVES)
#endif
#if MRX_WANT(0, 1)
MRX_STEP_KERNEL(mrx_k_cim_step_obs, false, true, MRX_STEP_WAVES)      // + fused observation (mrx_cim_set_observation)
#endif
#if MRX_WANT(1, 0)
MRX_STEP_KERNEL(mrx_k_cim_step_tab, true, false, MRX_STEP_WAVES)
#endif
#if MRX_WANT(1, 1)
MRX_STEP_KERNEL(mrx_k_cim_step_tab_obs, true, true, MRX_STEP_WAVES)
#endif
#undef MRX_WANT
#undef MRX_STEP_KERNEL
