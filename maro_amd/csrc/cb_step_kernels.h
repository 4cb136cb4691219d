// cb_step_kernels.h — the citi_bike reset and step kernels, shared by the generic build (cb_engine.hip) and the plan-specialised
// build (cb_spec.hip, MRX_SPECIALIZED: CD() dimensions are constants).  No include guard on purpose: it only instantiates kernels.
extern "C" __global__ void __launch_bounds__(64)
mrx_k_cb_reset(CbParams K, const int32_t* __restrict__ tt, int n_times, const uint8_t* __restrict__ mask) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= K.n_envs || (mask && !mask[e])) return;
  if (tt)
    for (int i = 0; i < CD(tt_cap); i++) K.tt[(size_t)i * CD(stride) + e] = i < n_times ? tt[(size_t)e * n_times + i] : 1;
  cb::reset_env(K, e);
}

extern "C" __global__ void __launch_bounds__(64)
mrx_k_cb_step(CbParams K, const int32_t* __restrict__ actions, const int32_t* __restrict__ n_actions, const uint8_t* __restrict__ mask,
              int32_t* __restrict__ decisions, int32_t* __restrict__ scope, long long* __restrict__ metrics, uint8_t* __restrict__ done, int lanes) {
  // `lanes` envs per wave (mrx_cb_set_lanes_per_wave): the other lanes of the wave retire at once.  A wave runs the UNION of
  // its lanes' control flow, and steps differ a lot in length, so a small batch is faster spread thin over many waves
  if ((int)threadIdx.x >= lanes) return;
  const int e = blockIdx.x * lanes + threadIdx.x;
  if (e >= K.n_envs || (mask && !mask[e])) return;
  int na = (actions && n_actions) ? n_actions[e] : 0;
  if (na > CD(max_actions)) na = CD(max_actions);
  cb::step_env(K, e, actions ? actions + (size_t)e * CD(max_actions) * 3 : nullptr, na, decisions + (size_t)e * 8,
               scope + (size_t)e * CD(scope_cap) * 2, (int64_t*)metrics + (size_t)e * 3, done + e);
}

