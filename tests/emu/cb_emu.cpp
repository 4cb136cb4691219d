// cb_emu.cpp — TEST INFRASTRUCTURE ONLY.  Compiles the citi_bike engine's device source
// (maro_amd/csrc/cb_device.h, one env per lane, no cross-lane operations) for the host, one call per env, so the
// `-m "not gpu"` suite can compare the real kernel logic with the CPU oracle.  The product never loads this.
#include <stdio.h>
#include <stdlib.h>

#include "wave_emu.h"   // 64-fiber wave emulation (defines MRX_DEV): only the wave-cooperative decision step (cb_wave.h) uses it
#include "../../maro_amd/csrc/cb_layout.h"
#include "../../maro_amd/csrc/cb_device.h"
#include "../../maro_amd/csrc/cb_wave.h"

struct CbEmu {
  CbHostPlan plan;
  uint8_t* ws = nullptr;
  int wave_mode = 0;     // cb_emu_set_wave_decisions: 1 = steps go through cb::decision_step_wave first, like mrx_cb_step does
  bool reverse = false;  // lane order of the wave emulator (forward / reverse exposes missing syncs)
  long handled = 0, general = 0;
  wave::EmuWave wave{};
};

extern "C" {

void* cb_emu_create(const mrx_cb_topology* t, const mrx_cb_config* c, char* errbuf, int errlen) {
  CbEmu* e = new CbEmu();
  std::string err;
  if (cb_plan(t, c, &e->plan, &err) != MRX_OK) {
    snprintf(errbuf, errlen, "%s", err.c_str());
    delete e;
    return nullptr;
  }
  e->ws = (uint8_t*)aligned_alloc(256, (size_t)e->plan.workspace_bytes);
  memset(e->ws, 0xCD, (size_t)e->plan.workspace_bytes);  // poison: nothing may rely on zeroed HBM
  memcpy(e->ws + e->plan.const_off, e->plan.const_blob.data(), e->plan.const_blob.size());
  cb_plan_bind(&e->plan, e->ws);
  return e;
}

void cb_emu_destroy(void* h) {
  CbEmu* e = (CbEmu*)h;
  free(e->ws);
  delete e;
}

void cb_emu_get_layout(void* h, mrx_cb_layout* out) { *out = ((CbEmu*)h)->plan.layout; }
void* cb_emu_workspace(void* h) { return ((CbEmu*)h)->ws; }

void cb_emu_reset(void* h, const int32_t* tt, int n_times, const uint8_t* mask) {
  CbEmu* e = (CbEmu*)h;
  const CbParams& K = e->plan.kp;
  for (int env = 0; env < K.n_envs; env++) {
    if (mask && !mask[env]) continue;
    if (tt) for (int i = 0; i < K.tt_cap; i++) K.tt[CB_IX(CD(aos), CD(stride), CD(tt_cap), i, env)] = i < n_times ? tt[(size_t)env * n_times + i] : 1;
    cb::reset_env(K, env);
  }
}

void cb_emu_set_step_budget(void* h, int max_records) { ((CbEmu*)h)->plan.kp.step_budget = max_records; }

// mrx_cb_set_observation for the harness: rows = every station (lane path) or the stations of the action scope (wave path)
void cb_emu_set_observation(void* h, const int32_t* attrs, int n, double* obs) {
  CbParams& K = ((CbEmu*)h)->plan.kp;
  for (int i = 0; i < n && i < 8; i++) K.obs_attr[i] = attrs[i];
  K.obs_n = n;
  K.obs = n > 0 ? obs : nullptr;
}

void cb_emu_step(void* h, const int32_t* actions, const int32_t* n_actions, const uint8_t* mask, int32_t* dec, int32_t* scope,
                 int64_t* met, uint8_t* done) {
  CbEmu* e = (CbEmu*)h;
  const CbParams& K = e->plan.kp;
  static int32_t scr[2 * cb::CBW_MAX];
  for (int env = 0; env < K.n_envs; env++) {
    if (mask && !mask[env]) continue;
    const int na = n_actions ? n_actions[env] : 0;
    const int nac = na < K.max_actions ? na : K.max_actions;
    const int32_t* act = actions ? actions + (size_t)env * K.max_actions * 3 : nullptr;
    if (e->wave_mode && K.decision_mode == 0 && K.start_tick % K.res == 0 && K.mask_words <= 64) {
      bool ok = false;
      e->wave.reverse = e->reverse;
      wave::run_wave(e->wave, [&]() {
        const bool r = cb::decision_step_wave(K, env, act, (actions && n_actions) ? nac : 0, dec + (size_t)env * 8, scope + (size_t)env * K.scope_cap * 2,
                                              met + (size_t)env * 3, done + env, scr);
        if (wave::lane() == 0) ok = r;
      });
      if (ok) { e->handled++; continue; }
      e->general++;
#ifdef MRX_CB_LDSFRAME
      if (e->wave_mode == 2) {  // the general step in its wave form too (plan-specialised LDS-frame builds: mrx_k_cb_replay_wave)
        static int32_t scr2[2 * cb::CBW_MAX + 8];
        wave::run_wave(e->wave, [&]() {
          cb::step_env_wave(K, env, act, nac, dec + (size_t)env * 8, scope + (size_t)env * K.scope_cap * 2, met + (size_t)env * 3, done + env, scr2);
        });
        continue;
      }
#endif
    }
    cb::step_env(K, env, act, nac, nullptr, dec + (size_t)env * 8, scope + (size_t)env * K.scope_cap * 2, met + (size_t)env * 3, done + env);
    if (K.obs && !e->wave_mode) cb::write_observation(K, env, dec + (size_t)env * 8);   // (what mrx_k_cb_step does after step_env)
  }
}

void cb_emu_set_wave_decisions(void* h, int on, int reverse) { ((CbEmu*)h)->wave_mode = on; ((CbEmu*)h)->reverse = reverse != 0; }
long cb_emu_wave_handled(void* h) { return ((CbEmu*)h)->handled; }
long cb_emu_wave_general(void* h) { return ((CbEmu*)h)->general; }

// Joint modes (mrx_cb_step_joint): S rows per env
void cb_emu_step_joint(void* h, const int32_t* actions, const int32_t* n_actions, const int32_t* n_answered, const uint8_t* mask, int32_t* dec,
                       int32_t* scope, int64_t* met, uint8_t* done) {
  CbEmu* e = (CbEmu*)h;
  const CbParams& K = e->plan.kp;
  const size_t S = (size_t)K.S;
  for (int env = 0; env < K.n_envs; env++) {
    if (mask && !mask[env]) continue;
    int nans = (actions && n_answered) ? n_answered[env] : 0;
    if (nans < 0) nans = 0;
    cb::step_env(K, env, actions ? actions + (size_t)env * S * K.max_actions * 3 : nullptr, nans, n_actions ? n_actions + (size_t)env * S : nullptr,
                 dec + (size_t)env * S * 8, scope + (size_t)env * S * K.scope_cap * 2, met + (size_t)env * 3, done + env);
  }
}

void cb_emu_query(void* h, int node_type, const int32_t* ticks, int nt, int ticks_per_env, const int32_t* nodes, int nn,
                  int nodes_per_env, const int32_t* attrs, int na, double* out) {
  CbEmu* e = (CbEmu*)h;
  const CbParams& K = e->plan.kp;
  int row_slots = 0;
  for (int i = 0; i < na; i++) row_slots += cb::attr_slots(K, node_type, attrs[i]);
  const long long rows = (long long)K.n_envs * nt * nn;
  for (long long r = 0; r < rows; r++)
    for (int c = 0; c < row_slots; c++)
      out[r * row_slots + c] = cb::query_elem(K, node_type, ticks, nt, ticks_per_env, nodes, nn, nodes_per_env, attrs, na, r, c);
}

void cb_emu_random_policy(void* h, const int32_t* dec, const int32_t* scope, int64_t step, int32_t* actions, int32_t* n_actions) {
  CbEmu* e = (CbEmu*)h;
  const CbParams& K = e->plan.kp;
  for (int env = 0; env < K.n_envs; env++)
    cb::random_policy_env(K, env, dec + (size_t)env * 8, scope + (size_t)env * K.scope_cap * 2, step,
                          actions + (size_t)env * K.max_actions * 3, n_actions + env);
}
}

// The plan's integer dimensions as "NAME value" lines (tools: kernel specialisation experiments).
extern "C" int cb_emu_dump_dims(void* h, char* buf, int len) {
  const CbParams& K = ((CbEmu*)h)->plan.kp;
  std::string o;
#define D(f) o += std::string(#f) + " " + std::to_string((long long)K.f) + "\n";
  D(stride) D(S) D(start_tick) D(max_tick) D(res) D(ring_slots) D(max_actions) D(dres) D(extra_cost_mode) D(n_filters) D(FW)
  D(w_mask) D(w_words) D(pool_cap) D(tt_cap) D(scope_cap) D(mask_words) D(nb_stride)
#undef D
  for (int i = 0; i < 4; i++) o += "f_type[" + std::to_string(i) + "] " + std::to_string(K.f_type[i]) + "\nf_num[" + std::to_string(i) + "] " + std::to_string(K.f_num[i]) + "\nf_win[" + std::to_string(i) + "] " + std::to_string(K.f_win[i]) + "\n";
  if ((int)o.size() + 1 > len) return -1;
  memcpy(buf, o.c_str(), o.size() + 1);
  return (int)o.size();
}
