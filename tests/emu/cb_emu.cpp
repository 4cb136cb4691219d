// cb_emu.cpp — TEST INFRASTRUCTURE ONLY.  Compiles the citi_bike engine's device source
// (maro_amd/csrc/cb_device.h, one env per lane, no cross-lane operations) for the host, one call per env, so the
// `-m "not gpu"` suite can compare the real kernel logic with the CPU oracle.  The product never loads this.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "wave_emu.h"   // 64-fiber wave emulation (defines MRX_DEV): only the wave-cooperative decision step (cb_wave.h) uses it
#include "../../maro_amd/csrc/cb_layout.h"
#include "../../maro_amd/csrc/cb_device.h"
#include "../../maro_amd/csrc/cb_wave.h"

struct CbEmu {
  CbHostPlan plan;
  uint8_t* ws = nullptr;
  int pool_stage = CB_POOL_STAGE;   // cb_emu_set_pool_stage: step_env_wave's K.pool_stage (env-major builds) — the product launches with CB_POOL_STAGE
  int period = 1;        // cb_emu_set_replay_period: mrx_cb_set_replay_period (the general step on every n-th call; deferred envs in between)
  long calls = 0;
  int overlap = 1;       // cb_emu_set_replay_overlap: the split mrx_cb_step makes (classify, then the two wave kernels on disjoint envs)
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
  CbParams Kr = K;          // what mrx_cb_step passes to mrx_k_cb_replay_wave: the delivery pool's hot end staged in LDS
  Kr.pool_stage = e->pool_stage;
  // mrx_cb_set_replay_period (wave-stepped plans with the wave-form general step): on the calls in between, envs that leave their tick
  // are deferred by cb::defer_env_wave instead of replayed
  const bool defer = e->wave_mode == 2 && e->period > 1 && (++e->calls % e->period) != 0;
  auto defer_env = [&](int env, const int32_t* act, int nac) {
    wave::run_wave(e->wave, [&]() {
      cb::defer_env_wave(K, env, act, nac, dec + (size_t)env * 8, scope + (size_t)env * K.scope_cap * 2, met + (size_t)env * 3, done + env);
    });
  };
#ifdef MRX_CB_LDSFRAME
  if (e->wave_mode == 2 && e->overlap && K.decision_mode == 0 && K.start_tick % K.res == 0 && K.mask_words <= 64) {
    // mrx_cb_step with the replay overlap on: mrx_k_cb_classify over every env FIRST (reads only), then the replay kernel's envs and
    // the in-tick kernel's envs — disjoint sets, here one after the other in the order that would expose a dependence between them
    static int32_t scr2[2 * cb::CBW_MAX + 8];
    std::vector<uint8_t> todo((size_t)K.n_envs, 2);
    e->wave.reverse = e->reverse;
    auto args = [&](int env, const int32_t*& act, int& nac) {
      const int na = n_actions ? n_actions[env] : 0;
      nac = na < K.max_actions ? na : K.max_actions;
      act = actions ? actions + (size_t)env * K.max_actions * 3 : nullptr;
    };
    for (int env = 0; env < K.n_envs; env++) {
      if (mask && !mask[env]) continue;
      const int32_t* act; int nac;
      args(env, act, nac);
      bool ok = false;
      wave::run_wave(e->wave, [&]() {
        cb::WavePre P;
        const bool r = cb::decision_step_wave_pre(K, env, (actions && n_actions) ? nac : 0, P);
        if (wave::lane() == 0) ok = r;
      });
      todo[env] = ok ? 0 : 1;
    }
    for (int pass = 0; pass < 2; pass++)
      for (int env = 0; env < K.n_envs; env++) {
        const int32_t* act; int nac;
        args(env, act, nac);
        if (pass == 0 && todo[env] == 1 && defer) {
          defer_env(env, act, (actions && n_actions) ? nac : 0);
        } else if (pass == 0 && todo[env] == 1) {
          e->general++;
          wave::run_wave(e->wave, [&]() {
            cb::step_env_wave(Kr, env, act, nac, dec + (size_t)env * 8, scope + (size_t)env * K.scope_cap * 2, met + (size_t)env * 3, done + env, scr2);
          });
        } else if (pass == 1 && todo[env] == 0) {
          bool ok = false;
          wave::run_wave(e->wave, [&]() {
            const bool r = cb::decision_step_wave(K, env, act, (actions && n_actions) ? nac : 0, dec + (size_t)env * 8, scope + (size_t)env * K.scope_cap * 2,
                                                  met + (size_t)env * 3, done + env, scr);
            if (wave::lane() == 0) ok = r;
          });
          if (!ok) { fprintf(stderr, "cb_emu: env %d classified in-tick but decision_step_wave left it alone\n", env); abort(); }
          e->handled++;
        }
      }
    return;
  }
#endif
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
      if (e->wave_mode == 2 && defer) { defer_env(env, act, (actions && n_actions) ? nac : 0); e->general--; continue; }
      if (e->wave_mode == 2) {  // the general step in its wave form too (plan-specialised LDS-frame builds: mrx_k_cb_replay_wave)
        static int32_t scr2[2 * cb::CBW_MAX + 8];
        wave::run_wave(e->wave, [&]() {
          cb::step_env_wave(Kr, env, act, nac, dec + (size_t)env * 8, scope + (size_t)env * K.scope_cap * 2, met + (size_t)env * 3, done + env, scr2);
        });
        continue;
      }
#endif
    }
    cb::step_env(K, env, act, nac, nullptr, dec + (size_t)env * 8, scope + (size_t)env * K.scope_cap * 2, met + (size_t)env * 3, done + env);
    if (K.obs && !e->wave_mode) cb::write_observation(K, env, dec + (size_t)env * 8);   // (what mrx_k_cb_step does after step_env)
  }
}

// cb::cbw_rank on the wave emulator: candidate i = (v[i], key[i]), i < n <= 128; rank_out[i] = how many come before it
void cb_emu_rank(void* h, int n, int mode, const int32_t* v, const int32_t* key, int32_t* rank_out) {
  CbEmu* e = (CbEmu*)h;
  const CbParams& K = e->plan.kp;
  e->wave.reverse = e->reverse;
  wave::run_wave(e->wave, [&]() {
    const int lane = wave::lane();
    const int vv[2] = {v[lane], v[64 + lane]}, kk[2] = {key[lane], key[64 + lane]};
    int rank[2];
    cb::cbw_rank(K, n, mode, vv, kk, rank);
    rank_out[lane] = rank[0];
    rank_out[64 + lane] = rank[1];
  });
}

void cb_emu_set_pool_stage(void* h, int entries) { ((CbEmu*)h)->pool_stage = entries < 0 ? 0 : entries > CB_POOL_STAGE ? CB_POOL_STAGE : entries; }
void cb_emu_set_replay_period(void* h, int n) { ((CbEmu*)h)->period = n < 1 ? 1 : n; }
void cb_emu_set_replay_overlap(void* h, int on) { ((CbEmu*)h)->overlap = on; }
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
