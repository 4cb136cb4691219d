// cim_emu.cpp — TEST INFRASTRUCTURE ONLY.  Compiles the engine's device source
// (maro_amd/csrc/cim_device.h) for the host on top of the fiber wave emulator, so the
// `-m "not gpu"` suite can compare the real kernel logic with the CPU oracle.  The product
// (maro_amd/, libmaro_amd.so) never loads this library.
#include "wave_emu.h"

struct int4 { int x, y, z, w; };

#include "../../maro_amd/csrc/cim_device.h"
#include "../../maro_amd/csrc/cim_layout.h"

struct Emu {
  CimObs obs = {};
  long long agent_key = -1;
  CimHostPlan plan;
  uint8_t* ws = nullptr;
  int32_t* lds = nullptr;
  int lds_cap_words = 0;
  wave::EmuWave wave;
};

extern "C" {

// The plan's integer dimensions / layout offsets as "NAME value" lines (tools: kernel specialisation experiments).
int emu_dump_dims(void* h, char* buf, int len);

void* emu_create(const mrx_cim_topology* t, const mrx_cim_config* c, char* errbuf, int errlen) {
  Emu* e = new Emu();
  std::string err;
  if (cim_plan(t, c, &e->plan, &err) != MRX_OK) {
    snprintf(errbuf, errlen, "%s", err.c_str());
    delete e;
    return nullptr;
  }
  e->ws = (uint8_t*)aligned_alloc(256, (size_t)e->plan.workspace_bytes);
  memset(e->ws, 0xCD, (size_t)e->plan.workspace_bytes);  // poison: nothing may rely on zeroed HBM
  memcpy(e->ws + e->plan.const_off, e->plan.const_blob.data(), e->plan.const_blob.size());
  cim_plan_bind(&e->plan, e->ws);
  {
    const CimParams& kp = e->plan.kp;
    int w = kp.lds_words_reset > kp.lds_words ? kp.lds_words_reset : kp.lds_words;
    if (kp.lds_words_gen > w) w = kp.lds_words_gen;
    e->lds_cap_words = w;
    e->lds = (int32_t*)aligned_alloc(256, ((size_t)w * 4 + 511) / 256 * 256);
  }
  return e;
}

void emu_destroy(void* h) {
  Emu* e = (Emu*)h;
  wave::free_wave(e->wave);
  free(e->ws);
  free(e->lds);
  delete e;
}

void emu_get_layout(void* h, mrx_cim_layout* out) { *out = ((Emu*)h)->plan.layout; }
void* emu_workspace(void* h) { return ((Emu*)h)->ws; }
int emu_lds_words(void* h) { return ((Emu*)h)->plan.kp.lds_words; }

void emu_reset(void* h, const int64_t* seed_cmd, const uint8_t* mask, int reverse) {
  Emu* e = (Emu*)h;
  const CimParams& K = e->plan.kp;
  e->wave.reverse = reverse != 0;
  for (int env = 0; env < K.n_envs; env++) {
    if (mask && !mask[env]) continue;
    memset(e->lds, 0xAB, (size_t)K.lds_words_reset * 4);
    long long cmd = seed_cmd ? (long long)seed_cmd[env] : -1;
    wave::run_wave(e->wave, [&]() { cim::reset_env(K, env, e->lds, cmd); });
    if (K.pregen && K.orders_stride && cmd != -1) {
      memset(e->lds, 0xAB, (size_t)e->lds_cap_words * 4);
      wave::run_wave(e->wave, [&]() { cim::gen_order_table(K, env, e->lds); });
    }
  }
}

// the host twin of mrx_k_cim_schedule (cim_engine.hip): full-path envs first (the long ones, hint 2, at the head), then the
// fast-hinted ones, -1 padded
static void emu_schedule(Emu* e, const uint8_t* mask) {
  const CimParams& K = e->plan.kp;
  int nt = 0, na = 0;
  for (int env = 0; env < K.n_envs; env++) if ((!mask || mask[env]) && (K.hint[env] & 2)) K.order[nt++] = env | MRX_ORDER_TICK;
  for (int env = 0; env < K.n_envs; env++) if ((!mask || mask[env]) && K.hint[env] && !(K.hint[env] & 2)) K.order[nt++] = env | MRX_ORDER_TICK;
  na = nt;
  for (int env = 0; env < K.n_envs; env++) if ((!mask || mask[env]) && !K.hint[env]) K.order[na++] = env;
  for (int i = na; i < K.n_envs; i++) K.order[i] = -1;
  for (int i = 0; i < 16; i++) K.sched[i] = i == 0 ? nt : i == 1 ? na : i >= 8 ? -1 : 0;
}

// mode 1: unsorted (workgroup b = env b, hint probed); 2: sorted one-env-per-workgroup launch; 4: split step, its looped
// full-path kernel with `pipe_waves` waves; 5: the fast kernel of 4 + one workgroup per full-path entry
void emu_step(void* h, const int32_t* actions, const int32_t* n_actions, const uint8_t* mask, int32_t* dec,
              int64_t* met, uint8_t* done, int reverse, const int32_t* n_answered, int mode, int pipe_waves) {
  Emu* e = (Emu*)h;
  const CimParams& K = e->plan.kp;
  e->wave.reverse = reverse != 0;
  cim::StepBatch B = {actions, n_actions, n_answered, dec, (long long*)met, done};
  const bool obs = e->obs.np > 0 || e->obs.nv > 0;
  if (e->obs.agent_mode) {   // as launch_step (cim_engine.hip): the device agent's key of this step
    e->obs.agent_key = e->agent_key;
    if (e->agent_key >= 0) e->agent_key++;
  }
  if (mode >= 2) emu_schedule(e, mask);
  if (mode == 4) {  // split step: lane-parallel fast kernel (64 envs per wave, no LDS), then the looped full-path kernel
    for (int b0 = 0; b0 < K.n_envs; b0 += 64)
      wave::run_wave(e->wave, [&]() {
        if (obs) cim::fast_lanes_env<true>(K, e->obs, B, mask, b0 + wave::lane());
        else cim::fast_lanes_env<false>(K, e->obs, B, mask, b0 + wave::lane());
      });
    for (int w = 0; w < pipe_waves; w++) {
      memset(e->lds, 0xAB, (size_t)K.lds_words * 4);
      wave::run_wave(e->wave, [&]() {
        if (K.pregen) { if (obs) cim::step_loop<true, true>(K, e->obs, e->lds, w, pipe_waves, B); else cim::step_loop<true, false>(K, e->obs, e->lds, w, pipe_waves, B); }
        else { if (obs) cim::step_loop<false, true>(K, e->obs, e->lds, w, pipe_waves, B); else cim::step_loop<false, false>(K, e->obs, e->lds, w, pipe_waves, B); }
      });
    }
    return;
  }
  if (mode == 5) {  // the fast kernel of form 4, then one workgroup per entry of the full-path list
    for (int b0 = 0; b0 < K.n_envs; b0 += 64)
      wave::run_wave(e->wave, [&]() {
        if (obs) cim::fast_lanes_env<true>(K, e->obs, B, mask, b0 + wave::lane());
        else cim::fast_lanes_env<false>(K, e->obs, B, mask, b0 + wave::lane());
      });
  }
  for (int b = 0; b < K.n_envs; b++) {
    int env = b, path = cim::PATH_PROBE;
    if (mode == 5 && b >= K.sched[0]) break;
    if (mode == 2 || mode == 5) {
      const int en = K.order[b];
      if (en < 0) continue;
      env = en & (MRX_ORDER_TICK - 1);
      path = (en & MRX_ORDER_TICK) ? cim::PATH_FULL : cim::PATH_FAST;
    } else if (mask && !mask[env]) {
      continue;
    }
    memset(e->lds, 0xAB, (size_t)K.lds_words * 4);
    wave::run_wave(e->wave, [&]() {
#define EMU_STEP(PG, OBS) cim::step_env<PG, OBS>(K, e->obs, env, e->lds, cim::step_io(K, B, env), path)
      if (K.pregen) { if (obs) EMU_STEP(true, true); else EMU_STEP(true, false); }
      else { if (obs) EMU_STEP(false, true); else EMU_STEP(false, false); }
#undef EMU_STEP
    });
  }
}

void emu_set_observation(void* h, const int32_t* pa, int np, const int32_t* va, int nv, double* ports, double* vessel) {
  Emu* e = (Emu*)h;
  const CimObs keep = e->obs;
  memset(&e->obs, 0, sizeof(e->obs));
  e->obs.agent_mode = keep.agent_mode; e->obs.agent_actions = keep.agent_actions; e->obs.agent_n_actions = keep.agent_n_actions; e->obs.agent_count = keep.agent_count;
  e->obs.hist_n = keep.hist_n; e->obs.hist_frames = keep.hist_frames; e->obs.hist = keep.hist;
  for (int i = 0; i < 4; i++) e->obs.hist_attr[i] = keep.hist_attr[i];
  for (int i = 0; i < np; i++) e->obs.pa[i] = pa[i];
  for (int i = 0; i < nv; i++) e->obs.va[i] = va[i];
  e->obs.np = np; e->obs.nv = nv; e->obs.ports = ports; e->obs.vessel = vessel;
  memset(e->plan.kp.hint, 1, (size_t)e->plan.kp.n_envs);  // as mrx_cim_set_observation: the next step of every env writes the whole block
  e->obs.i_empty = e->obs.i_tc = -1;
  for (int i = 0; i < np; i++) {
    e->obs.pa_packed |= (unsigned)pa[i] << (4 * i);
    if (pa[i] == PA_EMPTY) e->obs.i_empty = i;
    if (pa[i] == PA_TRANSFER_COST) e->obs.i_tc = i;
  }
}

void emu_set_device_agent(void* h, int mode, int32_t* actions, int32_t* n_actions, int32_t* counts, long long next_key) {
  Emu* e = (Emu*)h;
  e->obs.agent_mode = mode; e->obs.agent_actions = mode ? actions : nullptr; e->obs.agent_n_actions = mode ? n_actions : nullptr;
  e->obs.agent_count = mode ? counts : nullptr;
  e->agent_key = next_key;
}

void emu_set_port_history(void* h, const int32_t* attrs, int n, int32_t* hist, int frames) {
  Emu* e = (Emu*)h;
  e->obs.hist_n = n; e->obs.hist = n ? hist : nullptr; e->obs.hist_frames = n ? frames : 0;
  for (int i = 0; i < n; i++) e->obs.hist_attr[i] = attrs[i];
}

void emu_query(void* h, int node_type, const int32_t* ticks, int nt, int per_env, const int32_t* nodes, int nn, int nodes_per_env,
               const int32_t* attrs, int na, double* out) {
  Emu* e = (Emu*)h;
  const CimParams& K = e->plan.kp;
  int row_slots = 0;
  for (int i = 0; i < na; i++) row_slots += cim::attr_slots(K, node_type, attrs[i]);
  long long rows = (long long)K.n_envs * nt * nn;
  for (long long r = 0; r < rows; r++)
    for (int c = 0; c < row_slots; c++)
      out[r * row_slots + c] = cim::query_elem(K, node_type, ticks, nt, per_env, nodes, nn, nodes_per_env, attrs, na, r, c);
}

long emu_rounds(void* h) { return ((Emu*)h)->wave.rounds; }
}

extern "C" int emu_dump_dims(void* h, char* buf, int len) {
  const CimParams& K = ((Emu*)h)->plan.kp;
  std::string o;
#define D(f) o += std::string(#f) + " " + std::to_string((long long)K.f) + "\n";
  D(P) D(V) D(R) D(NT) D(NRP) D(past_n) D(future_n) D(vrows) D(FW) D(S) D(H) D(SMAX) D(T) D(start_tick) D(resolution)
  D(max_actions) D(period) D(vol) D(total_containers) D(order_mode) D(use_order_rng) D(use_buffer_rng) D(has_order_init)
  D(idx_order_init) D(idx_route) D(idx_order_num) D(idx_buffer) D(f_ports) D(f_vessels) D(f_fop) D(f_fov) D(f_plans)
  D(misc_cap) D(NC) D(PW) D(PWH) D(pv_evt) D(pv_next) D(pv_pos) D(pv_krl) D(pv_period) D(pv_rfull) D(pv_rempty) D(REC_W)
  D(l_frame) D(l_priv) D(l_mt0) D(l_mt1) D(l_dsrc) D(l_dtgt) D(l_oq) D(l_odelay) D(l_srcn) D(l_misc) D(lds_words) D(l_ctab) D(wg_waves) D(l_rfull) D(lds_words_lean) D(lean_ok) D(l_mt2) D(l_mt3) D(r_mt0) D(r_mt1) D(lds_words_reset)
  D(ctab_words) D(decision_mode) D(data_mode) D(data_T) D(pregen) D(NTP) D(order_half) D(order_fast) D(lds_words_gen)
#undef D
  if ((int)o.size() + 1 > len) return -1;
  memcpy(buf, o.c_str(), o.size() + 1);
  return (int)o.size();
}
