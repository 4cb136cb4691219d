// cim_layout.h — host-side planning of the engine's HBM workspace and constant tables.
// Plain C++ (no HIP): used by the C-ABI implementation (cim_engine.hip) and by the CPU
// emulation harness in tests/emu/.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/maro_amd.h"
#include "cim_params.h"

struct CimHostPlan {
  CimParams kp;                 // device pointers valid after cim_plan_bind()
  mrx_cim_layout layout;
  std::vector<uint8_t> const_blob;  // constant tables, uploaded at workspace + const_off
  int64_t const_off = 0;
  int64_t workspace_bytes = 0;
  // byte offsets of per-env arrays inside the workspace
  int64_t o_live, o_ring, o_ring_fi, o_priv, o_rec, o_status, o_tick, o_nstops, o_order_prop, o_mt, o_stops, o_seed, o_vperiod, o_orders, o_hint, o_order, o_sched;
  // relative offsets of const tables inside const_blob, in the order of CimParams' const pointers
  std::vector<std::pair<size_t, int64_t>> binds;  // (byte offset of a pointer field inside kp, offset in const_blob)
  int64_t ctab_rel = 0;
  int64_t shared_orders_rel = -1;  // real data files: the order table inside const_blob, shared by every env
};

namespace cim_layout_detail {
inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
struct Arena {
  int64_t top = 0;
  int64_t take(int64_t bytes) { int64_t o = align_up(top, 256); top = o + bytes; return o; }
};
template <class T>
inline int64_t blob_put(std::vector<uint8_t>& blob, const T* src, size_t n, size_t align = 64) {
  size_t off = (blob.size() + align - 1) / align * align;
  blob.resize(off + sizeof(T) * (n ? n : 1), 0);
  if (n) memcpy(blob.data() + off, src, sizeof(T) * n);
  return (int64_t)off;
}
}  // namespace cim_layout_detail

// Safe per-vessel bound on the number of unrolled stops (cim_data_generator.py:18-115): every
// stop advances the clock by at least max(1,ceil(duration-noise)) + ceil(dist/(speed+noise)).
inline int cim_stop_bound(const mrx_cim_topology* t, int v, int max_tick) {
  int r = t->vessel_route[v];
  int L = t->route_offset[r + 1] - t->route_offset[r];
  double cyc = 0;
  for (int i = 0; i < L; i++) {
    double park = ceil(t->vessel_duration[v] - t->vessel_duration_noise[v]);
    if (park < 1) park = 1;
    double sp = t->vessel_speed[v] + t->vessel_speed_noise[v];
    double sail = sp > 0 ? ceil(t->route_dist[t->route_offset[r] + i] / sp) : 1;
    if (sail < 0) sail = 0;
    cyc += park + sail;
  }
  if (cyc < 1) cyc = 1;
  double cycles = ceil((double)(max_tick + 1) / cyc) + 1;
  return (int)(cycles * L) + t->future_stop_number + 2;
}

// set by the engine library (not by the host-only emulator build): returns 1 when the device's fp64 division does not match
// cim::of_div (see the order_fast conditions below), 0 when it does or when no device is present
typedef int (*order_fast_probe_fn)();
inline order_fast_probe_fn& order_fast_probe() {
  static order_fast_probe_fn f = nullptr;
  return f;
}

inline int cim_plan(const mrx_cim_topology* t, const mrx_cim_config* c, CimHostPlan* pl, std::string* err) {
  using namespace cim_layout_detail;
  auto fail = [&](const char* m) { if (err) *err = m; return (int)MRX_ERR_UNSUPPORTED; };
  if (!t || !c) { if (err) *err = "null topology/config"; return MRX_ERR_INVALID_ARG; }
  if (c->n_envs <= 0 || c->durations <= 0 || c->snapshot_resolution <= 0) { if (err) *err = "n_envs, durations and snapshot_resolution must be positive"; return MRX_ERR_INVALID_ARG; }
  if (t->n_ports <= 0 || t->n_ports > 64) return fail("engine limit: 1..64 ports (one lane per port)");
  if (t->n_vessels <= 0 || t->n_vessels > 64) return fail("engine limit: 1..64 vessels (one lane per vessel)");
  if (t->container_volume <= 0) return fail("container_volume must be positive");
  if (c->n_envs >= (1 << 30)) return fail("engine limit: n_envs < 2^30");
  // the topology arrives through the C ABI: every index / offset array is range-checked before it is used as an index
  {
    auto bad = [&](const char* m) { if (err) *err = m; return (int)MRX_ERR_INVALID_ARG; };
    if (t->period < 1 || !t->order_dist) return bad("topology: period must be >= 1 (order_dist[period])");
    if (t->n_routes <= 0 || t->n_targets < 0 || t->n_route_points <= 0) return bad("topology: n_routes / n_route_points must be positive, n_targets >= 0");
    if (!t->route_offset || !t->route_port || !t->target_offset || !t->vessel_route || !t->vessel_start_offset) return bad("topology: null index array");
    if (t->n_targets > 0 && !t->target_port) return bad("topology: null target_port");
    if (t->route_offset[0] != 0 || t->route_offset[t->n_routes] != t->n_route_points) return bad("topology: route_offset must run from 0 to n_route_points");
    for (int r = 0; r < t->n_routes; r++) if (t->route_offset[r + 1] < t->route_offset[r]) return bad("topology: route_offset must be non-decreasing");
    for (int i = 0; i < t->n_route_points; i++) if (t->route_port[i] < 0 || t->route_port[i] >= t->n_ports) return bad("topology: route_port out of range");
    if (t->target_offset[0] != 0 || t->target_offset[t->n_ports] != t->n_targets) return bad("topology: target_offset must run from 0 to n_targets");
    for (int p = 0; p < t->n_ports; p++) if (t->target_offset[p + 1] < t->target_offset[p]) return bad("topology: target_offset must be non-decreasing");
    for (int i = 0; i < t->n_targets; i++) if (t->target_port[i] < 0 || t->target_port[i] >= t->n_ports) return bad("topology: target_port out of range");
    for (int v = 0; v < t->n_vessels; v++) {
      const int r = t->vessel_route[v];
      if (r < 0 || r >= t->n_routes) return bad("topology: vessel_route out of range");
      const int L = t->route_offset[r + 1] - t->route_offset[r];
      if (t->vessel_start_offset[v] < 0 || t->vessel_start_offset[v] >= (L > 0 ? L : 1)) return bad("topology: vessel_start_offset out of range");
    }
  }
  CimParams& k = pl->kp;
  memset(&k, 0, sizeof(k));
  const int P = t->n_ports, V = t->n_vessels, R = t->n_routes, NT = t->n_targets, NRP = t->n_route_points;
  k.n_envs = c->n_envs; k.P = P; k.V = V; k.R = R; k.NT = NT; k.NRP = NRP;
  k.past_n = t->past_stop_number; k.future_n = t->future_stop_number;
  if (k.past_n < 0 || k.future_n < 0 || k.past_n > 16 || k.future_n > 16) return fail("stop_number out of range (0..16)");
  k.vrows = 10;  // the scalar vessel attributes; the four stop lists are derived at query time (cim::stop_list_value)
  k.start_tick = c->start_tick; k.T = c->start_tick + c->durations; k.resolution = c->snapshot_resolution;
  k.max_actions = c->max_actions > 0 ? c->max_actions : 1;
  if (c->decision_mode < 0 || c->decision_mode > 2) { if (err) *err = "decision_mode must be 0, 1 or 2"; return MRX_ERR_INVALID_ARG; }
  k.decision_mode = c->decision_mode;
  k.period = t->period; k.vol = t->container_volume; k.total_containers = t->total_containers; k.order_mode = t->order_mode;
  k.sample_noise = t->sample_noise;
  int total_frames = (int)ceil((double)c->durations / (double)c->snapshot_resolution);
  k.S = c->max_snapshots > 0 ? c->max_snapshots : total_frames;
  // frame layout
  k.f_ports = 0; k.f_vessels = PA_COUNT * P; k.f_fop = k.f_vessels + k.vrows * V; k.f_fov = k.f_fop + NT;  // full_on_ports is stored per (src, dst) order pair: every other cell of the P x P matrix is always 0
  // compact full_on_vessels / vessel_plans: one cell per (vessel, distinct port on its route)
  std::vector<int32_t> v_cbase(V), route_cidx(NRP ? NRP : 1), cidx_dense((size_t)V * P, -1);
  {
    std::vector<int> route_nd(R, 0);
    for (int r = 0; r < R; r++) {
      std::vector<int> seen;
      for (int i = t->route_offset[r]; i < t->route_offset[r + 1]; i++) {
        int c = -1;
        for (size_t j = 0; j < seen.size(); j++) if (seen[j] == t->route_port[i]) c = (int)j;
        if (c < 0) { c = (int)seen.size(); seen.push_back(t->route_port[i]); }
        route_cidx[i] = c;
      }
      route_nd[r] = (int)seen.size();
    }
    int nc = 0;
    for (int v = 0; v < V; v++) {
      const int r = t->vessel_route[v];
      v_cbase[v] = nc;
      for (int i = t->route_offset[r]; i < t->route_offset[r + 1]; i++) cidx_dense[(size_t)v * P + t->route_port[i]] = nc + route_cidx[i];
      nc += route_nd[r];
    }
    k.NC = nc;
  }
  k.f_plans = k.f_fov + k.NC; k.FW = k.f_plans + k.NC;
  k.FW = (k.FW + 3) / 4 * 4;  // 16-byte rows for vector copies
  // RNG usage (draws with zero noise cannot influence any value)
  bool on = false, bn = false;
  for (int p = 0; p < P; p++) { on |= t->source_noise[p] != 0; bn |= t->empty_return_noise[p] != 0 || t->full_return_noise[p] != 0; }
  for (int i = 0; i < NT; i++) on |= t->target_noise[i] != 0;
  k.use_order_rng = on; k.use_buffer_rng = bn;
  bool oi = false;
  if (t->sample_noise != 0) for (int tk = 0; tk < k.T; tk++) oi |= t->order_dist[tk % t->period] != 0;
  k.has_order_init = oi;
  k.idx_order_init = oi ? 0 : -1; k.idx_route = oi ? 1 : 0; k.idx_order_num = k.idx_route + 1; k.idx_buffer = k.idx_route + 2;
  // data read from files (dump folder / real data files): nothing is generated, so the streams are created lazily by
  // the first episode in this order (fresh process): order_number at the first get_orders, buffer_time at the first
  // buffer draw; the real-data container never touches order_number (cim_data_container.py:250-252, 304-307, 422-443)
  k.data_mode = t->data_mode; k.data_T = t->data_max_tick; k.data_seed = t->seed;
  if (t->data_mode) {
    if (t->data_mode != 1 && t->data_mode != 2) return fail("data_mode must be 0, 1 or 2");
    if (!t->fixed_n_stops || !t->fixed_stops_arrival || !t->fixed_stops_leave || !t->fixed_vessel_period || t->fixed_max_stops <= 0)
      return fail("data_mode != 0 needs the fixed stop tables");
    if (k.T > t->data_max_tick) return fail("start_tick + durations exceeds the data set's max_tick");
    k.has_order_init = 0; k.idx_order_init = -1;
    if (t->data_mode == 1) { if (!t->fixed_order_prop) return fail("dump data needs fixed_order_prop"); k.idx_order_num = 0; k.idx_buffer = 1; k.idx_route = 2; }
    else { if (!t->fixed_orders) return fail("real data needs fixed_orders"); k.idx_buffer = 0; k.idx_order_num = 1; k.idx_route = 2; k.use_order_rng = 0; }
  }
  // pending-return horizon
  int H = 1;
  for (int p = 0; p < P; p++) {
    int a = (int)ceil(t->empty_return_base[p] + fabs(t->empty_return_noise[p])) + 1;
    int b = (int)ceil(t->full_return_base[p] + fabs(t->full_return_noise[p])) + 1;
    if (a > H) H = a;
    if (b > H) H = b;
  }
  if (H > 64) return fail("engine limit: return buffer ticks must be < 64");
  k.H = H;
  // stop-table capacity
  int smax = 0;
  for (int v = 0; v < V; v++) {
    int b = cim_stop_bound(t, v, k.T);
    if (b > smax) smax = b;
    if (ceil(t->vessel_duration[v] + fabs(t->vessel_duration_noise[v])) > 255) return fail("engine limit: parking duration must be <= 255 ticks");
    if (t->vessel_speed[v] - fabs(t->vessel_speed_noise[v]) <= 0) return fail("sailing speed minus noise must stay positive");
  }
  if (c->max_stops > 0) smax = c->max_stops;
  if (t->data_mode) smax = t->fixed_max_stops;
  k.SMAX = (smax + 3) / 4 * 4;
  if (k.T >= (1 << 23)) return fail("engine limit: max_tick < 2^23");
  if (t->total_containers >= (1 << 24)) return fail("engine limit: total_containers < 2^24");
  // private state
  k.pv_evt = PH_COUNT; k.pv_next = k.pv_evt + V; k.pv_pos = k.pv_next + V; k.pv_krl = k.pv_pos + V; k.pv_period = k.pv_krl + V; k.pv_rempty = k.pv_period + V;
  k.PWH = (k.pv_rempty + H * P + 3) / 4 * 4;   // the head: what every build stages in LDS (16-byte rows)
  k.pv_rfull = k.PWH;                          // pending full returns [H][NT]: LDS in the generic layout, registers in a lean build
  k.PW = (k.pv_rfull + H * NT + 3) / 4 * 4;
  // derived integer tables
  std::vector<int32_t> pair_src(NT ? NT : 1), v_route_base(V), v_route_len(V), v_total_space(V), leg_off(V + 1), leg_time,
      v_period(V), er_delay(P), fr_delay(P), rec_off(V + 1);
  for (int p = 0; p < P; p++) {
    for (int j = t->target_offset[p]; j < t->target_offset[p + 1]; j++) pair_src[j] = p;
    er_delay[p] = (int)ceil(t->empty_return_base[p]);  // apply_noise(base, 0) == base
    fr_delay[p] = (int)ceil(t->full_return_base[p]);
  }
  int rec_w = 0;
  for (int v = 0; v < V; v++) {
    int r = t->vessel_route[v];
    int L = t->route_offset[r + 1] - t->route_offset[r];
    if (L <= 0 || L > 62) return fail("engine limit: route length 1..62");
    v_route_base[v] = t->route_offset[r]; v_route_len[v] = L;
    v_total_space[v] = (int)floor((double)t->vessel_capacity[v] / (double)t->container_volume);  // vessel.py:69
    leg_off[v] = (int)leg_time.size();
    int per = 0;
    for (int i = 0; i < L; i++) {
      // vessel_future_stops_prediction.py:72 / cim_data_generator.py:92: duration + ceil(dist / speed)
      int leg = (int)(t->vessel_duration[v] + ceil(t->route_dist[t->route_offset[r] + i] / t->vessel_speed[v]));
      leg_time.push_back(leg);
      per += leg;
    }
    v_period[v] = per;  // cim_data_generator.py:93-101: sum over the first route_length stops = one full cycle
    rec_off[v] = rec_w;
    rec_w += (L + 1) * (L + 1);
  }
  leg_off[V] = (int)leg_time.size(); rec_off[V] = rec_w;
  k.REC_W = (rec_w + 3) / 4 * 4;
  // LDS plan (word offsets; doubles 8-byte aligned)
  // Order table (mrx_cim_config.order_table): only `fixed` order mode is state independent
  k.pregen = (t->order_mode == 0 && NT > 0 && c->order_table >= 0) ? 1 : 0;
  if (t->data_mode == 2) {  // orders come from a file: the table is the input format (one copy shared by every env)
    if (c->order_table < 0) return fail("real data files need the order table (order_table >= 0)");
    k.pregen = NT > 0 ? 1 : 0;
  }
  k.NTP = (NT + 7) / 8 * 8;   // (16-byte rows for either element size; padding uint16 rows to whole 128-byte lines measured no difference)
  // uint16 elements only when the plan PROVES every quantity <= 65535.  The proof needs two facts:
  //  (1) no noised ratio can be negative: apply_noise is base + uniform(-noise, noise) with no clipping (utils.py:30-42), so every
  //      source and target base must be >= |noise|.  Then list_sum_normalize yields ratios in [0, 1], ceil(n * ratio) <= n, and
  //      cim_data_container.py:354-393 gives cur_num <= cur_port_order_num <= remaining_orders <= orders_to_gen.  (With a negative
  //      ratio `remaining_orders` GROWS and a later ratio can be far above 1: no bound exists, so those plans keep int32.)
  //  (2) orders_to_gen = order_proportion[tick] <= max over the period of floor(clip(dist + noise, 0, 1) * total_containers)
  //      (parsers.py:57-106) <= 65535.
  k.order_half = 0;
  if (k.pregen && t->data_mode != 2) {
    double mx = 0;
    if (t->data_mode == 1) { for (int i = 0; i < t->data_max_tick; i++) if (t->fixed_order_prop[i] > mx) mx = t->fixed_order_prop[i]; }
    else {
      for (int i = 0; i < t->period; i++) { double c = t->order_dist[i] + fabs(t->sample_noise); c = c > 1 ? 1 : c; if (c * (double)t->total_containers > mx) mx = c * (double)t->total_containers; }
    }
    bool nonneg = true;
    for (int p = 0; p < P; p++) if (!(t->source_base[p] >= fabs(t->source_noise[p]))) nonneg = false;
    for (int i = 0; i < NT; i++) if (!(t->target_base[i] >= fabs(t->target_noise[i]))) nonneg = false;
    k.order_half = (nonneg && mx <= 65535.0) ? 1 : 0;
  }
  const int dsrc_w = 2 * ((P + 1) / 2 * 2);
  const int dtgt_w = 2 * (NT + 1) > 3 * 64 ? 2 * (NT + 1) : 3 * 64;
  int w = 0;
  k.l_frame = w; w += k.FW;
  k.l_priv = w; w += k.PWH;
  if (!k.pregen) { k.l_mt0 = w; w += MT_WORDS; }  // with the order table the order stream and the generator's fp64
  k.l_mt1 = w; w += MT_WORDS;                    // scratch live in the reset / order-table kernels' LDS only (see below)
  w = (w + 1) / 2 * 2;
  if (!k.pregen) { k.l_dsrc = w; w += dsrc_w; }
  k.misc_cap = dtgt_w / 3;
  if (!k.pregen) { k.l_dtgt = w; w += dtgt_w; k.l_misc = k.l_dtgt; }  // dtgt doubles as the discharge-record merge list:
  else {  // (key, v, q) x misc_cap in phase B2, then the NT-entry prefix array of phase B3
    const int mw = NT + 1 > 3 * 64 ? (NT + 2) / 2 * 2 : 3 * 64;
    k.l_misc = w; w += mw; k.misc_cap = mw / 3;
  }
  k.l_srcn = w; w += P;
  w = (w + 1) / 2 * 2;
  k.lds_words = (w + 3) / 4 * 4;   // (the staged tables and the two per-pair arrays follow: below, once ctab_words is known)

  // ---- constant blob
  std::vector<uint8_t>& B = pl->const_blob;
  B.clear();
  // Each table is appended to the blob and remembered as (field of kp, relative offset); cim_plan_bind()
  // turns them into device pointers.  "ctab" = the contiguous block the step kernel stages in LDS: the six
  // per-port fp64 tables and the int tables that are read inside serial (wave-uniform) loops.
  std::vector<std::pair<size_t, int64_t>>& binds = pl->binds;
  binds.clear();
  size_t align = 64;  // tables outside the LDS-staged block are padded to cache-line-ish boundaries, inside it packed
  auto put_d = [&](const double* CimParams::*f, const double* src, size_t n) {
    binds.push_back({(size_t)((char*)&(k.*f) - (char*)&k), blob_put(B, src, n, align < 8 ? 8 : align)});
  };
  auto put_i = [&](const int32_t* CimParams::*f, const int32_t* src, size_t n) {
    binds.push_back({(size_t)((char*)&(k.*f) - (char*)&k), blob_put(B, src, n, align)});
  };
  put_d(&CimParams::tgt_base, t->target_base, NT); put_d(&CimParams::tgt_noise, t->target_noise, NT);
  put_d(&CimParams::v_speed, t->vessel_speed, V); put_d(&CimParams::v_speed_noise, t->vessel_speed_noise, V);
  put_d(&CimParams::v_dur, t->vessel_duration, V); put_d(&CimParams::v_dur_noise, t->vessel_duration_noise, V);
  put_d(&CimParams::route_dist, t->route_dist, NRP); put_d(&CimParams::order_dist, t->order_dist, t->period);
  // int32 originals (global-memory users: reset kernel, fast path, query)
  put_i(&CimParams::tgt_off, t->target_offset, P + 1); put_i(&CimParams::tgt_port, t->target_port, NT);
  put_i(&CimParams::route_port, t->route_port, NRP);
  put_i(&CimParams::v_route_base, v_route_base.data(), V); put_i(&CimParams::v_route_len, v_route_len.data(), V);
  put_i(&CimParams::leg_off, leg_off.data(), V + 1); put_i(&CimParams::leg_time, leg_time.data(), leg_time.size());
  put_i(&CimParams::rec_off, rec_off.data(), V + 1);
  put_i(&CimParams::v_cbase, v_cbase.data(), V); put_i(&CimParams::route_cidx, route_cidx.data(), NRP);
  if (k.pregen) { put_d(&CimParams::src_base, t->source_base, P); put_d(&CimParams::src_noise, t->source_noise, P); }
  const int64_t ctab_begin = (int64_t)((B.size() + 63) / 64 * 64);
  B.resize((size_t)ctab_begin, 0);
  align = 4;
  // with the order table the source ratios are read by the order-table kernel only (from global memory)
  if (!k.pregen) { put_d(&CimParams::src_base, t->source_base, P); put_d(&CimParams::src_noise, t->source_noise, P); }
  put_d(&CimParams::er_base, t->empty_return_base, P); put_d(&CimParams::er_noise, t->empty_return_noise, P);
  put_d(&CimParams::fr_base, t->full_return_base, P); put_d(&CimParams::fr_noise, t->full_return_noise, P);
  if (!k.use_buffer_rng) { put_i(&CimParams::er_delay, er_delay.data(), P); put_i(&CimParams::fr_delay, fr_delay.data(), P); }
  {
    bool fits = true;
    auto put_h = [&](const uint16_t* CimParams::*f, const int32_t* src, size_t n) {
      std::vector<uint16_t> h(n ? n : 1, 0);
      for (size_t i = 0; i < n; i++) { if (src[i] < 0 || src[i] > 65535) fits = false; h[i] = (uint16_t)src[i]; }
      binds.push_back({(size_t)((char*)&(k.*f) - (char*)&k), blob_put(B, h.data(), h.size(), 2)});
    };
    put_h(&CimParams::h_tgt_off, t->target_offset, P + 1);  // (target_port is only read by the reset kernel, from the int32 original)
    put_h(&CimParams::h_route_port, t->route_port, NRP);
    put_h(&CimParams::h_v_route_base, v_route_base.data(), V); put_h(&CimParams::h_v_route_len, v_route_len.data(), V);
    put_h(&CimParams::h_leg_off, leg_off.data(), V + 1); put_h(&CimParams::h_leg_time, leg_time.data(), leg_time.size());
    put_h(&CimParams::h_rec_off, rec_off.data(), V + 1);
    put_h(&CimParams::h_v_cbase, v_cbase.data(), V); put_h(&CimParams::h_route_cidx, route_cidx.data(), NRP);
    put_h(&CimParams::h_pair_src, pair_src.data(), NT);
    if (!fits) return fail("engine limit: a topology table entry (offsets, leg times) exceeds 65535");
  }
  B.resize((B.size() + 63) / 64 * 64, 0);
  const int64_t ctab_end = (int64_t)B.size();
  align = 64;
  if (k.use_buffer_rng) { put_i(&CimParams::er_delay, er_delay.data(), P); put_i(&CimParams::fr_delay, fr_delay.data(), P); }
  put_i(&CimParams::v_cap, t->vessel_capacity, V); put_i(&CimParams::v_init_empty, t->vessel_init_empty, V);
  put_i(&CimParams::p_cap, t->port_capacity, P); put_i(&CimParams::p_init_empty, t->port_init_empty, P);
  put_i(&CimParams::v_period, v_period.data(), V); put_i(&CimParams::v_route, t->vessel_route, V);
  put_i(&CimParams::pair_src, pair_src.data(), NT);
  put_i(&CimParams::v_start, t->vessel_start_offset, V); put_i(&CimParams::v_total_space, v_total_space.data(), V);
  put_i(&CimParams::cidx_dense, cidx_dense.data(), (size_t)V * P);
  {
    std::vector<int32_t> pair_dense((size_t)P * P, -1);
    for (int p = 0; p < P; p++)
      for (int j = t->target_offset[p]; j < t->target_offset[p + 1]; j++) pair_dense[(size_t)p * P + t->target_port[j]] = j;
    put_i(&CimParams::pair_dense, pair_dense.data(), (size_t)P * P);
  }
  {  // lean_tab (cim_params.h): what a lean step kernel keeps in registers for the whole step
    std::vector<int32_t> lt((size_t)k.NTP + 5 * 64, 0);
    for (int j = 0; j < NT; j++) lt[j] = pair_src[j] | ((t->target_offset[pair_src[j]] & 0xffffff) << 8);
    for (int v = 0; v < V; v++) {
      const int nd = (v + 1 < V ? v_cbase[v + 1] : k.NC) - v_cbase[v];
      lt[(size_t)k.NTP + v] = v_route_len[v] | (nd << 6) | (v_route_base[v] << 12);
      lt[(size_t)k.NTP + 64 + v] = leg_off[v] | (rec_off[v] << 16);
      lt[(size_t)k.NTP + 128 + v] = v_cbase[v];
      lt[(size_t)k.NTP + 192 + v] = t->vessel_capacity[v];
    }
    for (int p = 0; p < P; p++) lt[(size_t)k.NTP + 256 + p] = t->target_offset[p] | ((t->target_offset[p + 1] - t->target_offset[p]) << 16);
    put_i(&CimParams::lean_tab, lt.data(), lt.size());
  }
  int64_t shared_orders_rel = -1;
  if (t->data_mode) {
    std::vector<uint32_t> fx((size_t)V * k.SMAX, 0);
    for (int v = 0; v < V; v++) {
      if (t->fixed_n_stops[v] < 2 || t->fixed_n_stops[v] > t->fixed_max_stops) return fail("every vessel needs at least two stops in the data set");
      for (int i = 0; i < t->fixed_n_stops[v]; i++) {
        const int a = t->fixed_stops_arrival[(size_t)v * t->fixed_max_stops + i], l = t->fixed_stops_leave[(size_t)v * t->fixed_max_stops + i];
        if (a < 0 || a >= (1 << 23) || l - a <= 0 || l - a > 255) return fail("engine limit: stop arrival < 2^23 and parking 1..255 ticks");
        fx[(size_t)v * k.SMAX + i] = ((uint32_t)a << 8) | (uint32_t)(l - a);
      }
    }
    auto put_u = [&](const uint32_t* CimParams::*f, const uint32_t* src, size_t n) {
      binds.push_back({(size_t)((char*)&(k.*f) - (char*)&k), blob_put(B, src, n, 64)});
    };
    put_u(&CimParams::fx_stops, fx.data(), fx.size());
    put_i(&CimParams::fx_nstops, t->fixed_n_stops, V); put_i(&CimParams::fx_vperiod, t->fixed_vessel_period, V);
    if (t->data_mode == 1) put_i(&CimParams::fx_order_prop, t->fixed_order_prop, t->data_max_tick);
    if (t->data_mode == 2 && k.pregen) {
      std::vector<int32_t> tab((size_t)c->durations * k.NTP, 0);
      for (int d = 0; d < c->durations; d++)
        for (int j = 0; j < NT; j++) tab[(size_t)d * k.NTP + j] = t->fixed_orders[(size_t)(c->start_tick + d) * NT + j];
      shared_orders_rel = blob_put(B, tab.data(), tab.size(), 256);
    }
  }
  pl->ctab_rel = ctab_begin;
  k.ctab_words = (int)((ctab_end - ctab_begin) / 4);
  k.l_ctab = (k.lds_words + 3) / 4 * 4;   // (16-byte aligned: also the stride between the envs of a multi-wave workgroup)
  // LEAN builds (cim_device.h MRX_LEAN) keep the two arrays that are only ever indexed by "my lane's order pair" — the pending
  // full returns [H][NT] and the tick's order quantities [NT] — in registers; they come LAST in the LDS layout, so that a lean
  // launch simply reserves lds_words_lean and every other offset is the same for both kinds of build.
  k.lds_words_lean = (k.l_ctab + k.ctab_words + 3) / 4 * 4;
  k.l_rfull = k.lds_words_lean;
  k.l_oq = k.l_rfull + (H * NT + 3) / 4 * 4;  // order quantity | (buffer ticks + 1) << 24
  k.l_odelay = k.l_oq;
  k.lds_words = (k.l_oq + NT + 1 + 3) / 4 * 4;
  k.lean_ok = (k.pregen && NT <= 192 && H <= 4) ? 1 : 0;
  // (lean_tab packs route base / leg offset / record offset into 16-19 bit fields: checked by the 16-bit table copies above)
  if (NRP >= (1 << 19) || rec_w >= (1 << 15)) k.lean_ok = 0;
  if (k.SMAX >= (1 << 16)) k.lean_ok = 0;   // (the lean tick keeps a vessel's next stop index in 16 bits of its row word: cim_device.h vr_set / vr_k)
  // Envs per workgroup of the plan-specialised step kernel.  Every wave still owns one env and never talks to another one,
  // but the k waves of a workgroup share ONE staged copy of the topology tables: k * l_ctab + ctab_words words per workgroup.
  // gfx950 hands out LDS in 1280-byte granules, 128 per CU (measured: tools/hbm_pattern_bench --residency); take the
  // smallest k <= 4 that maximises the resident waves (global_trade.22p: 10 one-wave workgroups, or 4 x 3 waves = 12).
  {
    int best_k = 1, best_waves = 0;
    for (int kk = 1; kk <= 4; kk++) {
      const long long bytes = ((long long)kk * k.l_ctab + k.ctab_words + 3) / 4 * 4 * 4;
      const long long gran = (bytes + 1279) / 1280;
      const int waves = gran > 128 ? 0 : (int)(128 / gran) * kk;
      if (waves > best_waves) { best_waves = waves; best_k = kk; }
    }
    // Measured on global_trade.22p (profiles/r03_lds_diet.md): 4 x 3 waves = 12 waves per CU run no faster than 10 one-wave
    // workgroups (262 vs 261 M env-steps/s; 5 x 2 waves: 246 M) — a workgroup's LDS is only released when its slowest wave
    // is done, which costs what the extra residency gains.  So the default stays one env per workgroup; the knob remains.
    best_k = 1;
    if (const char* ev = getenv("MRX_CIM_WG_WAVES")) { const int v = atoi(ev); if (v >= 1 && v <= 16 && k.lean_ok) best_k = v; }  // experiments (lean layout only: the envs' blocks must end at l_ctab)
    k.wg_waves = best_k;
  }
  // reset_env: four RNG streams side by side (order-init, route, order, buffer: all seeded, the data generated, three of them
  // persisted), and only THEN frame and private head are initialised in the same LDS words — the reset kernel is a latency chain
  // (route unrolling: ~5000 sequential stops per env), its throughput is its occupancy: 10 KB of LDS = 16 waves per CU
  k.l_mt2 = 0; k.l_mt3 = MT_WORDS; k.r_mt0 = 2 * MT_WORDS; k.r_mt1 = 3 * MT_WORDS;
  k.lds_words_reset = 4 * MT_WORDS > k.FW + k.PWH ? 4 * MT_WORDS : (k.FW + k.PWH + 3) / 4 * 4;
  if (k.pregen) {
    // the order-table kernel: order RNG state, generator scratch, staged tables
    int g = 0;
    k.g_mt0 = g; g += MT_WORDS;
    k.g_dsrc = g; g += dsrc_w;
    k.g_dtgt = g; g += dtgt_w;
    k.g_oq = g; g += NT + 1;
    k.g_srcn = g; g += P;
    g = (g + 1) / 2 * 2;
    k.g_srctab = g; g += 4 * P;  // source ratio tables (base, noise) as doubles
    k.g_ctab = (g + 3) / 4 * 4;
    k.lds_words_gen = (k.g_ctab + k.ctab_words + 3) / 4 * 4;
    // cim::gen_order_table_fast (branch-free ticks, ~a third of the generic generator's instructions) relies on:
    //  * the uint16 table's proof (order_half: no noised ratio can be negative, every quantity <= 65535 — so all integer work
    //    fits int32 and `remaining` only shrinks), with a MARGIN: every noised ratio is exactly 0 or >= 2^-40 and every list's sum
    //    is >= 2^-40, so no sum is zero and every quotient x / sum has x = 0 or both operands in [2^-40, 2^20] — the range in
    //    which the hardware's fp64 division expansion applies no scaling and the per-port shared reciprocal reproduces it bit for bit;
    //  * at most 192 draws per tick (three per lane) and a lane left for the source sum (P <= 63);
    //  * the order stream in use (noise somewhere) and data generated on the device or from a dump (data_mode 0 / 1).
    // MRX_ORDER_FAST=0 in the environment of the PLANNING process keeps the generic generator (tests compare the two).
    k.order_fast = 0;
    {
      const char* ev = getenv("MRX_ORDER_FAST");
      bool ok = k.order_half && k.use_order_rng && P <= 63 && P + NT <= 192 && !(ev && atoi(ev) == 0);
      // the shared-reciprocal division is bit-exact only while the compiler's fp64 division IS the expansion restated in
      // cim::of_recip / of_div: where a device is present, the engine library checks that once per process on the device itself
      // (cim_engine.hip: mrx_k_cim_of_selfcheck) and vetoes the generator on a mismatch (a toolchain that lowers fdiv differently)
      if (ok && order_fast_probe()) {
        static const int veto = order_fast_probe()();
        if (veto) ok = false;
      }
      const double margin = 9.094947017729282e-13, cap = 1048576.0;  // 2^-40, 2^20
      // (an entry may also be exactly zero — base 0, noise 0: a pure destination port — as long as its list's sum is not)
      auto entry_ok = [&](double b, double n) { return (b == 0.0 && n == 0.0) || b - fabs(n) >= margin; };
      double tot = 0, tot_min = 0;
      for (int p = 0; p < P && ok; p++) {
        if (!entry_ok(t->source_base[p], t->source_noise[p])) ok = false;
        tot += t->source_base[p] + fabs(t->source_noise[p]);
        tot_min += t->source_base[p] - fabs(t->source_noise[p]);
      }
      if (!(tot <= cap && tot_min >= margin)) ok = false;
      for (int p = 0; p < P && ok; p++) {
        double ts = 0, ts_min = 0;
        for (int i = t->target_offset[p]; i < t->target_offset[p + 1]; i++) {
          if (!entry_ok(t->target_base[i], t->target_noise[i])) ok = false;
          ts += t->target_base[i] + fabs(t->target_noise[i]);
          ts_min += t->target_base[i] - fabs(t->target_noise[i]);
        }
        if (t->target_offset[p + 1] > t->target_offset[p] && !(ts <= cap && ts_min >= margin)) ok = false;
      }
      if (ok) {
        int slots = (P + 3) / 4 * 4;
        for (int p = 0; p < P; p++) slots += (t->target_offset[p + 1] - t->target_offset[p] + 3) / 4 * 4;
        k.gf_slots = slots;                  // [slots, slots + 4): zeros; [slots + 4, slots + 6): the idle lanes' scratch
        int f = 0;
        k.gf_win = f; f += 2 * MT_WORDS;
        k.gf_val = f; f += 2 * (slots + 6);
        f = (f + 3) / 4 * 4;
        k.gf_rec = f; f += 8 * (P + 1);
        k.gf_pre = f; f += NT + 2;
        f = (f + 3) / 4 * 4;
        k.gf_row = f; f += k.NTP / 2;
        k.gf_seg = f; f += P + 1;
        k.order_fast = 1;
        k.lds_words_gen = (f + 3) / 4 * 4;
      }
    }
  }
  if ((int64_t)k.lds_words_reset * 4 > 160 * 1024 || (int64_t)k.lds_words * 4 > 160 * 1024) return fail("engine limit: per-env state exceeds 160 KiB of LDS");

  // ---- workspace carve-up
  Arena A;
  const int64_t N = c->n_envs;
  pl->const_off = A.take((int64_t)B.size());
  pl->o_live = A.take(N * k.FW * 4);
  pl->o_ring = A.take(N * (int64_t)k.S * k.FW * 4);
  pl->o_ring_fi = A.take(N * k.S * 4);
  pl->o_priv = A.take(N * k.PW * 4);
  pl->o_rec = A.take(N * (int64_t)k.REC_W * 4);
  pl->o_status = A.take(N * 4);
  pl->o_tick = A.take(N * 4);
  pl->o_nstops = A.take(N * V * 4);
  pl->o_order_prop = A.take(N * (int64_t)k.T * 4);
  pl->o_mt = A.take(N * MTS_COUNT * MT_WORDS * 4);
  pl->o_stops = A.take(N * (int64_t)V * k.SMAX * 4);
  pl->o_seed = A.take(N * 8);
  pl->o_vperiod = A.take(N * V * 4);
  pl->o_hint = A.take(N + 64);  // (mrx_k_cim_schedule reads whole 16-byte pieces)
  pl->o_order = A.take(N * 4);
  pl->o_sched = A.take(64 + 64 * MRX_PIPE_MAX_WAVES);  // counters + dummies of cim::regs_load, then the scratch pieces of cim::regs_store
  pl->shared_orders_rel = shared_orders_rel;
  k.orders_stride = shared_orders_rel >= 0 ? 0 : (long long)c->durations * k.NTP;
  pl->o_orders = (k.pregen && shared_orders_rel < 0) ? A.take(N * (int64_t)c->durations * k.NTP * (k.order_half ? 2 : 4)) : 0;
  pl->workspace_bytes = align_up(A.top, 256);

  mrx_cim_layout& Lo = pl->layout;
  memset(&Lo, 0, sizeof(Lo));
  Lo.n_envs = c->n_envs; Lo.n_ports = P; Lo.n_vessels = V; Lo.frame_words = k.FW; Lo.ring_slots = k.S;
  Lo.max_stops = k.SMAX; Lo.horizon = k.H;
  Lo.frame_off_ports = k.f_ports; Lo.frame_off_vessels = k.f_vessels; Lo.frame_off_full_on_ports = k.f_fop;
  Lo.frame_off_full_on_vessels = k.f_fov; Lo.frame_off_vessel_plans = k.f_plans;
  Lo.off_live = pl->o_live; Lo.off_ring = pl->o_ring; Lo.off_ring_fi = pl->o_ring_fi; Lo.off_status = pl->o_status;
  Lo.off_tick = pl->o_tick; Lo.off_seed = pl->o_seed; Lo.off_stops = pl->o_stops; Lo.off_nstops = pl->o_nstops;
  Lo.off_order_prop = pl->o_order_prop; Lo.off_vessel_period = pl->o_vperiod;  // per env: depends on how many stops were unrolled
  Lo.off_orders = shared_orders_rel >= 0 ? pl->const_off + shared_orders_rel : pl->o_orders; Lo.order_row_words = k.NTP; Lo.order_table_on = k.pregen; Lo.order_elem_bytes = k.pregen ? (k.order_half ? 2 : 4) : 0;
  Lo.workspace_bytes = pl->workspace_bytes;
  return MRX_OK;
}

// Resolve device pointers once the workspace base address is known.
inline void cim_plan_bind(CimHostPlan* pl, void* base_) {
  uint8_t* base = (uint8_t*)base_;
  CimParams& k = pl->kp;
  uint8_t* cb = base + pl->const_off;
  for (auto& bd : pl->binds) *(const void**)((char*)&k + bd.first) = cb + bd.second;
  k.ctab = (const int32_t*)(cb + pl->ctab_rel);
  k.live = (int32_t*)(base + pl->o_live); k.ring = (int32_t*)(base + pl->o_ring); k.ring_fi = (int32_t*)(base + pl->o_ring_fi);
  k.priv = (int32_t*)(base + pl->o_priv); k.rec = (int32_t*)(base + pl->o_rec); k.status = (int32_t*)(base + pl->o_status);
  k.tick = (int32_t*)(base + pl->o_tick); k.nstops = (int32_t*)(base + pl->o_nstops);
  k.order_prop = (int32_t*)(base + pl->o_order_prop); k.mt = (uint32_t*)(base + pl->o_mt);
  k.stops = (uint32_t*)(base + pl->o_stops); k.seed = (int64_t*)(base + pl->o_seed);
  k.vperiod = (int32_t*)(base + pl->o_vperiod);
  k.hint = base + pl->o_hint; k.order = (int32_t*)(base + pl->o_order); k.sched = (int32_t*)(base + pl->o_sched);
  k.orders = !k.pregen ? nullptr : pl->shared_orders_rel >= 0 ? (int32_t*)(cb + pl->shared_orders_rel) : (int32_t*)(base + pl->o_orders);
}
