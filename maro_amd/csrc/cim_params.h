// cim_params.h — POD kernel-argument block shared by host (layout/C-ABI) and device code.
#pragma once
#include <stdint.h>

// Frame attribute ids (also the ids of mrx_cim_attr_id); names follow the reference schema
// (cim/port.py:9-46, cim/vessel.py:14-45, cim/matrix.py:24-28).
enum { PA_CAPACITY, PA_EMPTY, PA_FULL, PA_ON_SHIPPER, PA_ON_CONSIGNEE, PA_SHORTAGE, PA_ACC_SHORTAGE,
       PA_BOOKING, PA_ACC_BOOKING, PA_FULFILLMENT, PA_ACC_FULFILLMENT, PA_TRANSFER_COST, PA_COUNT };
enum { VA_CAPACITY, VA_EMPTY, VA_FULL, VA_REMAINING_SPACE, VA_EARLY_DISCHARGE, VA_IS_PARKING, VA_LOC_PORT_IDX,
       VA_ROUTE_IDX, VA_LAST_LOC_IDX, VA_NEXT_LOC_IDX, VA_PAST_STOP_LIST, VA_PAST_STOP_TICK_LIST,
       VA_FUTURE_STOP_LIST, VA_FUTURE_STOP_TICK_LIST, VA_COUNT };
enum { MA_FULL_ON_PORTS, MA_FULL_ON_VESSELS, MA_VESSEL_PLANS, MA_COUNT };

// private per-env header words (priv[0..PH))
enum { PH_TICK, PH_FLAGS, PH_PEND_LO, PH_PEND_HI, PH_CUR_VESSEL, PH_OPNUM_LO, PH_OPNUM_HI, PH_IDX_ORDER,
       PH_IDX_BUFFER, PH_IDX_ROUTE, PH_ZOMBIE_LO, PH_ZOMBIE_HI /* start_tick > 0: vessels whose first departure fell before start_tick */,
       PH_ACCB_LO, PH_ACCB_HI, PH_ACCS_LO, PH_ACCS_HI /* cached metrics: sums of acc_booking / acc_shortage */, PH_COUNT = 16 };
enum { FL_FRESH = 1, FL_FINISHED = 2, FL_LONG = 4 /* the env's next full-path step will advance two ticks or more (scheduling only) */ };

#define MT_WORDS 624
enum { MTS_ORDER = 0, MTS_BUFFER = 1, MTS_ROUTE = 2, MTS_COUNT = 3 };

struct CimParams {
  // ---- dimensions
  int n_envs, P, V, R, NT, NRP, past_n, future_n, vrows, FW, S, H, SMAX, T, start_tick, resolution,
      max_actions, period;
  int vol, total_containers, order_mode;
  int use_order_rng, use_buffer_rng, has_order_init;
  int idx_order_init, idx_route, idx_order_num, idx_buffer;  // SimRandom creation indices
  double sample_noise;
  // ---- frame offsets (words inside one frame)
  int f_ports, f_vessels, f_fop, f_fov, f_plans;
  int misc_cap;  // capacity (entries) of the discharge-record merge list
  int NC;  // compact matrix cells: sum over vessels of the distinct ports on its route (full_on_vessels / vessel_plans
           // are stored [vessel][route port] — every other cell of the dense V x P matrices is constant 0 / -1)
  // ---- private-state layout (words)
  int PW, PWH /* the head staged in LDS: everything but the pending full returns */, pv_evt, pv_next, pv_pos, pv_krl, pv_period, pv_rfull, pv_rempty;
  int REC_W;
  // ---- LDS layout (word offsets)
  int l_frame, l_priv, l_mt0, l_mt1, l_dsrc, l_dtgt, l_oq, l_odelay, l_srcn, l_misc, lds_words;
  int l_ctab, ctab_words;  // serial-access int tables staged in LDS by the step kernel
  int wg_waves;            // envs (= waves) per workgroup of the plan-specialised step kernel: they share ONE staged copy of the tables
  int l_mt2, l_mt3, r_mt0, r_mt1, lds_words_reset;  // reset kernel only: route / order-init / order / buffer streams (its own layout)
  int l_rfull, lds_words_lean, lean_ok;  // pending full returns in the generic LDS layout; what a lean launch reserves (cim_device.h MRX_LEAN)
  int decision_mode;  // 0 Sequential, 1 Joint, 2 JointWithSequentialAction (core.py:349-366)
  int data_mode;      // 0 generated at reset, 1 dump folder, 2 real data files (mrx_cim_topology.data_mode)
  int data_T;         // ticks covered by the fixed order proportion
  long long data_seed;
  long long orders_stride;  // ELEMENTS between two envs' order tables (0: one table shared by every env)
  int order_half;           // 1: the table holds uint16 elements (every quantity provably <= 65535), 0: int32
  const uint32_t* fx_stops;  // [V][SMAX] (arrival << 8 | parking)
  const int32_t *fx_nstops, *fx_vperiod, *fx_order_prop;
  int pregen, NTP;  // order table: rows of NTP (= NT rounded up to 4) words, one per tick of the episode
  int g_mt0, g_dsrc, g_dtgt, g_oq, g_srcn, g_srctab, g_ctab, lds_words_gen;  // LDS layout of the order-table kernel
  // the branch-free order-table generator (cim::gen_order_table_fast): 1 when the plan proves what it relies on (cim_layout.h);
  // its own LDS layout: a two-block MT window, the noised ratios in 4-padded summation segments, port records, the
  // hand-out prefix, the uint16 row
  int order_fast, gf_win, gf_val, gf_rec, gf_pre, gf_row, gf_seg, gf_slots;
  // ---- constant tables (device)
  const double *src_base, *src_noise, *tgt_base, *tgt_noise, *er_base, *er_noise, *fr_base, *fr_noise,
      *v_speed, *v_speed_noise, *v_dur, *v_dur_noise, *route_dist, *order_dist;
  const int32_t *tgt_off, *tgt_port, *pair_src, *route_port, *v_route_base, *v_route_len, *v_start, *v_cap,
      *v_init_empty, *v_total_space, *p_cap, *p_init_empty, *leg_off, *leg_time, *v_period, *er_delay,
      *fr_delay, *rec_off, *v_route, *v_cbase, *route_cidx, *cidx_dense, *pair_dense;
  const int32_t* lean_tab;  // per-lane static words of the lean step kernel, read ONCE per step into registers (cim_device.h LeanStat):
                            //   [k]              k < NTP      order pair k: source port | tgt_off[source] << 8     (lean plans: NT <= 192)
                            //   [NTP + v]        lane = vessel: route length | distinct route ports << 6 | route base << 12
                            //   [NTP + 64 + v]   leg_off | rec_off << 16
                            //   [NTP + 128 + v]  first compact matrix cell (v_cbase)
                            //   [NTP + 192 + v]  vessel capacity
                            //   [NTP + 256 + p]  lane = port: tgt_off[p] | number of its order pairs << 16
  const int32_t* ctab;  // start of the contiguous block staged in LDS by the step kernel (ctab_words words):
                        // per-port fp64 tables, er/fr_delay, and 16-bit copies of the serial-access int tables
  const uint16_t *h_tgt_off, *h_route_port, *h_v_route_base, *h_v_route_len, *h_leg_off, *h_leg_time,
      *h_rec_off, *h_v_cbase, *h_route_cidx, *h_pair_src;
  // ---- per-env state (device)
  int32_t *live, *ring, *ring_fi, *priv, *rec, *status, *tick, *nstops, *order_prop, *vperiod, *orders;
  uint32_t *mt, *stops;
  int64_t* seed;
  // ---- step scheduling (engine-owned, rewritten every step)
  uint8_t* hint;   // [n_envs] 1: the env's next step needs the full path (a tick will run / fresh); 0: it stays inside the
                   // current tick (another vessel's decision is pending) or the episode is over -> fast path
  int32_t* order;  // [n_envs] env ids of the coming step, full-path envs first (MRX_ORDER_TICK set), -1 padded
  int32_t* sched;  // [4] n_tick, n_active (mrx_k_cim_schedule)
};
#define MRX_ORDER_TICK 0x40000000  // order[] entry flag: full-path env
#define MRX_PIPE_MAX_WAVES 4096    // persistent step kernel: at most this many waves (each owns a 64-byte scratch line behind sched[16])

// The integer fields of CimParams that are constant for one (topology, config) plan: the set a specialised build
// (cim_spec.hip) receives as MRXC_<field> macros (KD() in cim_device.h).
#define MRX_CIM_DIM_FIELDS(X) \
  X(P) \
  X(V) \
  X(R) \
  X(NT) \
  X(NRP) \
  X(past_n) \
  X(future_n) \
  X(vrows) \
  X(FW) \
  X(S) \
  X(H) \
  X(SMAX) \
  X(T) \
  X(start_tick) \
  X(resolution) \
  X(max_actions) \
  X(period) \
  X(vol) \
  X(total_containers) \
  X(order_mode) \
  X(use_order_rng) \
  X(use_buffer_rng) \
  X(has_order_init) \
  X(idx_order_init) \
  X(idx_route) \
  X(idx_order_num) \
  X(idx_buffer) \
  X(f_ports) \
  X(f_vessels) \
  X(f_fop) \
  X(f_fov) \
  X(f_plans) \
  X(misc_cap) \
  X(NC) \
  X(PW) \
  X(PWH) \
  X(pv_evt) \
  X(pv_next) \
  X(pv_pos) \
  X(pv_krl) \
  X(pv_period) \
  X(pv_rfull) \
  X(pv_rempty) \
  X(REC_W) \
  X(l_frame) \
  X(l_priv) \
  X(l_mt0) \
  X(l_mt1) \
  X(l_dsrc) \
  X(l_dtgt) \
  X(l_oq) \
  X(l_odelay) \
  X(l_srcn) \
  X(l_misc) \
  X(lds_words) \
  X(l_ctab) \
  X(ctab_words) \
  X(wg_waves) \
  X(decision_mode) \
  X(data_mode) \
  X(data_T) \
  X(pregen) \
  X(order_half) \
  X(NTP) \
  X(l_mt2) \
  X(l_mt3) \
  X(r_mt0) \
  X(r_mt1) \
  X(l_rfull) \
  X(lds_words_lean) \
  X(lean_ok) \
  X(lds_words_reset) \
  X(g_mt0) \
  X(g_dsrc) \
  X(g_dtgt) \
  X(g_oq) \
  X(g_srcn) \
  X(g_srctab) \
  X(g_ctab) \
  X(lds_words_gen) \
  X(order_fast) \
  X(gf_win) \
  X(gf_val) \
  X(gf_rec) \
  X(gf_pre) \
  X(gf_row) \
  X(gf_seg) \
  X(gf_slots)

// Observation fused into the step kernel (mrx_cim_set_observation): per stepped env with a new decision,
//   ports  [n_envs][P][np] = snapshot_list["ports"][decision frame :: port attrs]
//   vessel [n_envs][nv]    = snapshot_list["vessels"][decision frame : decision vessel : vessel attrs]
struct CimObs {
  int np, nv;
  int pa[8], va[8];
  unsigned pa_packed;  // pa[] as 4-bit codes (lane-varying attribute index without indexing a kernel argument)
  int i_empty, i_tc;   // position of `empty` / `transfer_cost` in pa[] (-1: not requested)
  double *ports, *vessel;
  // Per-attribute retention (mrx_cim_set_port_history; runtime configuration, not part of a specialised plan): every
  // snapshot also stores the listed (integer) port attributes of its frame into hist [n_envs][hist_frames][hist_n][P]
  // — the whole episode of e.g. fulfillment / shortage (176 B per frame on 22p) next to a short ring for everything else.
  int hist_n, hist_frames;
  int hist_attr[4];
  int32_t* hist;
  // Device agent (mrx_cim_set_device_agent; runtime configuration): the step kernel ANSWERS the decision it has just raised —
  // the random legal agent of mrx_cim_random_policy, same draw — into the action buffers the next mrx_cim_step reads, so a
  // rollout with that agent is one launch per batch step.  agent_key >= 0: the draw is keyed on it (the step index the
  // separate launch would be given); < 0: on the decision's own (tick, vessel).
  int agent_mode;                 // 0 off, 1 random legal
  long long agent_key;
  int32_t *agent_actions, *agent_n_actions;   // [n_envs][max_actions][4], [n_envs]
  int32_t* agent_count;           // [n_envs] decisions answered (added to), or null
};
