// cim_device.h — device code of the batched CIM rollout engine (one env per wave64).
//
// Included after a wave-collective header: maro_amd/csrc/wave.h (gfx950, the product) or
// tests/emu/wave_emu.h (CPU fiber emulation used only by the test-suite).  Everything here is
// written against wave::{lane,sync,ballot,shfl,reduce_add}; there is no other platform switch.
//
// What it replaces in the reference (per env, per tick; SURVEY.md §9.2 gives the normative order):
//   EventBuffer.execute + Env._simulate       maro/event_buffer/event_buffer.py:190-247, simulator/core.py:317-381
//   CimBusinessEngine.step/post_step/handlers  simulator/scenarios/cim/business_engine.py:122-224, 448-748
//   CimSyntheticDataContainer._gen_orders      data_lib/cim/cim_data_container.py:309-398
//   gen_cim_data/_extend_route/order proportion data_lib/cim/cim_data_generator.py:18-205, parsers.py:57-106
//   SimRandom + CPython random.Random (MT19937) simulator/utils/sim_random.py:10-100
// The event queue is gone: a tick is a fixed sequence of phases over state staged in LDS —
//   A order generation  B1 departures  B2 due returns + discharges  B3 orders  B4 arrivals/loads
//   B5 decisions (one env-step each)   C post_step/snapshot.
#pragma once
#include <math.h>
#include <stdint.h>

#include "cim_params.h"

// KD(f): an integer dimension / layout offset of the plan (CimParams::f).  The generic kernels read it from the kernel
// arguments; a build specialised for one (topology, config) plan (cim_spec.hip, compiled at engine creation) turns every
// one of them into a compile-time constant MRXC_f — which removes ~60 live SGPRs and their spill traffic, folds the
// divisions / loop bounds / LDS offsets, and drops the VGPR count from ~197 to ~118.
// OD(f) / ODA(va, i): the same for the fused observation's configuration (CimObs: attribute counts and ids) — with np a
// constant the per-element `i / np` of the observation writer is a multiply, and the attribute ids need no scalar loads.
#ifdef MRX_SPECIALIZED
#define KD(f) (MRXC_##f)
#define OD(f) (MRXC_obs_##f)
#define ODA(f, i) (MRXC_obs_##f(i))
#else
#define KD(f) (K.f)
#define OD(f) (O.f)
#define ODA(f, i) (O.f[i])
#endif

namespace cim {

// Phase timer hook: a no-op in the product build; tools/profile build (-DMRX_PROFILE_PHASES) supplies a
// real cim::Prof before including this header to attribute wave cycles to the phases below.
#ifndef MRX_PROFILE_PHASES
struct Prof {
  MRX_DEV void mark(int) {}
  MRX_DEV void mark(int, long long) {}
  MRX_DEV void flush() {}
};
#endif
enum { PF_LOAD, PF_ACTION, PF_POST_STEP, PF_MT_LOAD, PF_ORDER_GEN, PF_DEPART_RETURNS, PF_ORDERS, PF_ARRIVALS, PF_OUTPUT,
       PF_STORE, PF_COUNT };

// ------------------------------------------------------------------------------------------
// serial-access topology tables: global (L2) by default, re-pointed at the LDS copy by the step kernel
struct Tabs {
  const uint16_t *tgt_off, *route_port, *v_route_base, *v_route_len, *leg_off, *leg_time, *rec_off, *v_cbase,
      *route_cidx, *pair_src;  // 16-bit copies (cim_plan checks the ranges)
  const int32_t *er_delay, *fr_delay;
  const double *src_base, *src_noise, *er_base, *er_noise, *fr_base, *fr_noise;
};

// LDS view of one env
struct Lds {
  Tabs tab;
  int32_t* frame;
  int32_t* priv;   // header, per-vessel rows, pending empty returns (PWH words)
  int32_t* rfull;  // pending full returns [H][NT] (generic layout; a lean build keeps them in registers: RingRegs)
  uint32_t* mt_ord;
  uint32_t* mt_buf;
  double* dsrc;
  double* dtgt;
  int32_t* oq;
  int32_t* srcn;
  int32_t* misc;
};

MRX_DEV Lds make_lds(const CimParams& K, int32_t* b) {
  Lds L;
  L.frame = b + KD(l_frame);
  L.priv = b + KD(l_priv);
  L.rfull = b + KD(l_rfull);
  L.mt_ord = (uint32_t*)(b + KD(l_mt0));
  L.mt_buf = (uint32_t*)(b + KD(l_mt1));
  L.dsrc = (double*)(b + KD(l_dsrc));
  L.dtgt = (double*)(b + KD(l_dtgt));
  L.oq = b + KD(l_oq);
  L.srcn = b + KD(l_srcn);
  L.misc = b + KD(l_misc);
  L.tab.tgt_off = K.h_tgt_off; L.tab.route_port = K.h_route_port;
  L.tab.v_route_base = K.h_v_route_base; L.tab.v_route_len = K.h_v_route_len;
  L.tab.leg_off = K.h_leg_off; L.tab.leg_time = K.h_leg_time;
  L.tab.er_delay = K.er_delay; L.tab.fr_delay = K.fr_delay; L.tab.rec_off = K.h_rec_off;
  L.tab.v_cbase = K.h_v_cbase; L.tab.route_cidx = K.h_route_cidx; L.tab.pair_src = K.h_pair_src;
  L.tab.src_base = K.src_base; L.tab.src_noise = K.src_noise; L.tab.er_base = K.er_base; L.tab.er_noise = K.er_noise;
  L.tab.fr_base = K.fr_base; L.tab.fr_noise = K.fr_noise;
  return L;
}

// LEAN builds (plan-specialised, order table on, at most 192 order pairs and 4 return-ring slots: CimParams::lean_ok): the two
// per-pair arrays that are only ever indexed by "my lane's pair" live in REGISTERS instead of LDS —
//   rf[h][b]  pending full returns of pair 64 b + lane in ring slot h (RING_FULL), loaded / stored straight from / to the
//             env's private HBM row,
//   oq[b]     the tick's order quantity of that pair (the order-table row, already prefetched into registers)
// — and the workgroup is launched with CimParams::lds_words_lean: 2.5 KB less LDS per env = 2 more resident envs per CU.
#if defined(MRX_SPECIALIZED)
#if MRXC_lean_ok
#define MRX_LEAN 1
#endif
#endif
#if defined(MRX_LEAN) && !defined(MRX_NO_LEAN2)
#define MRX_LEAN2 1  // the one-round-trip-per-phase tick (run_tick_lean) on top of the lean register layout
#endif
#ifdef MRX_LEAN
enum { RR_NB = (MRXC_NT + 63) / 64, RR_H = MRXC_H };
struct RingRegs { int rf[RR_H][RR_NB]; };
#else
struct RingRegs {};
#endif

// ... and so do the words a lane needs over and over (lean builds only):
//   LeanStat  static words of "my" order pairs / vessel / port (CimParams::lean_tab), loaded once per step straight from L2
//   VRows     the per-vessel scheduling words (lane = vessel), loaded once per step from the staged state and kept coherent with
//             it: only a vessel's own departure (phase B1) and arrival (B4) change them
// so that every phase of a tick starts with ONE LDS round trip (its gathers) instead of a chain of dependent table lookups.
#ifdef MRX_LEAN2
struct LeanStat {
  int ps[RR_NB];            // pair 64 b + lane: source port | tgt_off[source] << 8
  int vs0, vs1, vs2;  // lane = vessel: route length | distinct ports << 6 | route base << 12; leg_off | rec_off << 16; v_cbase
  int pt;                   // lane = port: tgt_off[p] | number of its pairs << 16
};
struct VRows {
  int evt, next;  // the vessel's next event tick, the cached arrival tick of its next stop
  int w;          // next_loc_idx | route position << 16 | next_loc mod (L + 1) << 22 | is_parking << 28   (stop tables hold < 65536 stops)
};
MRX_DEV int vr_k(const VRows& W) { return W.w & 0xffff; }
MRX_DEV int vr_pos(const VRows& W) { return (W.w >> 16) & 63; }
MRX_DEV int vr_krl(const VRows& W) { return (W.w >> 22) & 63; }
MRX_DEV int vr_park(const VRows& W) { return (W.w >> 28) & 1; }
MRX_DEV void vr_set(VRows& W, int k_, int pos_, int krl_, int park_) { W.w = k_ | (pos_ << 16) | (krl_ << 22) | (park_ << 28); }
struct Lean { LeanStat s; VRows w; };
#else
struct Lean {};
#endif

#define FP(a, p) L.frame[KD(f_ports) + (a) * KD(P) + (p)]
#define FV(a, v) L.frame[KD(f_vessels) + (a) * KD(V) + (v)]
// The four stop-list attributes of a vessel (past_stop_list / past_stop_tick_list / future_stop_list / future_stop_tick_list) are
// NOT stored in the frame: they are functions of the frame's next_loc_idx / last_loc_idx, the env's stop table and the static
// route tables (stop_list_value below; mrx_cim_query and the DQN state gather expand them on the fly), exactly like the constant
// cells of the dense matrices.  The frame carries the 10 scalar vessel rows only: 8064 -> 5488 bytes for global_trade.22p, i.e.
// a third less state to move per step and per snapshot, and a third less LDS per resident env.
#define FOPK(k) L.frame[KD(f_fop) + (k)] /* full_on_ports of order pair k = (src, dst) */
#define FOVC(v, c) L.frame[KD(f_fov) + T.v_cbase[v] + (c)]   /* full_on_vessels[v][c-th distinct route port] */
#define PLANC(v, c) L.frame[KD(f_plans) + T.v_cbase[v] + (c)] /* vessel_plans, same indexing */
#define V_EVT(v) L.priv[KD(pv_evt) + (v)]
#define V_NEXT(v) L.priv[KD(pv_next) + (v)]
#define V_POS(v) L.priv[KD(pv_pos) + (v)]  /* (start + next_loc) mod route_len */
#define V_KRL(v) L.priv[KD(pv_krl) + (v)]  /* next_loc mod (route_len + 1) */
#define V_PERIOD(v) L.priv[KD(pv_period) + (v)]  /* vessel_period_without_noise of this env's data */
#define U(x) wave::uniform(x)
// start_tick is not a multiple of the snapshot resolution: a frame's post_step snapshot then falls on an EARLY tick of the frame,
// and the pre-decision snapshots (core.py:345) of its later ticks are what stays in the snapshot list — they can no longer be
// aliased to the live frame, every decision writes one (and takes the full path: the fast path never stages the frame)
#define MRX_UNALIGNED_FRAMES (KD(resolution) > 1 && KD(start_tick) % KD(resolution) != 0)
#define RING_FULL(slot, k) L.rfull[(slot) * KD(NT) + (k)]
#define RING_EMPTY(slot, p) L.priv[KD(pv_rempty) + (slot) * KD(P) + (p)]

MRX_DEV float bits_f(int32_t x) { union { int32_t i; float f; } u; u.i = x; return u.f; }
MRX_DEV int32_t f_bits(float f) { union { int32_t i; float f; } u; u.f = f; return u.i; }

// ------------------------------------------------------------------------------------------
// MT19937 == CPython random.Random, state (624 words) in LDS, regenerated by the whole wave.
// stage the serial-access tables in LDS and re-point L.tab at the copy (complete with lds_dma_wait)
MRX_DEV void copy_in_async(int32_t* lds_dst, const int32_t* gsrc, int n_words);
MRX_DEV void stage_tables(const CimParams& K, Lds& L, int32_t* c) {
  copy_in_async(c, K.ctab, KD(ctab_words));
#define MRX_HTAB(f) L.tab.f = (const uint16_t*)c + (K.h_##f - (const uint16_t*)K.ctab)
  MRX_HTAB(tgt_off); MRX_HTAB(route_port); MRX_HTAB(v_route_base); MRX_HTAB(v_route_len);
  MRX_HTAB(leg_off); MRX_HTAB(leg_time); MRX_HTAB(rec_off); MRX_HTAB(v_cbase); MRX_HTAB(route_cidx); MRX_HTAB(pair_src);
#undef MRX_HTAB
  L.tab.er_delay = c + (K.er_delay - K.ctab); L.tab.fr_delay = c + (K.fr_delay - K.ctab);
#define MRX_DTAB(f) L.tab.f = (const double*)(c + ((const int32_t*)K.f - K.ctab))
  if (!KD(pregen)) { MRX_DTAB(src_base); MRX_DTAB(src_noise); }  // with the order table they stay in global memory
  MRX_DTAB(er_base); MRX_DTAB(er_noise); MRX_DTAB(fr_base); MRX_DTAB(fr_noise);
#undef MRX_DTAB
}

MRX_DEV uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680U;
  y ^= (y << 15) & 0xefc60000U;
  y ^= (y >> 18);
  return y;
}

// In-place twist.  new[k] = x[(k+397)%624] ^ f(x[k], x[(k+1)%624]); processed 208 words per round (three rounds),
// reads-then-writes, which is dependency-safe because the recurrence distance (227) exceeds 208: a round reads old words at
// k, k + 1 and (k < 227) k + 397, and new words at k - 227 that an EARLIER round wrote.  (Rounds of 64 words were ten dependent
// LDS round trips per twist; a buffer-stream twist happens every second or third tick of global_trade.22p.)
MRX_DEV void mt_twist(uint32_t* mt) {
  const int l = wave::lane();
  static_assert(MT_WORDS == 3 * 208, "round size");
  for (int k0 = 0; k0 < MT_WORDS; k0 += 208) {
    uint32_t x[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool on = j * 64 + l < 208;
      const int k = on ? k0 + j * 64 + l : k0;  // (idle lanes of the last quarter read a valid word)
      const int k1 = (k + 1 == MT_WORDS) ? 0 : k + 1;
      const int km = (k + 397 >= MT_WORDS) ? k + 397 - MT_WORDS : k + 397;
      const uint32_t y = (mt[k] & 0x80000000U) | (mt[k1] & 0x7fffffffU);
      x[j] = mt[km] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
    }
    wave::sync();
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (j * 64 + l < 208) mt[k0 + j * 64 + l] = x[j];
    wave::sync();
  }
}

// The lane holding `rank` (0 <= rank < n) receives the rank-th next random() of the stream;
// lanes with rank < 0 receive 0.  n is wave-uniform (<= 64).  Advances the stream by n draws.
// `twisted` is set when the state array was regenerated (only then does it have to be written back; a draw without a
// twist only moves the cursor, which lives in the private header).
MRX_DEV double mt_draw_batch(uint32_t* mt, int& idx, int rank, int n, bool& twisted) {
  uint32_t a = 0, b = 0;
  const bool has = rank >= 0;
  const int w0 = idx + 2 * rank, w1 = w0 + 1;
  if (has && w0 < MT_WORDS) a = mt_temper(mt[w0]);
  if (has && w1 < MT_WORDS) b = mt_temper(mt[w1]);
  if (idx + 2 * n > MT_WORDS) {  // wave-uniform
    wave::sync();
    mt_twist(mt);
    twisted = true;
    if (has && w0 >= MT_WORDS) a = mt_temper(mt[w0 - MT_WORDS]);
    if (has && w1 >= MT_WORDS) b = mt_temper(mt[w1 - MT_WORDS]);
    idx = idx + 2 * n - MT_WORDS;
  } else {
    idx += 2 * n;
  }
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// The same for up to NB draws per lane in ONE LDS round trip: lane's b-th draw has rank[b] (-1: none) among the n draws this call
// hands out (n wave-uniform, 2 n < MT_WORDS).  Stream order == rank order, exactly as NB successive mt_draw_batch calls.
template <int NB>
MRX_DEV void mt_draw_multi(uint32_t* mt, int& idx, const int (&rank)[NB], int n, bool& twisted, double (&r)[NB]) {
  uint32_t a[NB], b[NB];
#pragma unroll
  for (int i = 0; i < NB; i++) {  // branch-free: clamped addresses, masked values (all reads in flight together)
    const int w0 = idx + 2 * rank[i], w1 = w0 + 1;
    const uint32_t ra = mt[(rank[i] >= 0 && w0 < MT_WORDS) ? w0 : 0], rb = mt[(rank[i] >= 0 && w1 < MT_WORDS) ? w1 : 0];
    a[i] = (rank[i] >= 0 && w0 < MT_WORDS) ? mt_temper(ra) : 0u;
    b[i] = (rank[i] >= 0 && w1 < MT_WORDS) ? mt_temper(rb) : 0u;
  }
  if (idx + 2 * n > MT_WORDS) {  // wave-uniform
    wave::sync();
    mt_twist(mt);
    twisted = true;
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int w0 = idx + 2 * rank[i], w1 = w0 + 1;
      const uint32_t ra = mt[(rank[i] >= 0 && w0 >= MT_WORDS) ? w0 - MT_WORDS : 0], rb = mt[(rank[i] >= 0 && w1 >= MT_WORDS) ? w1 - MT_WORDS : 0];
      if (rank[i] >= 0 && w0 >= MT_WORDS) a[i] = mt_temper(ra);
      if (rank[i] >= 0 && w1 >= MT_WORDS) b[i] = mt_temper(rb);
    }
    idx = idx + 2 * n - MT_WORDS;
  } else {
    idx += 2 * n;
  }
#pragma unroll
  for (int i = 0; i < NB; i++) r[i] = ((double)(a[i] >> 5) * 67108864.0 + (double)(b[i] >> 6)) * (1.0 / 9007199254740992.0);
}

// one random() delivered to every lane (wave-uniform control flow)
MRX_DEV double mt_draw_uniform(uint32_t* mt, int& idx) { bool tw = false; return mt_draw_batch(mt, idx, 0, 1, tw); }

// one raw 32-bit output delivered to every lane
MRX_DEV uint32_t mt_next_u32(uint32_t* mt, int& idx) {
  if (idx >= MT_WORDS) {
    wave::sync();
    mt_twist(mt);
    idx = 0;
  }
  return mt_temper(mt[idx++]);
}

// random.seed(int): init_by_array over the 32-bit digits of |seed|.  Serial; run by ONE lane
// (several lanes may seed different arrays concurrently).
MRX_DEV void mt_seed_serial(uint32_t* mt, long long seed) {
  unsigned long long a = seed < 0 ? (unsigned long long)(-(seed + 1)) + 1ull : (unsigned long long)seed;
  uint32_t key[2] = {(uint32_t)(a & 0xffffffffull), (uint32_t)(a >> 32)};
  const int key_length = key[1] ? 2 : 1;
  mt[0] = 19650218U;
  for (int i = 1; i < MT_WORDS; i++) mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  int i = 1, j = 0;
  for (int k = MT_WORDS; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
    i++; j++;
    if (i >= MT_WORDS) { mt[0] = mt[MT_WORDS - 1]; i = 1; }
    if (j >= key_length) j = 0;
  }
  for (int k = MT_WORDS - 1; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
    i++;
    if (i >= MT_WORDS) { mt[0] = mt[MT_WORDS - 1]; i = 1; }
  }
  mt[0] = 0x80000000U;
}

// value + rand.uniform(-noise, noise)  with uniform(a,b) = a + (b-a)*random()
// (data_lib/cim/utils.py:30-41, CPython Lib/random.py).  Compiled with -ffp-contract=off.
MRX_DEV double apply_noise(double value, double noise, double r) {
  const double a = -noise;
  return value + (a + (noise - a) * r);
}

// ------------------------------------------------------------------------------------------
// coalesced wave copies (16 B per lane)
// n_words is a multiple of 4 for every row the engine copies (cim_plan rounds FW / PW; MT_WORDS = 624)
MRX_DEV void copy_words(int32_t* dst, const int32_t* src, int n_words) {
  const int l = wave::lane();
  const int n4 = n_words >> 2;
  const int4* s4 = (const int4*)src;
  int4* d4 = (int4*)dst;
#if defined(__HIP_DEVICE_COMPILE__)
  // LDS -> HBM write-back of frame / private state / snapshot: non-temporal stores.  The rows are next read one env-step
  // later (tens of µs, usually by another CU), so letting them allocate in L2 only evicts the topology tables and order rows
  // the other waves are using: measured +6.5 % env-steps/s (the matching `nt` on the LDS-DMA loads changes nothing).
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  // (deliberately NOT fully unrolled: a burst of all the row's stores at once measured 10 % slower than read -> store -> read ...)
#ifndef MRX_COPY_UNROLL
#define MRX_COPY_UNROLL 1
#endif
#pragma unroll MRX_COPY_UNROLL
  for (int i = l; i < n4; i += 64) __builtin_nontemporal_store(((const v4i_*)s4)[i], (v4i_*)d4 + i);
#else
  for (int i = l; i < n4; i += 64) d4[i] = s4[i];
#endif
}

// HBM -> LDS through LDS-DMA (all chunks in flight at once); rows are multiples of 4 words.
// Complete with wave::lds_dma_wait().
MRX_DEV void copy_in_async(int32_t* lds_dst, const int32_t* gsrc, int n_words) {
  const int l = wave::lane();
  const int n4 = n_words >> 2;
  for (int c = 0; c < n4; c += 64)
    if (c + l < n4) wave::lds_dma_16(lds_dst + c * 4, gsrc + (size_t)(c + l) * 4);
}

MRX_DEV int stop_arrival(uint32_t s) { return (int)(s >> 8); }
MRX_DEV int stop_parking(uint32_t s) { return (int)(s & 0xffu); }

// start_tick > 0 (core.py:46: "usually used for pre-processed data streaming"): the reference's event buffer only ever
// executes ticks >= start_tick, so the VESSEL_DEPARTURE events that business_engine.py:371-379 scheduled at earlier leave
// ticks are never run, while every later one still fires ON SCHEDULE, whatever the vessel's state (:634-656 just does
// next_loc_idx += 1).  A vessel whose first departure fell before start_tick therefore never arrives anywhere again
// (stops[next_loc_idx].arrival_tick is always in the past, :151-160): it keeps "departing" at the leave ticks of its
// later stops.  Reproduced literally: those vessels are flagged in the private header, V_NEXT holds the schedule index
// of their next departure event.
MRX_DEV uint64_t zombie_mask(const Lds& L) {
  return ((uint64_t)(uint32_t)L.priv[PH_ZOMBIE_HI] << 32) | (uint32_t)L.priv[PH_ZOMBIE_LO];
}

// ------------------------------------------------------------------------------------------
// vessel_plans of a vessel that is at route position `pos` since tick `arrival` (vessel_sailing_plan_wrapper.py:24-28 over the
// noise-free legs of vessel_future_stops_prediction.py:49-85); the predicted stops themselves are derived at query time
MRX_DEV void write_plans(const CimParams& K, Lds& L, int v, int pos, int arrival) {
  const Tabs& T = L.tab;
  const int Lr = T.v_route_len[v], rb = T.v_route_base[v], lo = T.leg_off[v];
  int tick = arrival, x = pos;  // x walks the route cyclically
  for (int i = 0; i < Lr; i++) {
    tick += T.leg_time[lo + x];
    x = (x + 1 == Lr) ? 0 : x + 1;
    PLANC(v, T.route_cidx[rb + x]) = tick;  // (later stops overwrite earlier)
  }
}


// Everything a tick needs from HBM/L2, requested one memory round trip ahead of its use (a wave issues in
// order, so un-hidden loads stall the serial sections):
//  * per-pair fp64 noise tables and source-port ids for the first 3 x 64 pairs (lane k holds pair k0+lane),
//  * the tick's order count,
//  * for the vessels that will arrive in that tick (known before the tick starts: a vessel departing in tick t
//    cannot arrive in tick t): their discharge records (lane j = j-th candidate load stop) and stop-table
//    entries (lane a = a-th arriving vessel).
struct TickPf {
  double tb[3], tn[3];
  int src[3];
  int q[4], key[4];
  int kk[4];  // order pair (arrival port -> port of the i-th next stop) of the a-th arriving vessel, lane i; -1: none
  int pa;     // lean builds: lane a holds the a-th arriving vessel's port | its compact matrix column << 8
  int ns, otg;
  int oqr[3];  // order table: quantities of pairs lane + 64 b of the coming tick
  uint32_t stk, stk1;
  uint64_t arr_mask;
};

// All prefetch loads are branch-free (addresses clamped to valid memory, validity re-derived at the point of
// use), so the compiler has no control-flow merge that would force an early s_waitcnt.
// consume every prefetched register in one straight-line place (see wave::touch)
MRX_DEV void tick_prefetch_land(TickPf& pf) {
#pragma unroll
  for (int b = 0; b < 3; b++) { wave::touch(pf.tb[b]); wave::touch(pf.tn[b]); wave::touch(pf.src[b]); }
#pragma unroll
  for (int a = 0; a < 4; a++) { wave::touch(pf.q[a]); wave::touch(pf.key[a]); wave::touch(pf.kk[a]); }
#ifdef MRX_LEAN2
  wave::touch(pf.pa);
#endif
  wave::touch(pf.ns); wave::touch(pf.otg); wave::touch(pf.stk); wave::touch(pf.stk1);
#pragma unroll
  for (int b = 0; b < 3; b++) wave::touch(pf.oqr[b]);
}

// static per-pair tables of pairs lane + 64 b: source port (phases B2/B3) and, for the order generator, the
// target ratios
MRX_DEV void tick_prefetch_static(const CimParams& K, TickPf& pf, bool generator) {
  const int lane = wave::lane();
  const int last = KD(NT) > 0 ? KD(NT) - 1 : 0;
#pragma unroll
  for (int b = 0; b < 3; b++) {
    int k = b * 64 + lane;
    k = k < last ? k : last;
    pf.src[b] = K.pair_src[k];
    if (generator) { pf.tb[b] = K.tgt_base[k]; pf.tn[b] = K.tgt_noise[k]; }
  }
}

// stop-table entries, discharge records and order pairs of the first (up to) four vessels of `mask`
MRX_DEV void tick_prefetch_arrivals(const CimParams& K, int env, Lds& L, uint64_t mask, TickPf& pf) {
  const int lane = wave::lane();
  const int V = KD(V), P = KD(P);
  const Tabs& T = L.tab;
  const int32_t* g_rec = K.rec + (size_t)env * KD(REC_W);
  // per-vessel words as rows (lane = vessel): one LDS trip for all of them, then register reads per arriving vessel
  const int lv = lane < V ? lane : 0;
  const int r_next = FV(VA_NEXT_LOC_IDX, lv), r_krl = V_KRL(lv), r_pos = V_POS(lv);
  const int r_len = T.v_route_len[lv], r_rb = T.v_route_base[lv], r_rec = T.rec_off[lv];
  // lane a (< 4) fetches the stop-table entries of the a-th arriving vessel
  {
    uint64_t m = mask;
    int v = 0;
    for (int a = 0; a < 4; a++) {
      const int va = m ? __builtin_ctzll(m) : 0;
      if (m) m &= m - 1;
      if (lane == a) v = va;
    }
    const int k = wave::shfl(r_next, lane < 4 ? v : 0);
    const size_t srow = ((size_t)env * V + v) * KD(SMAX);
    pf.ns = K.nstops[(size_t)env * V + v];
    pf.stk = K.stops[srow + (k < KD(SMAX) ? k : 0)];
    pf.stk1 = K.stops[srow + (k + 1 < KD(SMAX) ? k + 1 : 0)];
  }
  // lane j fetches the j-th candidate discharge record (and its load tick) of each of those vessels, and the order
  // pair that ships from the arrival port to the port of the vessel's j-th next stop
  uint64_t m = mask;
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int v = m ? __builtin_ctzll(m) : 0;  // wave-uniform; vessel 0 is a harmless stand-in when fewer arrive
    if (m) m &= m - 1;
    const int k = wave::readlane(r_next, v), Lr = wave::readlane(r_len, v), krl = wave::readlane(r_krl, v);
    const int pos = wave::readlane(r_pos, v), rb = wave::readlane(r_rb, v);
    const int RL = Lr + 1;
    const int sidx = k - Lr + lane;
    int col = krl + 1 + lane;
    if (col >= RL) col -= RL;
    const bool ok = lane < Lr && sidx >= 0;
    pf.q[a] = g_rec[wave::readlane(r_rec, v) + krl * RL + (ok ? col : 0)];
    pf.key[a] = (int)K.stops[((size_t)env * V + v) * KD(SMAX) + (ok ? sidx : 0)];
    int xn = pos + 1 + (lane < Lr ? lane : 0);  // < 2 Lr: one conditional subtraction instead of an integer modulo
    if (xn >= Lr) xn -= Lr;
    pf.kk[a] = K.pair_dense[(int)T.route_port[rb + pos] * P + (int)T.route_port[rb + xn]];
  }
}

// element `i` of row `r` (tick index) of env's order table: uint16 or int32 elements (CimParams::order_half)
MRX_DEV int order_cell(const CimParams& K, int env, int r, int i) {
  const size_t e = (size_t)env * (size_t)K.orders_stride + (size_t)r * KD(NTP) + i;
  return KD(order_half) ? (int)((const uint16_t*)K.orders)[e] : K.orders[e];
}

// PG = the episode's orders were drawn by mrx_cim_reset (CimParams::pregen): the tick's row of the order table
// replaces the order count and the generator's inputs.
template <bool PG>
MRX_DEV void tick_prefetch(const CimParams& K, int env, Lds& L, int t, TickPf& pf) {
  const int lane = wave::lane();
  if constexpr (PG) {
#pragma unroll
    for (int b = 0; b < 3; b++) pf.oqr[b] = order_cell(K, env, t - KD(start_tick), b * 64 + lane < KD(NTP) ? b * 64 + lane : 0);
    pf.otg = 0;
  } else {
    pf.otg = K.order_prop[(size_t)env * KD(T) + t];
  }
  bool arr = lane < KD(V) && !FV(VA_IS_PARKING, lane) && FV(VA_NEXT_LOC_IDX, lane) > 0 && V_EVT(lane) == t;
  if (KD(start_tick) > 0) arr = arr && !((zombie_mask(L) >> lane) & 1ull);  // (their V_EVT is a departure; they never arrive)
  pf.arr_mask = wave::ballot(arr);
  tick_prefetch_arrivals(K, env, L, pf.arr_mask, pf);
}

// ==========================================================================================
// Order generation of one tick (cim_data_container.py:309-398): `otg` orders -> L.oq[pair].  Used by the step kernel
// (online generation) and by the reset kernel (order table).  pf.tb / tn / src hold the target tables of pairs
// lane + 64 b (tick_prefetch_static).
MRX_DEV void gen_orders(const CimParams& K, Lds& L, long long otg, int& idx_ord, const TickPf& pf, bool& ord_twisted) {
  const int lane = wave::lane();
  const int P = KD(P), NT = KD(NT);
  const Tabs& T = L.tab;
  for (int k = lane; k < NT; k += 64) L.oq[k] = 0;
  double ns = 0.0;
  if (KD(use_order_rng)) {
    const double r = mt_draw_batch(L.mt_ord, idx_ord, lane < P ? lane : -1, P, ord_twisted);
    if (lane < P) ns = apply_noise(T.src_base[lane], T.src_noise[lane], r);
  } else if (lane < P) {
    ns = T.src_base[lane] + 0.0;
  }
  if (lane < P) L.dsrc[lane] = ns;
  wave::sync();
  // list_sum_normalize (utils.py:44-56): left-to-right fp64 sum, then one division per port (lane-parallel)
  // (lane p holds term p: the terms are broadcast out of registers — no LDS round trip per term — and added in list order)
  double tot = 0.0;
  {
    union { double d; int w[2]; } u;
    u.d = ns;
    for (int p = 0; p < P; p++) {
      union { double d; int w[2]; } b;
      b.w[0] = wave::bcast(u.w[0], p); b.w[1] = wave::bcast(u.w[1], p);
      tot += b.d;
    }
  }
  long long c = 0;
  if (lane < P) {
    const double ratio = (tot == 0.0) ? ns : ns / tot;
    c = (long long)ceil((double)otg * ratio);
    if (c > 0x7fffffffll) c = 0x7fffffffll;
    if (c < -0x7fffffffll) c = -0x7fffffffll;
  }
  // sequential split with early break (:354-375): n_p = min(c_p, remaining), remaining -= n_p, stop at remaining == 0
  int brk = P;
  const uint64_t negm = wave::ballot(lane < P && c < 0);
  if (!negm) {  // usual case: a clamped prefix sum
    const long long incl = (long long)wave::scan_incl_add((int)(lane < P ? (c < otg ? c : otg) : 0));  // terms capped at otg: no overflow
    const long long rem_top = otg - (incl - (c < otg ? c : otg));  // remaining orders when port `lane` is reached
    const uint64_t zm = wave::ballot(lane < P && rem_top <= 0);
    if (zm) brk = __builtin_ctzll(zm);
    if (lane < P) L.srcn[lane] = (int32_t)(rem_top <= 0 ? 0 : (c < rem_top ? c : rem_top));
  } else {  // a negative noised ratio makes `remaining` grow: replay the reference loop literally
    if (lane < P) L.srcn[lane] = (int32_t)c;
    wave::sync();
    long long remaining = otg;
    for (int p = 0; p < P; p++) {
      if (remaining == 0) { brk = p; break; }
      long long cp = (long long)U(L.srcn[p]);
      if (cp > remaining) cp = remaining;
      remaining -= cp;
      if (lane == 0) L.srcn[p] = (int32_t)cp;
    }
  }
  const int NTb = T.tgt_off[brk];
#define MRX_TGT_BATCH(k0, TB, TN)                                                              \
  {                                                                                          \
    const int k = (k0) + lane;                                                               \
    const int n = (NTb - (k0)) < 64 ? (NTb - (k0)) : 64;                                      \
    if (KD(use_order_rng)) {                                                                   \
      const double r = mt_draw_batch(L.mt_ord, idx_ord, k < NTb ? lane : -1, n, ord_twisted);  \
      if (k < NTb) L.dtgt[k] = apply_noise(TB, TN, r);                                       \
    } else if (k < NTb) {                                                                    \
      L.dtgt[k] = (TB) + 0.0;                                                                \
    }                                                                                        \
  }
  if (NTb > 0) MRX_TGT_BATCH(0, pf.tb[0], pf.tn[0])
  if (NTb > 64) MRX_TGT_BATCH(64, pf.tb[1], pf.tn[1])
  if (NTb > 128) MRX_TGT_BATCH(128, pf.tb[2], pf.tn[2])
  for (int k0 = 192; k0 < NTb; k0 += 64) MRX_TGT_BATCH(k0, K.tgt_base[k0 + lane < NT ? k0 + lane : 0], K.tgt_noise[k0 + lane < NT ? k0 + lane : 0])
#undef MRX_TGT_BATCH
  wave::sync();
  // per-port normaliser: left-to-right sum of its noised target ratios (:361-366)
  if (lane < brk) {
    const int off = T.tgt_off[lane], cnt = T.tgt_off[lane + 1] - off;
    double ts = 0.0;
    for (int j = 0; j < cnt; j += 4) {  // four terms requested per LDS round trip, added strictly left to right
      const double d0 = L.dtgt[off + j], d1 = L.dtgt[off + (j + 1 < cnt ? j + 1 : j)], d2 = L.dtgt[off + (j + 2 < cnt ? j + 2 : j)],
                   d3 = L.dtgt[off + (j + 3 < cnt ? j + 3 : j)];
      ts += d0;
      if (j + 1 < cnt) ts += d1;
      if (j + 2 < cnt) ts += d2;
      if (j + 3 < cnt) ts += d3;
    }
    L.dsrc[lane] = ts;
  }
  wave::sync();
  // one division per (src, dst) pair, lane-parallel: raw ceil(n_src * ratio) (:380)
#define MRX_PAIR_BATCH(k0, SRC)                                                                     \
  {                                                                                               \
    const int k = (k0) + lane;                                                                    \
    if (k < NTb) {                                                                                \
      const int sp = (SRC);                                                                       \
      const long long n_p = L.srcn[sp];                                                           \
      long long cur = 0;                                                                          \
      if (n_p > 0) {                                                                              \
        const double ts = L.dsrc[sp], x = L.dtgt[k];                                              \
        cur = (long long)ceil((double)n_p * ((ts == 0.0) ? x : x / ts));                          \
        if (cur > 0x7fffffffll) cur = 0x7fffffffll;                                               \
        if (cur < -0x7fffffffll) cur = -0x7fffffffll;                                             \
      }                                                                                           \
      L.oq[k] = (int32_t)cur;                                                                     \
    }                                                                                             \
  }
  if (NTb > 0) MRX_PAIR_BATCH(0, pf.src[0])
  if (NTb > 64) MRX_PAIR_BATCH(64, pf.src[1])
  if (NTb > 128) MRX_PAIR_BATCH(128, pf.src[2])
  for (int k0 = 192; k0 < NTb; k0 += 64) MRX_PAIR_BATCH(k0, K.pair_src[k0 + lane < NT ? k0 + lane : 0])
#undef MRX_PAIR_BATCH
  wave::sync();
  // sequential hand-out per source port (:381-393): cur = min(cur, remaining), remaining -= cur; only positive orders exist.
  // With no negative raw quantity (a negative noised ratio; rare) `remaining` only shrinks, so the hand-out is closed form:
  // cur_j = min(raw_j, max(0, n_p - sum of the port's earlier raw quantities)) — an inclusive prefix over all pairs (the same
  // device as phase B3's), lane-parallel, instead of 22 lanes walking up to 21 targets each.
  uint64_t negq = 0;
  for (int k0 = 0; k0 < NTb; k0 += 64) negq |= wave::ballot(k0 + lane < NTb && L.oq[k0 + lane] < 0);
  if (!negq) {
    uint32_t* pre = (uint32_t*)L.dtgt;  // the noised ratios are dead by now; unsigned: the running total may wrap, a port's share cannot
    uint32_t carry = 0;
#define MRX_HAND_SCAN(k0, SRC)                                                 \
    {                                                                          \
      const int k = (k0) + lane;                                               \
      int v = 0;                                                               \
      if (k < NTb) {                                                           \
        const int n_p = L.srcn[(SRC)];                                         \
        const int q = L.oq[k];                                                 \
        v = q < n_p ? q : (n_p > 0 ? n_p : 0); /* n_p < 2^24: 64 of them fit */ \
      }                                                                        \
      const uint32_t incl = (uint32_t)wave::scan_incl_add(v) + carry;          \
      carry = (uint32_t)wave::bcast((int)incl, 63);                            \
      if (k < NTb) pre[k] = incl;                                              \
    }
    if (NTb > 0) MRX_HAND_SCAN(0, pf.src[0])
    if (NTb > 64) MRX_HAND_SCAN(64, pf.src[1])
    if (NTb > 128) MRX_HAND_SCAN(128, pf.src[2])
    for (int k0 = 192; k0 < NTb; k0 += 64) MRX_HAND_SCAN(k0, K.pair_src[k0 + lane < NT ? k0 + lane : 0])
#undef MRX_HAND_SCAN
    wave::sync();
#define MRX_HAND_OUT(k0, SRC)                                                                   \
    {                                                                                           \
      const int k = (k0) + lane;                                                                \
      if (k < NTb) {                                                                            \
        const int sp = (SRC);                                                                   \
        const int n_p = L.srcn[sp], q = L.oq[k], off = T.tgt_off[sp];                           \
        const int v = q < n_p ? q : (n_p > 0 ? n_p : 0);                                        \
        const int before = (int)(pre[k] - (uint32_t)v - (off > 0 ? pre[off - 1] : 0u));         \
        const int rem = n_p - before;                                                           \
        const int cur = rem <= 0 ? 0 : (q < rem ? q : rem);                                     \
        L.oq[k] = cur > 0 ? cur : 0;                                                            \
      }                                                                                         \
    }
    if (NTb > 0) MRX_HAND_OUT(0, pf.src[0])
    if (NTb > 64) MRX_HAND_OUT(64, pf.src[1])
    if (NTb > 128) MRX_HAND_OUT(128, pf.src[2])
    for (int k0 = 192; k0 < NTb; k0 += 64) MRX_HAND_OUT(k0, K.pair_src[k0 + lane < NT ? k0 + lane : 0])
#undef MRX_HAND_OUT
  } else if (lane < brk) {
    const long long n_p = L.srcn[lane];
    if (n_p > 0) {
      const int off = T.tgt_off[lane], cnt = T.tgt_off[lane + 1] - off;
      long long rem = n_p;
      for (int j = 0; j < cnt; j++) {
        long long cur = L.oq[off + j];
        if (cur > rem) cur = rem;
        rem -= cur;
        L.oq[off + j] = cur > 0 ? (int32_t)cur : 0;
      }
    }
  }
}

// ==========================================================================================
// RESET: Env.reset / set_seed + data generation + frame initialisation, all on the device
// (core.py:143-170; cim/business_engine.py:226-242, 321-398; cim_data_container_helpers.py:56-73;
//  cim_data_generator.py:18-205; parsers.py:57-106; sim_random.py:35-63)
MRX_DEV void reset_env(const CimParams& K, int env, int32_t* lds, long long cmd) {
  Lds L = make_lds(K, lds);
  uint32_t* mt_route = (uint32_t*)(lds + KD(l_mt2));
  uint32_t* mt_oinit = (uint32_t*)(lds + KD(l_mt3));
  L.mt_ord = (uint32_t*)(lds + KD(r_mt0));   // the reset kernel's own layout: four streams side by side; frame and private
  L.mt_buf = (uint32_t*)(lds + KD(r_mt1));   // head are initialised in the same words once the streams are persisted
  const int lane = wave::lane();
  const int P = KD(P), V = KD(V), TT = KD(T);
  uint32_t* g_mt = K.mt + (size_t)env * MTS_COUNT * MT_WORDS;
  int32_t* g_priv = K.priv + (size_t)env * KD(PW);
  int32_t* g_live = K.live + (size_t)env * KD(FW);
  const Tabs& T = L.tab;

  // ---- base seed
  long long base;
  if (KD(data_mode)) {
    // dump folder / real data files: every (re)load re-seeds the registry with the data set's own seed
    // (cim_data_container_helpers.py:79-85, 118-123), whatever set_seed / the redraw say
    base = K.data_seed;
  } else if (cmd >= 0) {
    base = cmd;
  } else if (cmd == -1) {
    base = K.seed[env];
  } else {
    // reset(keep_seed=False): seed = random[ROUTE_INIT].randint(0, 4095) drawn from the stream as the
    // previous data generation left it (cim_data_container_helpers.py:58-60)
    int idx = g_priv[PH_IDX_ROUTE];
    copy_words((int32_t*)mt_route, (const int32_t*)(g_mt + MTS_ROUTE * MT_WORDS), MT_WORDS);
    wave::sync();
    uint32_t r = mt_next_u32(mt_route, idx) >> (32 - 13);  // 4096.bit_length() == 13
    while (r >= 4096u) r = mt_next_u32(mt_route, idx) >> (32 - 13);
    base = (long long)r;
    wave::sync();
  }

  // ---- seed the streams: stream i of the registry gets base + i (sim_random.py:35-63)
  {
    uint32_t* arr = lane == 0 ? L.mt_ord : lane == 1 ? L.mt_buf : lane == 2 ? mt_route : mt_oinit;
    const int sidx = lane == 0 ? KD(idx_order_num) : lane == 1 ? KD(idx_buffer) : lane == 2 ? KD(idx_route) : KD(idx_order_init);
    if (lane < 3 || (lane == 3 && KD(has_order_init))) mt_seed_serial(arr, base + sidx);  // 4 lanes, 4 streams
  }
  wave::sync();
  int idx_route = MT_WORDS, idx_oi = MT_WORDS;

  // ---- order proportion (parsers.py:57-106)
  int32_t* g_prop = K.order_prop + (size_t)env * TT;
  int status = 0;
  uint32_t* g_stops = K.stops + (size_t)env * V * KD(SMAX);
  if (KD(data_mode)) {
    // stop tables, vessel periods and the order proportion come from files (cim_data_loader.py:360-450)
    for (int t = lane; t < TT; t += 64) g_prop[t] = (KD(data_mode) == 1 && t < KD(data_T)) ? K.fx_order_prop[t] : 0;
    for (int i = lane; i < V * KD(SMAX); i += 64) g_stops[i] = K.fx_stops[i];
    if (lane < V) { K.nstops[(size_t)env * V + lane] = K.fx_nstops[lane]; K.vperiod[(size_t)env * V + lane] = K.fx_vperiod[lane]; }
    // the frame initialisation below reads the stop table back through global memory
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    wave::sync();
  } else {
  for (int t0 = 0; t0 < TT; t0 += 64) {
    const int t = t0 + lane;
    double orders = t < TT ? K.order_dist[t % KD(period)] : 0.0;
    const bool nz = t < TT && orders != 0.0;
    if (KD(has_order_init)) {
      const uint64_t m = wave::ballot(nz);
      const int rank = nz ? __builtin_popcountll(m & ((1ull << lane) - 1ull)) : -1;
      bool tw = false;
      const double r = mt_draw_batch(mt_oinit, idx_oi, rank, __builtin_popcountll(m), tw);
      if (nz) orders = apply_noise(orders, K.sample_noise, r);
    }
    if (t < TT) {
      int32_t val = 0;
      if (nz) {
        double c = orders < 1.0 ? orders : 1.0;
        if (c < 0.0) c = 0.0;
        val = (int32_t)floor(c * (double)KD(total_containers));
      }
      g_prop[t] = val;
    }
  }

  // ---- route unrolling (cim_data_generator.py:18-115): one sequential stream across vessels, two draws per stop (parking
  // duration, then sailing speed), stops until future_stop_number + 1 of them lie past max_tick.  LANE-PARALLEL: a batch hands
  // the stream's next draws out two per lane — lane j is the j-th stop of the batch — so parking and sailing times of up to 64
  // stops are computed at once; arrival ticks are a prefix sum, "how many stops does the reference run" a ballot over
  // "arrives past max_tick" (arrival ticks only grow).  The wave-uniform loop of ~5000 sequential stops per env (~70 dependent
  // instructions each) was most of the reset kernel's time.  As before a batch never reaches past the current state block (draws
  // that stay unused are given back, and nothing can be given back across a regeneration); a stop whose two draws fall into
  // different batches carries its parking duration over.
  for (int v = 0; v < V; v++) {
    const int Lr = K.v_route_len[v], rb = K.v_route_base[v], start = K.v_start[v];
    const double speed = K.v_speed[v], sn = K.v_speed_noise[v], dur = K.v_dur[v], dn = K.v_dur_noise[v];
    // lane l holds leg l of the vessel's route (route length <= 62): distance and noise-free leg time
    union { double d; int w[2]; } my_dist;
    my_dist.d = lane < Lr ? K.route_dist[rb + lane] : 0.0;
    const int my_leg = lane < Lr ? K.leg_time[K.leg_off[v] + lane] : 0;
    int tick = 0, extra = 0, k = 0;          // as in the reference loop, at the start of the batch
    bool have_park = false, done = false;    // a parking duration drawn by the previous batch waits for its speed draw
    int park_carry = 0;
    while (!done) {  // wave-uniform
      const int left = (MT_WORDS - idx_route) / 2;
      // draws of this batch (>= 1).  (One word left in the block: the next draw straddles the regeneration anyway, and the
      // cursor then stands far enough into the new block for anything that is given back.)
      const int nd = (idx_route >= MT_WORDS - 1 || left > 126) ? 126 : left;
      const int c = have_park ? 1 : 0;
      // lane j: stop j of the batch.  Its parking draw is draw 2 j - c of the batch (stop 0 of a batch with a carried
      // parking duration has none), its speed draw is draw 2 j - c + 1.
      const int pi = 2 * lane - c, si = pi + 1;
      const int rank[2] = {(pi >= 0 && pi < nd) ? pi : -1, si < nd ? si : -1};
      double r[2];
      bool tw = false;
      mt_draw_multi<2>(mt_route, idx_route, rank, nd, tw, r);
      const int n_full = (nd + c) / 2;                   // stops of the batch with both draws
      const bool half = ((nd + c) & 1) != 0;             // ... and one more parking draw whose speed draw is in the next batch
      int parking = (int)ceil(apply_noise(dur, dn, r[0]));
      if (c && lane == 0) parking = park_carry;
      const double noised_speed = apply_noise(speed, sn, r[1]);
      int loc = start + k + lane;                        // route position of stop k + lane
      loc = loc % Lr;
      union { double d; int w[2]; } dist;
      dist.w[0] = wave::shfl(my_dist.w[0], loc); dist.w[1] = wave::shfl(my_dist.w[1], loc);
      const bool full = lane < n_full;
      const int sailing = full ? (int)ceil(dist.d / noised_speed) : 0;
      const int delta = full ? parking + sailing : 0;
      const int incl = wave::scan_incl_add(delta);
      const int t_before = tick + incl - delta, t_after = tick + incl;
      // the reference runs stop j while extra <= future_n BEFORE it; extra counts the stops (so far) that end past max_tick
      const uint64_t over = wave::ballot(full && t_after > TT);
      const int extra_before = extra + __builtin_popcountll(over & ((1ull << lane) - 1ull));
      bool run = full && extra_before <= KD(future_n);
      int n_run = __builtin_popcountll(wave::ballot(run));
      if (k + n_run >= 4 * KD(SMAX) + 1) {  // the reference has no such limit; a table this long is an error either way
        n_run = 4 * KD(SMAX) + 1 - k;
        run = run && lane < n_run;
        status |= 2;
        done = true;
      }
      if (wave::ballot(run && (parking <= 0 || parking > 255))) status |= 8;  // reference: assert parking_duration > 0
      if (wave::ballot(run && k + lane >= KD(SMAX))) status |= 2;            // MRX_ENV_STOP_OVERFLOW
      if (run && k + lane < KD(SMAX)) g_stops[(size_t)v * KD(SMAX) + k + lane] = ((uint32_t)t_before << 8) | (uint32_t)(parking & 0xff);
      // the loop is over once future_n + 1 of the stops run so far end past max_tick — possibly right after the batch's last full
      // stop, in which case the half stop's parking draw was never drawn
      if (extra + __builtin_popcountll(over & ((1ull << n_run) - 1ull)) > KD(future_n)) done = true;
      if (done) {
        idx_route -= 2 * (nd - (2 * n_run - c));  // the draws nobody asked for
      } else {
        have_park = half;
        park_carry = wave::bcast(parking, n_full < 64 ? n_full : 0);  // (lane n_full holds the half stop's parking duration)
        tick += wave::bcast(incl, 63);
        extra += __builtin_popcountll(over);
      }
      k += n_run;
    }
    // vessel_period_without_noise (:93-101): the noise-free leg times of the first route_length stops
    const int n_per = k < Lr ? k : Lr;
    int ploc = start + lane;
    ploc = ploc % Lr;
    const int leg_l = wave::shfl(my_leg, ploc);
    const int period = wave::bcast(wave::scan_incl_add(lane < n_per ? leg_l : 0), 63);
    if (lane == 0) { K.nstops[(size_t)env * V + v] = k < KD(SMAX) ? k : KD(SMAX); K.vperiod[(size_t)env * V + v] = period; }
  }
  __atomic_thread_fence(__ATOMIC_SEQ_CST);  // (the frame initialisation below reads the stop table back, other lanes than wrote it)
  }  // generated data
  wave::sync();

  // ---- persist RNG streams
  copy_words((int32_t*)(g_mt + MTS_ORDER * MT_WORDS), (const int32_t*)L.mt_ord, MT_WORDS);
  copy_words((int32_t*)(g_mt + MTS_BUFFER * MT_WORDS), (const int32_t*)L.mt_buf, MT_WORDS);
  copy_words((int32_t*)(g_mt + MTS_ROUTE * MT_WORDS), (const int32_t*)mt_route, MT_WORDS);
  wave::sync();  // (all four streams alias the frame / private-head region initialised next: cim_layout.h)

  // ---- frame (business_engine.py:321-356, 381-398) and private state
  for (int i = lane; i < KD(FW); i += 64) L.frame[i] = (i >= KD(f_plans) && i < KD(f_plans) + KD(NC)) ? -1 : 0;
  for (int i = lane; i < KD(PWH); i += 64) L.priv[i] = 0;
  wave::sync();
  if (lane < P) {
    FP(PA_CAPACITY, lane) = K.p_cap[lane];
    FP(PA_EMPTY, lane) = K.p_init_empty[lane];
  }
  if (lane < V) {
    const int v = lane;
    FV(VA_CAPACITY, v) = K.v_cap[v];
    FV(VA_ROUTE_IDX, v) = K.v_route[v];
    FV(VA_EMPTY, v) = K.v_init_empty[v];
    FV(VA_REMAINING_SPACE, v) = K.v_total_space[v] - K.v_init_empty[v];
    FV(VA_IS_PARKING, v) = 1;
    FV(VA_LOC_PORT_IDX, v) = K.route_port[K.v_route_base[v] + K.v_start[v]];
    write_plans(K, L, v, K.v_start[v], 0);
    V_EVT(v) = stop_parking(g_stops[(size_t)v * KD(SMAX)]);  // leave tick of stop 0 (arrival 0)
    V_POS(v) = K.v_start[v];
    V_KRL(v) = 0;
    V_PERIOD(v) = K.vperiod[(size_t)env * V + v];
    V_NEXT(v) = K.nstops[(size_t)env * V + v] > 1 ? stop_arrival(g_stops[(size_t)v * KD(SMAX) + 1]) : 0x7fffffff;
  }
  bool zombie = false;
  if (KD(start_tick) > 0 && lane < V) {  // (see zombie_mask) the first departure event that is still executed
    const int v = lane, ns = K.nstops[(size_t)env * V + v];
    int ks = 0;
    while (ks < ns && stop_arrival(g_stops[(size_t)v * KD(SMAX) + ks]) + stop_parking(g_stops[(size_t)v * KD(SMAX) + ks]) < KD(start_tick)) ks++;
    if (ks > 0) {
      zombie = true;
      const uint32_t st = g_stops[(size_t)v * KD(SMAX) + (ks < ns ? ks : 0)];
      V_EVT(v) = ks < ns ? stop_arrival(st) + stop_parking(st) : 0x7fffffff;
      V_NEXT(v) = ks;
    }
  }
  const uint64_t zm = wave::ballot(zombie);
  if (lane == 0) {
    L.priv[PH_TICK] = KD(start_tick);
    L.priv[PH_FLAGS] = FL_FRESH;
    L.priv[PH_IDX_ORDER] = MT_WORDS;
    L.priv[PH_IDX_BUFFER] = MT_WORDS;
    L.priv[PH_IDX_ROUTE] = idx_route;
    L.priv[PH_ZOMBIE_LO] = (int32_t)(uint32_t)(zm & 0xffffffffull);
    L.priv[PH_ZOMBIE_HI] = (int32_t)(uint32_t)(zm >> 32);
    K.seed[env] = base;
    K.status[env] = status;
    K.tick[env] = KD(start_tick);
    K.hint[env] = 1;  // the first step of an episode takes the full path
  }
  wave::sync();
  copy_words(g_live, L.frame, KD(FW));
  copy_words(g_priv, L.priv, KD(PWH));
  for (int i = KD(PWH) + lane; i < KD(PW); i += 64) g_priv[i] = 0;  // no pending full returns
  for (int i = lane; i < KD(S); i += 64) K.ring_fi[(size_t)env * KD(S) + i] = -1;
  int32_t* g_rec0 = K.rec + (size_t)env * KD(REC_W);
  for (int i = lane; i < KD(REC_W); i += 64) g_rec0[i] = 0;
}

// ==========================================================================================
// ORDER TABLE, branch-free form (CimParams::order_fast; the plan proves what it needs: cim_layout.h).  Same draws, same fp64
// operations in the same order as gen_orders (cim_data_container.py:309-398), organised so that a tick is ~350 vector
// instructions instead of ~1100 — the generic generator saturates the chip's VALU issue for 30 ms per reset of 16 384 envs:
//  * the stream lives in a TWO-block window (current block + the one after it, 1248 words): a tick's draws (<= 384 words)
//    never straddle a regeneration, so every draw is one unconditional 8-byte LDS read; the next block is generated into the
//    buffer that has just been used up, in four rounds (192 / 192 / 192 / 48 words: the recurrence distance is 227);
//  * the P source draws and the NT target draws of a tick are handed out as ONE run of P + NT draws over three lane batches
//    (the reference stops drawing targets at the port where the orders run out: those ports' quantities are 0 either way and
//    the cursor advances by what the reference would have drawn);
//  * the noised ratios go to LDS in per-list segments padded to four entries with +0.0, and P + 1 lanes add their list
//    left to right (the P target lists and the source list) in lock step, four terms per round trip, no lane-dependent
//    control flow: x + 0.0 == x for every non-negative x;
//  * one reciprocal per PORT instead of one division per PAIR: the fp64 division the compiler emits is
//    y = refine(refine(rcp(d))), q0 = n * y, r = fma(-d, q0, n), q = fma(r, y, q0) for operands that need no scaling
//    (v_div_scale / v_div_fmas / v_div_fixup are the identity on [2^-40, 2^20], which the plan guarantees); y only depends on
//    the denominator, so computing it once per port and finishing each pair with three operations gives the same bits;
//  * all quantities fit int32 (the uint16 table's proof), the row is assembled as uint16 in LDS and leaves in 16-byte stores.
MRX_DEV double of_recip(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(d);
  y = fma(y, fma(-d, y, 1.0), y);
  y = fma(y, fma(-d, y, 1.0), y);
  return y;
#else
  return d;  // (host build of the same source: of_div divides)
#endif
}
MRX_DEV double of_div(double n, double d, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double q0 = n * y;
  return fma(fma(-d, q0, n), y, q0);
#else
  (void)y;
  return n / d;
#endif
}

// next block of the MT19937 stream: nxt[k] = f(cur[k], cur[k + 1], k < 227 ? cur[k + 397] : nxt[k - 227]); cur / nxt are the two
// halves of the window (word offsets 0 / 624 in either order)
MRX_DEV void of_next_block(uint32_t* win, int cur, int nxt) {
  const int l = wave::lane();
#define MRX_OF_ROUND(K0, N)                                                                                  \
  _Pragma("unroll") for (int j = 0; j < ((N) + 63) / 64; j++) {                                              \
    const int k = (K0) + j * 64 + l;                                                                         \
    const bool on = (j + 1) * 64 <= (N) || l < (N) - j * 64;   /* (a whole batch: known at compile time) */      \
    const int kc = on ? k : (K0);                                                                            \
    const uint32_t a = win[cur + kc], b = win[kc + 1 == MT_WORDS ? nxt : cur + kc + 1];                      \
    const uint32_t m = win[kc < 227 ? cur + kc + 397 : nxt + kc - 227];                                      \
    const uint32_t y = (a & 0x80000000U) | (b & 0x7fffffffU);                                                \
    x[j] = m ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1U)) & 0x9908b0dfU);                                    \
  }                                                                                                          \
  wave::sync();                                                                                              \
  _Pragma("unroll") for (int j = 0; j < ((N) + 63) / 64; j++)                                                \
    if ((j + 1) * 64 <= (N) || l < (N) - j * 64) win[nxt + (K0) + j * 64 + l] = x[j];                        \
  wave::sync();
  uint32_t x[3];
  MRX_OF_ROUND(0, 192)
  MRX_OF_ROUND(192, 192)
  MRX_OF_ROUND(384, 192)
  MRX_OF_ROUND(576, 48)
#undef MRX_OF_ROUND
}

MRX_DEV void gen_order_table_fast(const CimParams& K, int env, int32_t* lds) {
  const int lane = wave::lane();
  const int P = KD(P), NT = KD(NT), ND = P + NT;
  uint32_t* win = (uint32_t*)(lds + KD(gf_win));
  double* val = (double*)(lds + KD(gf_val));
  int32_t* rec = lds + KD(gf_rec);
  int32_t* pre1 = lds + KD(gf_pre);
  uint16_t* row16 = (uint16_t*)(lds + KD(gf_row));
  int32_t* seg = lds + KD(gf_seg);
  const int ZERO = KD(gf_slots), TRASH = KD(gf_slots) + 4;
  copy_in_async((int32_t*)win, (const int32_t*)(K.mt + ((size_t)env * MTS_COUNT + MTS_ORDER) * MT_WORDS), MT_WORDS);

  // ---- static per-lane words (registers for the whole episode)
  // summation lanes: lane p < P adds port p's target list, lane P the source list
  const int cnt_p = lane < P ? K.tgt_off[lane + 1] - K.tgt_off[lane] : 0;
  const int pad_p = (cnt_p + 3) & ~3;
  const int Ppad = (P + 3) & ~3;
  const int incl_p = wave::scan_incl_add(pad_p);
  const int sum_seg = lane < P ? Ppad + incl_p - pad_p : 0;
  const int sum_len = lane < P ? pad_p : lane == P ? Ppad : 0;
  const int toff_l = K.tgt_off[lane <= P ? lane : P];  // lane p: first pair of port p; lane P: NT
  int max_len = 4;  // the longest list (padded): the summation loop's trip count, the same for every tick
  for (int len = 8; len <= 64; len += 4)
    if (wave::ballot(sum_len >= len)) max_len = len;
  if (lane < P) seg[lane] = sum_seg;
  for (int i = lane; i < KD(gf_slots) + 6; i += 64) val[i] = 0.0;   // the pads stay +0.0 for good
  for (int i = lane; i < KD(NTP) / 2; i += 64) ((uint32_t*)row16)[i] = 0u;
  if (lane == 0) {
    pre1[0] = 0;
    double* r = (double*)(rec + P * 8);  // the record of the lanes that hold no pair: 0 orders of a list whose sum is 1
    r[0] = 1.0; r[1] = 1.0;
    rec[P * 8 + 4] = 0;
  }
  wave::sync();
  // draw lanes: draw d = 64 b + lane of a tick's run is source port d (d < P) or order pair d - P
  double base[3], negn[3], cm[3];
  int slot[3], recw[3], prew[3], kk[3];
  bool pair[3];
#pragma unroll
  for (int b = 0; b < 3; b++) {
    const int d = b * 64 + lane;
    const bool valid = d < ND, src = d < P;
    const int k = valid && !src ? d - P : 0;
    const int sp = src ? d : K.pair_src[k];
    const double bs = src ? K.src_base[d] : K.tgt_base[k], nz = src ? K.src_noise[d] : K.tgt_noise[k];
    base[b] = bs;
    negn[b] = -nz;
    cm[b] = (nz - negn[b]) * (1.0 / 9007199254740992.0);  // (noise - a) * 2^-53: r = m * 2^-53 exactly, so (noise - a) * r == cm * m
    const int off = K.tgt_off[sp];
    slot[b] = !valid ? TRASH : src ? d : seg[sp] + (k - off);
    recw[b] = (valid && !src ? sp : P) * 8;
    prew[b] = off;
    kk[b] = k;
    pair[b] = valid && !src;
    wave::touch(base[b]); wave::touch(negn[b]); wave::touch(cm[b]); wave::touch(slot[b]); wave::touch(recw[b]); wave::touch(prew[b]);
  }
  const int32_t* g_prop = K.order_prop + (size_t)env * KD(T);
  const int D = KD(T) - KD(start_tick);
  wave::lds_dma_wait();
  // the stream as reset_env seeded it stands at the END of block 0 (cursor 624): block 1 = the first outputs
  of_next_block(win, 0, MT_WORDS);
  int cur = MT_WORDS, idx = 0;
  of_next_block(win, MT_WORDS, 0);
  for (int t0 = 0; t0 < D; t0 += 64) {
    int mine = t0 + lane < D ? g_prop[KD(start_tick) + t0 + lane] : 0;  // 64 ticks of order_proportion per load
    wave::touch(mine);  // wait for it HERE: inside the tick loop the same wait would also drain the previous tick's row stores
    const int n_here = D - t0 < 64 ? D - t0 : 64;
    for (int j = 0; j < n_here; j++) {  // wave-uniform
      const int otg = wave::bcast(mine, j);
      // ---- the tick's P + NT draws, noised ratios to their summation slots
      double x[3];
      const int wb = cur + idx;
#pragma unroll
      for (int b = 0; b < 3; b++) {
        int wi = wb + 2 * (b * 64 + lane);
        wi = wi >= 2 * MT_WORDS ? wi - 2 * MT_WORDS : wi;  // (even, so the two words of a draw never straddle the wrap)
        const uint32_t a = mt_temper(win[wi]), c = mt_temper(win[wi + 1]);
        const double m = (double)(a >> 5) * 67108864.0 + (double)(c >> 6);
        x[b] = base[b] + (negn[b] + cm[b] * m);
        val[slot[b]] = x[b];
      }
      wave::sync();
      // ---- list_sum_normalize's sums (utils.py:44-56), left to right, every list at once
      double acc = 0.0;
      for (int i = 0; i < max_len; i += 4) {
        const bool on = i < sum_len;
        const double* q = val + (on ? sum_seg + i : ZERO);
        const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        acc += q0; acc += q1; acc += q2; acc += q3;
      }
      // ---- sources (:340-375): c_p = ceil(orders * ratio_p), n_p = min(c_p, remaining), stop where remaining hits 0
      double tot;
      {
        union { double d; int w[2]; } u, v;
        u.d = acc;
        v.w[0] = wave::bcast(u.w[0], P); v.w[1] = wave::bcast(u.w[1], P);
        tot = v.d;
      }
      int c = 0;
      if (lane < P) c = (int)ceil((double)otg * (x[0] / tot));
      const int cq = c < otg ? c : otg;
      const int incl = wave::scan_incl_add(lane < P ? cq : 0);
      const int rem_top = otg - (incl - cq);  // remaining orders when port `lane` is reached
      const int n_p = (lane < P && rem_top > 0) ? (c < rem_top ? c : rem_top) : 0;
      const uint64_t zm = wave::ballot(lane < P && rem_top <= 0);
      const int brk = zm ? (int)__builtin_ctzll(zm) : P;
      const int NTb = wave::bcast(toff_l, brk);
      if (lane < P) {
        double* r = (double*)(rec + lane * 8);
        r[0] = acc;
        r[1] = of_recip(acc);
        rec[lane * 8 + 4] = n_p;
      }
      wave::sync();
      // ---- pairs (:376-393): raw = ceil(n_src * ratio), handed out in list order while the port's orders last
      int qv[3], vv[3], np[3], inc[3];
      int carry = 0;
#pragma unroll
      for (int b = 0; b < 3; b++) {
        const double* r = (const double*)(rec + recw[b]);
        const double ts = r[0], y = r[1];
        np[b] = rec[recw[b] + 4];
        const int raw = (int)ceil((double)np[b] * of_div(x[b], ts, y));
        qv[b] = (pair[b] && np[b] > 0) ? raw : 0;
        vv[b] = qv[b] < np[b] ? qv[b] : np[b];
        inc[b] = wave::scan_incl_add(vv[b]) + carry;
        carry = wave::bcast(inc[b], 63);
        if (pair[b]) pre1[kk[b] + 1] = inc[b];
      }
      wave::sync();
#pragma unroll
      for (int b = 0; b < 3; b++) {
        const int before = inc[b] - vv[b] - pre1[prew[b]];
        const int rem = np[b] - before;
        const int cur_q = rem <= 0 ? 0 : (qv[b] < rem ? qv[b] : rem);
        if (pair[b]) row16[kk[b]] = (uint16_t)cur_q;
      }
      wave::sync();
      {
        const size_t e0 = (size_t)env * (size_t)K.orders_stride + (size_t)(t0 + j) * KD(NTP);
        if (lane < KD(NTP) / 8) wave::st16_nt((int32_t*)((uint16_t*)K.orders + e0) + lane * 4, wave::lds_ld16((const int32_t*)row16 + lane * 4));
      }
      // ---- the cursor moves by what the reference drew: P sources + the targets of the ports before the break
      idx += 2 * (P + NTb);
      if (idx >= MT_WORDS) {  // wave-uniform: the current block is used up — the next one becomes current, its successor replaces it
        idx -= MT_WORDS;
        const int old = cur;
        cur = MT_WORDS - cur;
        of_next_block(win, cur, old);
      }
    }
  }
}

// ==========================================================================================
// ORDER TABLE (CimParams::pregen): in `fixed` order mode the orders of tick t are a function of
// order_proportion[t] and the order_number stream alone (cim_data_container.py:309-398), so the whole episode is
// drawn right after reset_env, in tick order, exactly as the reference would while stepping.  Runs as its own
// kernel with a small LDS footprint (RNG state, generator scratch, tables: ~9 KB) so many envs are resident per CU.
MRX_DEV void gen_order_table(const CimParams& K, int env, int32_t* lds) {
  if (KD(order_fast)) { gen_order_table_fast(K, env, lds); return; }
  const int lane = wave::lane();
  Lds L = make_lds(K, lds);
  L.mt_ord = (uint32_t*)(lds + KD(g_mt0));
  L.dsrc = (double*)(lds + KD(g_dsrc));
  L.dtgt = (double*)(lds + KD(g_dtgt));
  L.oq = lds + KD(g_oq);
  L.srcn = lds + KD(g_srcn);
  stage_tables(K, L, lds + KD(g_ctab));
  copy_in_async((int32_t*)L.mt_ord, (const int32_t*)(K.mt + ((size_t)env * MTS_COUNT + MTS_ORDER) * MT_WORDS), MT_WORDS);
  TickPf pf = {};
  tick_prefetch_static(K, pf, true);
  // the source ratio tables into LDS: a global load inside the tick loop would wait (vmcnt) for the previous tick's
  // row stores every time
  {
    double* sb = (double*)(lds + KD(g_srctab));
    if (lane < KD(P)) { sb[lane] = K.src_base[lane]; sb[KD(P) + lane] = K.src_noise[lane]; }
    L.tab.src_base = sb; L.tab.src_noise = sb + KD(P);
  }
  const int32_t* g_prop = K.order_prop + (size_t)env * KD(T);
  const int D = KD(T) - KD(start_tick);
  int idx_ord = MT_WORDS;  // the stream as reset_env seeded it
  wave::lds_dma_wait();
  tick_prefetch_land(pf);  // no load may still be pending inside the tick loop (its wait would also drain the row stores)
  for (int t0 = 0; t0 < D; t0 += 64) {
    int mine = t0 + lane < D ? g_prop[KD(start_tick) + t0 + lane] : 0;  // 64 ticks of order_proportion per load
    wave::touch(mine);  // wait for it HERE: inside the tick loop the same wait would also drain the previous tick's row stores
    const int n_here = D - t0 < 64 ? D - t0 : 64;
    for (int j = 0; j < n_here; j++) {  // wave-uniform
      bool tw = false;
      gen_orders(K, L, (long long)wave::bcast(mine, j), idx_ord, pf, tw);
      wave::sync();
      const size_t e0 = (size_t)env * (size_t)K.orders_stride + (size_t)(t0 + j) * KD(NTP);
      if (KD(order_half)) {  // two uint16 quantities per lane and store (rows are multiples of 8 elements)
        uint32_t* row2 = (uint32_t*)((uint16_t*)K.orders + e0);
        for (int k2 = lane; k2 < KD(NTP) / 2; k2 += 64) {
          const uint32_t lo = 2 * k2 < KD(NT) ? (uint32_t)L.oq[2 * k2] & 0xffffu : 0u, hi = 2 * k2 + 1 < KD(NT) ? (uint32_t)L.oq[2 * k2 + 1] & 0xffffu : 0u;
          row2[k2] = lo | (hi << 16);
        }
      } else {
        for (int k = lane; k < KD(NTP); k += 64) K.orders[e0 + k] = k < KD(NT) ? L.oq[k] : 0;
      }
      wave::sync();
    }
  }
}

// ==========================================================================================
// LEAN builds (plan-specialised, order table, <= 192 pairs): the same tick, written so that every phase starts with ONE LDS
// round trip.  What made the generic tick a chain of ~90 dependent LDS round trips per step (each ~130 cycles with twelve
// resident waves per CU) were table lookups feeding table lookups (pair -> source port -> its first pair -> prefix cell ...),
// read-modify-writes that wait for their read, and per-vessel words re-read in every phase.  Here: static words come out of
// registers (LeanStat), per-vessel words out of registers (VRows), every additive effect is a fire-and-forget ds_add, the
// gathers a phase needs are issued together with clamped addresses (validity re-derived afterwards), the three batches of
// phase B3 share one RNG hand-out (mt_draw_multi), and the discharge records are drawn by RANK instead of being sorted in place.
// Same arithmetic, same RNG stream positions, same results (goldens / fuzz on the emulator and the GPU).
#ifdef MRX_LEAN2
MRX_DEV void lean_stat_load(const CimParams& K, LeanStat& S) {
  const int lane = wave::lane();
  const int32_t* g = K.lean_tab;
#pragma unroll
  for (int b = 0; b < RR_NB; b++) { const int k = b * 64 + lane; S.ps[b] = g[k < KD(NT) ? k : 0]; }
  const int lv = lane < KD(V) ? lane : 0, lp = lane < KD(P) ? lane : 0;
  S.vs0 = g[KD(NTP) + lv]; S.vs1 = g[KD(NTP) + 64 + lv]; S.vs2 = g[KD(NTP) + 128 + lv];
  S.pt = g[KD(NTP) + 256 + lp];
}
MRX_DEV void vrows_load(const CimParams& K, const Lds& L, VRows& W) {
  const int lane = wave::lane();
  const int lv = lane < KD(V) ? lane : 0;
  const int park = FV(VA_IS_PARKING, lv), pos = V_POS(lv), krl = V_KRL(lv), k = FV(VA_NEXT_LOC_IDX, lv);
  W.evt = V_EVT(lv); W.next = V_NEXT(lv);
  vr_set(W, k, pos, krl, park);
}

// tick_prefetch_arrivals without its per-vessel LDS reads and without ds_bpermute: one gather (route ports) feeds the pair lookup
MRX_DEV void tick_prefetch_arrivals_lean(const CimParams& K, int env, Lds& L, uint64_t mask, TickPf& pf, const Lean& Z) {
  const int lane = wave::lane();
  const int V = KD(V), P = KD(P);
  const Tabs& T = L.tab;
  const int32_t* g_rec = K.rec + (size_t)env * KD(REC_W);
  {  // lane a (< 4) fetches the stop-table entries of the a-th arriving vessel
    uint64_t m = mask;
    int v = 0, kq = 0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const int va = m ? __builtin_ctzll(m) : 0;
      if (m) m &= m - 1;
      const int ka = wave::readlane(Z.w.w, va) & 0xffff;
      if (lane == a) { v = va; kq = ka; }
    }
    const size_t srow = ((size_t)env * V + v) * KD(SMAX);
    pf.ns = K.nstops[(size_t)env * V + v];
    pf.stk = K.stops[srow + (kq < KD(SMAX) ? kq : 0)];
    pf.stk1 = K.stops[srow + (kq + 1 < KD(SMAX) ? kq + 1 : 0)];
  }
  uint64_t m = mask;
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int v = m ? __builtin_ctzll(m) : 0;  // wave-uniform; vessel 0 is a harmless stand-in when fewer arrive
    if (m) m &= m - 1;
    const int ww = wave::readlane(Z.w.w, v), k = ww & 0xffff, krl = (ww >> 22) & 63, pos = (ww >> 16) & 63;
    const int vs0 = wave::readlane(Z.s.vs0, v), Lr = vs0 & 63, rb = (int)((unsigned)vs0 >> 12);
    const int reco = (int)((unsigned)wave::readlane(Z.s.vs1, v) >> 16);
    const int RL = Lr + 1;
    const int sidx = k - Lr + lane;
    int col = krl + 1 + lane;
    if (col >= RL) col -= RL;
    const bool ok = lane < Lr && sidx >= 0;
    pf.q[a] = g_rec[reco + krl * RL + (ok ? col : 0)];
    pf.key[a] = (int)K.stops[((size_t)env * V + v) * KD(SMAX) + (ok ? sidx : 0)];
    int xn = pos + 1 + (lane < Lr ? lane : 0);  // < 2 Lr: one conditional subtraction instead of an integer modulo
    if (xn >= Lr) xn -= Lr;
    const int p_arr = T.route_port[rb + pos], c_arr = T.route_cidx[rb + pos];
    pf.kk[a] = K.pair_dense[p_arr * P + (int)T.route_port[rb + xn]];
    if (lane == a) pf.pa = p_arr | (c_arr << 8);
  }
}

MRX_DEV void tick_prefetch_lean(const CimParams& K, int env, Lds& L, int t, TickPf& pf, const Lean& Z) {
  const int lane = wave::lane();
#pragma unroll
  for (int b = 0; b < 3; b++) pf.oqr[b] = order_cell(K, env, t - KD(start_tick), b * 64 + lane < KD(NTP) ? b * 64 + lane : 0);
  pf.otg = 0;
  bool arr = lane < KD(V) && !vr_park(Z.w) && vr_k(Z.w) > 0 && Z.w.evt == t;
  if (KD(start_tick) > 0) arr = arr && !((zombie_mask(L) >> lane) & 1ull);  // (their event is a departure; they never arrive)
  pf.arr_mask = wave::ballot(arr);
  tick_prefetch_arrivals_lean(K, env, L, pf.arr_mask, pf, Z);
}

MRX_DEV uint64_t run_tick_lean(const CimParams& K, int env, Lds& L, int t, TickPf& pf, int& idx_buf, int& status, Prof& prof,
                               bool& buf_twisted, RingRegs& rr, Lean& Z) {
  const int lane = wave::lane();
  const int P = KD(P), V = KD(V), NT = KD(NT), H = KD(H);
  const Tabs& T = L.tab;
  LeanStat& S = Z.s;
  VRows& W = Z.w;
  const uint64_t arr_mask = pf.arr_mask;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  int32_t* g_rec = K.rec + (size_t)env * KD(REC_W);
  int oqr[RR_NB];  // order quantity of pair 64 b + lane (later | (buffer ticks + 1) << 24)
#pragma unroll
  for (int b = 0; b < RR_NB; b++) oqr[b] = (b * 64 + lane < NT) ? pf.oqr[b] : 0;
  const int slot = t % H;
  // the empty returns that fall due now: nothing in this tick adds to ring slot `slot` (a delay of 0 is applied at once, delays
  // are < H), so the row is read before anything else and consumed after the discharges
  const int due_empty = RING_EMPTY(slot, lane < P ? lane : 0);
  prof.mark(PF_ORDER_GEN);

  // ---------------- B1. departures (business_engine.py:634-656): per-vessel words out of registers, written through
  const uint64_t zmask = KD(start_tick) > 0 ? zombie_mask(L) : 0ull;
  if (lane < V) {
    const int v = lane;
    const bool z = (zmask >> v) & 1ull;
    if ((z || vr_park(W)) && W.evt == t) {
      const int Lr = S.vs0 & 63;
      const int k1 = vr_k(W) + 1;
      FV(VA_NEXT_LOC_IDX, v) = k1;
      int x = vr_pos(W) + 1, y = vr_krl(W) + 1;
      x = x == Lr ? 0 : x; y = y == Lr + 1 ? 0 : y;
      V_POS(v) = x; V_KRL(v) = y;
      vr_set(W, k1, x, y, 0);
      FV(VA_IS_PARKING, v) = 0;
      FV(VA_LOC_PORT_IDX, v) = -1;
      if (z) {  // the next scheduled departure event (see zombie_mask)
        const uint32_t* srow = K.stops + ((size_t)env * V + v) * KD(SMAX);
        const int ks = W.next + 1, ns = K.nstops[(size_t)env * V + v];
        W.next = ks;
        V_NEXT(v) = ks;
        const uint32_t st = srow[ks < KD(SMAX) ? ks : KD(SMAX) - 1];
        W.evt = ks < ns ? stop_arrival(st) + stop_parking(st) : 0x7fffffff;
      } else {
        W.evt = W.next;  // arrival tick of the next stop, cached at the previous arrival
      }
      V_EVT(v) = W.evt;
    }
  }

  // ---------------- B2. returns / discharges scheduled for this tick: every effect is an addition (ds_add, no wait)
#pragma unroll
  for (int b = 0; b < RR_NB; b++) {  // RETURN_FULL :499-522, one lane per (src, dst) pair
    const int k = b * 64 + lane;
    int q = 0;
#pragma unroll
    for (int h = 0; h < RR_H; h++) { const bool mine = slot == h; q = mine ? rr.rf[h][b] : q; rr.rf[h][b] = mine ? 0 : rr.rf[h][b]; }
    if (k < NT && q) {
      const int src = S.ps[b] & 0xff;
      wave::lds_add(&FOPK(k), q);
      wave::lds_add(&FP(PA_ON_SHIPPER, src), -q);
      wave::lds_add(&FP(PA_FULL, src), q);
    }
  }
  if (arr_mask) {
    // DISCHARGE_FULL :658-693.  One BUFFER draw per original event, in the order the events were scheduled: (tick of the load,
    // vessel index) — SURVEY.md §9.2.  The records of all arriving vessels are compacted into a list (key, vessel | port << 6 |
    // full_on_vessels cell << 12, quantity); entry i then takes the draw of its RANK in that order — the effects are additions,
    // so only the hand-out of the draws depends on the order, and the list itself is never sorted.
    int32_t* ent = L.misc;
    int n_ent = 0, n_ves = 0, slot4 = 0;
    for (uint64_t m = arr_mask; m; m &= m - 1, slot4++) {  // wave-uniform
      if (slot4 == 4) {  // more than four arrivals in one tick (rare): fetch the next four now
        tick_prefetch_arrivals_lean(K, env, L, m, pf, Z);
        tick_prefetch_land(pf);
        slot4 = 0;
      }
      const int v = __builtin_ctzll(m);
      const int ww = wave::readlane(W.w, v), k = ww & 0xffff, krl = (ww >> 22) & 63;
      const int Lr = wave::readlane(S.vs0, v) & 63, RL = Lr + 1;
      const int reco = (int)((unsigned)wave::readlane(S.vs1, v) >> 16), cb = wave::readlane(S.vs2, v);
      const int pa = wave::bcast(pf.pa, slot4);
      const int sidx = k - Lr + lane;  // lane j looks at load stop k-Lr+j
      int q = 0, key = 0;
      int col = krl + 1 + lane;  // (k - Lr + lane) mod RL
      if (col >= RL) col -= RL;
      int32_t* cell = g_rec + reco + krl * RL + (lane < Lr ? col : 0);
      if (lane < Lr && sidx >= 0) {  // prefetched (tick_prefetch_arrivals_lean)
        q = slot4 == 0 ? pf.q[0] : slot4 == 1 ? pf.q[1] : slot4 == 2 ? pf.q[2] : pf.q[3];
        key = stop_arrival((uint32_t)(slot4 == 0 ? pf.key[0] : slot4 == 1 ? pf.key[1] : slot4 == 2 ? pf.key[2] : pf.key[3]));
      }
      const bool has = q > 0;
      const uint64_t hm = wave::ballot(has);
      if (has) {
        *cell = 0;
        const int i = n_ent + __builtin_popcountll(hm & lt_mask);
        if (i < KD(misc_cap)) { ent[3 * i] = key; ent[3 * i + 1] = v | ((pa & 0xff) << 6) | ((cb + (pa >> 8)) << 12); ent[3 * i + 2] = q; }
      }
      n_ent += __builtin_popcountll(hm);
      n_ves++;
    }
    if (n_ent > KD(misc_cap)) { status |= 16; n_ent = KD(misc_cap); }
    wave::sync();
    if (n_ves > 1 && n_ent > 64) {  // more records than lanes (very rare): sort the list after all (serial insertion sort)
      if (lane == 0) {
        for (int i = 1; i < n_ent; i++) {
          const int key = ent[3 * i], vv = ent[3 * i + 1], qq = ent[3 * i + 2];
          int j = i;
          while (j > 0 && ent[3 * (j - 1)] > key) { ent[3 * j] = ent[3 * (j - 1)]; ent[3 * j + 1] = ent[3 * (j - 1) + 1]; ent[3 * j + 2] = ent[3 * (j - 1) + 2]; j--; }
          ent[3 * j] = key; ent[3 * j + 1] = vv; ent[3 * j + 2] = qq;
        }
      }
      wave::sync();
    }
    for (int i0 = 0; i0 < n_ent; i0 += 64) {  // one lane per record
      const int i = i0 + lane;
      const bool has = i < n_ent;
      const int nb = (n_ent - i0) < 64 ? (n_ent - i0) : 64;
      const int key = has ? ent[3 * i] : 0x7fffffff, e1 = has ? ent[3 * i + 1] : 0, q = has ? ent[3 * i + 2] : 0;
      int rank = lane;
      if (n_ves > 1 && n_ent > 1 && n_ent <= 64) {  // stable rank by load tick (entries of one vessel are already in order)
        rank = 0;
        for (int j = 0; j < n_ent; j++) {  // wave-uniform
          const int kj = wave::readlane(key, j);
          rank += (kj < key || (kj == key && j < lane)) ? 1 : 0;
        }
      }
      const int v = e1 & 63, p = (e1 >> 6) & 63, fcell = (int)((unsigned)e1 >> 12);
      const double eb = T.er_base[p], en = T.er_noise[p];
      const double r = KD(use_buffer_rng) ? mt_draw_batch(L.mt_buf, idx_buf, has ? rank : -1, nb, buf_twisted) : 0.0;
      if (has) {
        wave::lds_add(&FV(VA_FULL, v), -q);
        wave::lds_add(&FV(VA_REMAINING_SPACE, v), q);
        wave::lds_add(&L.frame[KD(f_fov) + fcell], -q);
        const int b = KD(use_buffer_rng) ? (int)ceil(apply_noise(eb, en, r)) : T.er_delay[p];
        if (b == 0) {  // immediate RETURN_EMPTY
          wave::lds_add(&FP(PA_EMPTY, p), q);
        } else {
          wave::lds_add(&FP(PA_ON_CONSIGNEE, p), q);
          if (b > 0) { const int sl = slot + b >= H ? slot + b - H : slot + b; wave::lds_add(&RING_EMPTY(sl, p), q); }
        }
      }
    }
  }
  if (lane < P && due_empty) {  // RETURN_EMPTY :695-706
    wave::lds_add(&FP(PA_ON_CONSIGNEE, lane), -due_empty);
    wave::lds_add(&FP(PA_EMPTY, lane), due_empty);
    RING_EMPTY(slot, lane) = 0;
  }
  wave::sync();
  prof.mark(PF_DEPART_RETURNS);

  // ---------------- B3. orders (:448-497) — one BUFFER draw per order, in generation order; one lane per pair
  {
    int32_t* pre = L.misc;  // inclusive prefix of the order quantities over all pairs (ent[] is dead by now)
    if (lane < P) L.srcn[lane] = 0;  // per-port sum of immediately returned containers
    int rank[RR_NB], cnt = 0;
    double r[RR_NB];
#pragma unroll
    for (int b = 0; b < RR_NB; b++) {
      const bool has = oqr[b] > 0;
      const uint64_t m = wave::ballot(has);
      rank[b] = has ? cnt + __builtin_popcountll(m & lt_mask) : -1;
      cnt += __builtin_popcountll(m);
      r[b] = 0.0;
    }
    if (KD(use_buffer_rng)) mt_draw_multi<RR_NB>(L.mt_buf, idx_buf, rank, cnt, buf_twisted, r);
    int carry = 0, incl[RR_NB], bdv[RR_NB];
    bool any_imm = false;
#pragma unroll
    for (int b = 0; b < RR_NB; b++) {  // the pairs' delay tables: one more round trip for all three batches
      const int src = S.ps[b] & 0xff;
      bdv[b] = 0;
      if (KD(use_buffer_rng)) {
        const double fb = T.fr_base[src], fn = T.fr_noise[src];
        if (oqr[b] > 0) bdv[b] = (int)ceil(apply_noise(fb, fn, r[b]));
      } else if (oqr[b] > 0) {
        bdv[b] = T.fr_delay[src];
      }
    }
#pragma unroll
    for (int b = 0; b < RR_NB; b++) {
      const int k = b * 64 + lane;
      const int q = oqr[b];
      const bool has = q > 0;
      const int bd = bdv[b];
      incl[b] = wave::scan_incl_add(q) + carry;
      carry = wave::bcast(incl[b], 63);
      if (k < NT) pre[k] = incl[b];
      if (has) oqr[b] = q | ((bd < 0 ? 0 : (bd > 126 ? 127 : bd + 1)) << 24);
      any_imm = any_imm || (wave::ballot(has && bd == 0) != 0);
    }
    wave::sync();
    // exec_j = min(q_j, max(0, empty0 - sum of the port's earlier orders)): the sequential hand-out of :462-478
    int pbase[RR_NB], emp[RR_NB];
#pragma unroll
    for (int b = 0; b < RR_NB; b++) {  // the gathers of all three batches in one round trip
      const int src = S.ps[b] & 0xff, off = (int)((unsigned)S.ps[b] >> 8);
      pbase[b] = pre[off > 0 ? off - 1 : 0];
      emp[b] = FP(PA_EMPTY, src);
    }
#pragma unroll
    for (int b = 0; b < RR_NB; b++) {
      const int k = b * 64 + lane;
      const int qb = oqr[b];
      const int q = qb & 0xffffff;
      if (q > 0) {
        const int src = S.ps[b] & 0xff, off = (int)((unsigned)S.ps[b] >> 8);
        const int excl = incl[b] - q - (off > 0 ? pbase[b] : 0);
        const int avail = emp[b] - excl;
        const int exec = avail <= 0 ? 0 : (q < avail ? q : avail);
        const int bd = ((qb >> 24) & 0x7f) - 1;  // -1 = scheduled into the past: the containers never come back
        if (bd == 0) { wave::lds_add(&FOPK(k), exec); wave::lds_add(&L.srcn[src], exec); }  // RETURN_FULL right away (:494-497)
        else if (bd > 0) {
          int sl = slot + bd;
          sl = sl >= H ? sl - H : sl;
#pragma unroll
          for (int h = 0; h < RR_H; h++) rr.rf[h][b] += (sl == h) ? exec : 0;
        }
      }
    }
    wave::sync();
    if (lane < P) {  // port totals in closed form
      const int p = lane;
      const int off = S.pt & 0xffff, cnt_p = (int)((unsigned)S.pt >> 16);
      const int hi = pre[cnt_p > 0 ? off + cnt_p - 1 : 0], lo = pre[off > 0 ? off - 1 : 0];
      const int empty0 = FP(PA_EMPTY, p), imm_v = L.srcn[p], bk = FP(PA_BOOKING, p), sh = FP(PA_SHORTAGE, p);
      if (cnt_p > 0) {
        const int sumq = hi - (off > 0 ? lo : 0);
        if (sumq > 0) {
          const int short_ = sumq > empty0 ? sumq - empty0 : 0;
          const int exec = sumq - short_;
          const int imm = any_imm ? imm_v : 0;
          const int booking = bk + sumq, shortage = sh + short_;
          FP(PA_BOOKING, p) = booking; wave::lds_add(&FP(PA_ACC_BOOKING, p), sumq);
          FP(PA_SHORTAGE, p) = shortage; wave::lds_add(&FP(PA_ACC_SHORTAGE, p), short_);
          FP(PA_FULFILLMENT, p) = booking - shortage;  // port.py:88-97
          FP(PA_EMPTY, p) = empty0 - exec;
          wave::lds_add(&FP(PA_ON_SHIPPER, p), exec - imm);
          wave::lds_add(&FP(PA_FULL, p), imm);
        }
      }
    }
  }
  wave::sync();
  prof.mark(PF_ORDERS);

  // ---------------- B4. arrivals + full loading, in vessel order (:600-632, :524-598); lane i = i-th next stop
  if (arr_mask) {
    int a_idx = 0;
    if (__builtin_popcountll(arr_mask) > 4) {  // the discharge pass re-used the prefetch registers: fetch group 0 again
      tick_prefetch_arrivals_lean(K, env, L, arr_mask, pf, Z);
      tick_prefetch_land(pf);
    }
    const int lvv = lane < V ? lane : 0;
    const int r_full = FV(VA_FULL, lvv), r_empty = FV(VA_EMPTY, lvv), r_cap = FV(VA_CAPACITY, lvv);  // (after the discharges; an arrival only changes its own vessel's words)
    for (uint64_t m = arr_mask; m; m &= m - 1, a_idx++) {  // wave-uniform over the arriving vessels
      if (a_idx == 4) {
        tick_prefetch_arrivals_lean(K, env, L, m, pf, Z);
        tick_prefetch_land(pf);
        a_idx = 0;
      }
      const int v = __builtin_ctzll(m);
      const int ww = wave::readlane(W.w, v), k = ww & 0xffff, pos = (ww >> 16) & 63, krl = (ww >> 22) & 63, cap = wave::readlane(r_cap, v);
      int full = wave::readlane(r_full, v), empty = wave::readlane(r_empty, v);
      const int full0 = full, empty0 = empty;
      const int vs0 = wave::readlane(S.vs0, v), vs1 = wave::readlane(S.vs1, v), cb = wave::readlane(S.vs2, v);
      const int Lr = vs0 & 63, n_distinct = (vs0 >> 6) & 63, rb = (int)((unsigned)vs0 >> 12), RL = Lr + 1;
      const int lo = vs1 & 0xffff, reco = (int)((unsigned)vs1 >> 16);
      const int p = wave::bcast(pf.pa, a_idx) & 0xff;
      const int ns = wave::bcast(pf.ns, a_idx);  // prefetched by lane a_idx (tick_prefetch_arrivals_lean)
      const uint32_t st_k = (uint32_t)wave::bcast((int)pf.stk, a_idx);
      const uint32_t st_k1 = (uint32_t)wave::bcast((int)pf.stk1, a_idx);
      prof.mark(10);
      // lane i: the i-th stop after this one — route position, compact matrix column, predicted tick, pending orders to its port:
      // one round trip for the three gathers
      const bool act = lane < Lr;
      int xi = pos + lane, xn = pos + 1 + lane;  // leg out of stop i-1, position of stop i
      xi = act ? (xi >= Lr ? xi - Lr : xi) : 0;   // lanes < Lr stay below 2 Lr: a conditional subtraction, no integer division
      xn = act ? (xn >= Lr ? xn - Lr : xn) : 0;
      const int kk = a_idx == 0 ? pf.kk[0] : a_idx == 1 ? pf.kk[1] : a_idx == 2 ? pf.kk[2] : pf.kk[3];  // order pair (p -> port of stop i), -1: none
      const int leg_raw = T.leg_time[lo + xi], c_i = T.route_cidx[rb + xn], pend_raw = FOPK(kk >= 0 ? kk : 0);
      const int leg = act ? leg_raw : 0;
      const int tick_i = t + wave::scan_incl_add(leg);  // vessel_future_stops_prediction.py:49-85
      bool dup_later = false, dup_earlier = false;  // the route may visit a port twice ...
      if (n_distinct != Lr) {  // ... (wave-uniform) but most routes do not: as many plan cells as stops
        for (int j = 0; j < Lr; j++) {
          const int cj = wave::bcast(c_i, j);
          dup_later = dup_later || (j > lane && cj == c_i);
          dup_earlier = dup_earlier || (j < lane && cj == c_i);
        }
      }
      if (lane < Lr && !dup_later) L.frame[KD(f_plans) + cb + c_i] = tick_i;  // vessel_sailing_plan_wrapper.py:24-28 (later stops overwrite)
      prof.mark(11);
      // load full (:551-587): the sequential hand-out of `acceptable` over the next Lr stops is a clamped prefix sum;
      // a second visit of the same port within the window gets nothing (first visit took all, or space ran out)
      const int acceptable = (int)floor((double)(cap - full * KD(vol)) / (double)KD(vol));
      const bool lv = lane < Lr && (k + 1 + lane) < ns && !dup_earlier;  // python slice truncation at the end of the stop list
      const int pend = (lv && kk >= 0) ? pend_raw : 0;
      const int incl = wave::scan_incl_add(pend);
      int l = 0;
      if (acceptable > 0 && pend > 0) { const int room = acceptable - (incl - pend); l = room <= 0 ? 0 : (pend < room ? pend : room); }
      const int loaded_total = wave::bcast(incl < acceptable ? incl : (acceptable > 0 ? acceptable : 0), 63);
      if (l > 0) {
        FOPK(kk) = pend - l;
        wave::lds_add(&L.frame[KD(f_fov) + cb + c_i], l);
        int row = krl + 1 + lane;  // (k + 1 + lane) mod RL
        if (row >= RL) row -= RL;
        g_rec[reco + row * RL + krl] = l;  // cell is empty: (dst stop, load stop) pairs are unique
      }
      full += loaded_total;
      int early = 0;
      if ((long long)(full + empty) * KD(vol) > (long long)cap) {
        early = (full + empty) - (int)ceil((double)cap / (double)KD(vol));
        empty -= early;
      }
      const int evt_new = t + stop_parking(st_k), next_new = k + 1 < ns ? stop_arrival(st_k1) : 0x7fffffff;
      if (lane == 0) {
        FV(VA_LAST_LOC_IDX, v) = k;
        FV(VA_IS_PARKING, v) = 1;
        FV(VA_LOC_PORT_IDX, v) = p;
        wave::lds_add(&FP(PA_FULL, p), -loaded_total);
        wave::lds_add(&FP(PA_EMPTY, p), early);
        FV(VA_FULL, v) = full;
        FV(VA_EMPTY, v) = empty;
        FV(VA_EARLY_DISCHARGE, v) = early;
        wave::lds_add(&FV(VA_REMAINING_SPACE, v), (full0 + empty0) - (full + empty));  // vessel.py:113-120: total_space - full - empty
        V_EVT(v) = evt_new;
        V_NEXT(v) = next_new;
      }
      if (lane == v) { W.w |= 1 << 28; W.evt = evt_new; W.next = next_new; }
      wave::sync();
    }
  }
  prof.mark(PF_ARRIVALS);
  return arr_mask;
}
#endif  // MRX_LEAN2

// One tick, phases A..B4.  Returns the mask of vessels that arrived (their decisions follow).
template <bool PG>
MRX_DEV uint64_t run_tick(const CimParams& K, int env, Lds& L, int t, TickPf& pf, int& idx_ord, int& idx_buf, int& status, Prof& prof,
                          bool& ord_twisted, bool& buf_twisted, RingRegs& rr) {
  const int lane = wave::lane();
  const int P = KD(P), V = KD(V), NT = KD(NT), H = KD(H);
  const Tabs& T = L.tab;

  // ---------------- A. orders of this tick -> L.oq[pair]
  const uint64_t arr_mask = pf.arr_mask;
  int32_t* g_rec = K.rec + (size_t)env * KD(REC_W);
#ifdef MRX_LEAN
  int oqr[RR_NB];  // order quantity of pair 64 b + lane (later | (buffer ticks + 1) << 24), in registers
#pragma unroll
  for (int b = 0; b < RR_NB; b++) oqr[b] = (b * 64 + lane < NT) ? pf.oqr[b] : 0;
#define OQ_GET(b, k) oqr[b]
#define OQ_SET(b, k, x) oqr[b] = (x)
#else
#define OQ_GET(b, k) L.oq[k]
#define OQ_SET(b, k, x) L.oq[k] = (x)
#endif
  if constexpr (PG) {
    // drawn at reset (order table): the row was requested one round trip ago (tick_prefetch)
#ifndef MRX_LEAN
#pragma unroll
    for (int b = 0; b < 3; b++) { const int k = b * 64 + lane; if (k < NT) L.oq[k] = pf.oqr[b]; }
#endif
    if (NT > 192) {
      for (int k = 192 + lane; k < NT; k += 64) L.oq[k] = order_cell(K, env, t - KD(start_tick), k);
    }
  } else {
    long long otg = (long long)pf.otg;
    bool gen = true;
    if (KD(order_mode) == 1) {  // UNFIXED :327-333 (total_empty_number from business_engine.py:134-136)
      long long mine = 0;
      if (lane < P) mine += FP(PA_EMPTY, lane);
      if (lane < V) mine += FV(VA_EMPTY, lane);
      const long long delta = (long long)KD(total_containers) - wave::reduce_add(mine);
      if (otg <= delta) gen = false; else otg -= delta;
    }
    if (gen) gen_orders(K, L, otg, idx_ord, pf, ord_twisted);
    else for (int k = lane; k < NT; k += 64) L.oq[k] = 0;
  }
  wave::sync();
  prof.mark(PF_ORDER_GEN);

  // ---------------- B1. departures (business_engine.py:634-656; past stops = shift + append)
  const uint64_t zmask = KD(start_tick) > 0 ? zombie_mask(L) : 0ull;
  if (lane < V) {
    const int v = lane;
    const bool z = (zmask >> v) & 1ull;
    if ((z || FV(VA_IS_PARKING, v)) && V_EVT(v) == t) {
      // (the past-stop lists — shift + append of the stop that is being left — are derived from next_loc_idx: stop_list_value)
      const uint32_t* srow = K.stops + ((size_t)env * V + v) * KD(SMAX);
      FV(VA_NEXT_LOC_IDX, v) += 1;
      { const int Lr = T.v_route_len[v]; const int x = V_POS(v) + 1, y = V_KRL(v) + 1; V_POS(v) = x == Lr ? 0 : x; V_KRL(v) = y == Lr + 1 ? 0 : y; }
      FV(VA_IS_PARKING, v) = 0;
      FV(VA_LOC_PORT_IDX, v) = -1;
      if (z) {  // the next scheduled departure event
        const int ks = V_NEXT(v) + 1, ns = K.nstops[(size_t)env * V + v];
        V_NEXT(v) = ks;
        const uint32_t st = srow[ks < KD(SMAX) ? ks : KD(SMAX) - 1];
        V_EVT(v) = ks < ns ? stop_arrival(st) + stop_parking(st) : 0x7fffffff;
      } else {
        V_EVT(v) = V_NEXT(v);  // arrival tick of the next stop, cached at the previous arrival
      }
    }
  }

  // ---------------- B2. returns / discharges scheduled for this tick (all additive -> lane-parallel + LDS adds)
  const int slot = t % H;
  // source port of pair k from the staged table: no global load inside these loops (its s_waitcnt would land behind a
  // control-flow merge as vmcnt(0), which on gfx950 also drains the snapshot stores issued just before)
#define MRX_PAIR_SRC(k0, k) ((int)T.pair_src[(k)])
#pragma unroll
  for (int k0 = 0; k0 < NT; k0 += 64) {  // RETURN_FULL :499-522, one lane per (src, dst) pair
    const int k = k0 + lane;
#ifdef MRX_LEAN
    int q = 0;
#pragma unroll
    for (int h = 0; h < RR_H; h++) { const bool mine = slot == h; q = mine ? rr.rf[h][k0 / 64] : q; rr.rf[h][k0 / 64] = mine ? 0 : rr.rf[h][k0 / 64]; }
    if (k < NT && q) {
#else
    if (k < NT) {
      const int q = RING_FULL(slot, k);
      if (q) RING_FULL(slot, k) = 0;
      if (q) {
#endif
        const int src = MRX_PAIR_SRC(k0, k);
        FOPK(k) += q;  // the pair owns this cell
        wave::lds_add(&FP(PA_ON_SHIPPER, src), -q);
        wave::lds_add(&FP(PA_FULL, src), q);
#ifndef MRX_LEAN
      }
#endif
    }
  }
  wave::sync();
  if (arr_mask) {
    // DISCHARGE_FULL :658-693.  One BUFFER draw per original event, in the order the events were scheduled:
    // (tick of the load, vessel index) — SURVEY.md §9.2.  Records: rec[v][dst_stop % RL][load_stop % RL].
    int32_t* ent = L.misc;  // (key, v, q) triples
    int n_ent = 0, n_ves = 0;
    int slot4 = 0;
    const int lvd = lane < V ? lane : 0;
    const int d_k = FV(VA_NEXT_LOC_IDX, lvd), d_krl = V_KRL(lvd), d_len = T.v_route_len[lvd], d_rec = T.rec_off[lvd];  // rows, lane = vessel
    for (uint64_t m = arr_mask; m; m &= m - 1, slot4++) {  // wave-uniform
      if (slot4 == 4) {  // more than four arrivals in one tick (rare): fetch the next four now
        tick_prefetch_arrivals(K, env, L, m, pf);
        tick_prefetch_land(pf);
        slot4 = 0;
      }
      const int v = __builtin_ctzll(m);
      const int k = wave::readlane(d_k, v);
      const int Lr = wave::readlane(d_len, v), RL = Lr + 1;
      const int krl = wave::readlane(d_krl, v);
      const int sidx = k - Lr + lane;  // lane j looks at load stop k-Lr+j
      int q = 0, key = 0;
      int col = krl + 1 + lane;  // (k - Lr + lane) mod RL
      if (col >= RL) col -= RL;
      int32_t* cell = g_rec + wave::readlane(d_rec, v) + krl * RL + (lane < Lr ? col : 0);
      if (lane < Lr && sidx >= 0) {  // prefetched (tick_prefetch_arrivals)
        q = slot4 == 0 ? pf.q[0] : slot4 == 1 ? pf.q[1] : slot4 == 2 ? pf.q[2] : pf.q[3];
        key = stop_arrival((uint32_t)(slot4 == 0 ? pf.key[0] : slot4 == 1 ? pf.key[1] : slot4 == 2 ? pf.key[2] : pf.key[3]));
      }
      const bool has = q > 0;
      const uint64_t hm = wave::ballot(has);
      if (has) {
        *cell = 0;
        const int i = n_ent + __builtin_popcountll(hm & ((1ull << lane) - 1ull));
        if (i < KD(misc_cap)) { ent[3 * i] = key; ent[3 * i + 1] = v; ent[3 * i + 2] = q; }
      }
      n_ent += __builtin_popcountll(hm);
      n_ves++;
    }
    if (n_ent > KD(misc_cap)) { status |= 16; n_ent = KD(misc_cap); }
    wave::sync();
    if (n_ves > 1 && n_ent > 1 && n_ent <= 64) {
      // merge by load tick, stable (several vessels arriving in one tick): lane i ranks entry i among all entries
      const bool mine = lane < n_ent;
      const int key = mine ? ent[3 * lane] : 0x7fffffff, vv = mine ? ent[3 * lane + 1] : 0, qq = mine ? ent[3 * lane + 2] : 0;
      int rank = 0;
      for (int j = 0; j < n_ent; j++) {  // wave-uniform
        const int kj = wave::readlane(key, j);
        rank += (kj < key || (kj == key && j < lane)) ? 1 : 0;
      }
      wave::sync();
      if (mine) { ent[3 * rank] = key; ent[3 * rank + 1] = vv; ent[3 * rank + 2] = qq; }
      wave::sync();
    } else if (n_ves > 1 && n_ent > 1) {  // more records than lanes: serial insertion sort
      if (lane == 0) {
        for (int i = 1; i < n_ent; i++) {
          const int key = ent[3 * i], vv = ent[3 * i + 1], qq = ent[3 * i + 2];
          int j = i;
          while (j > 0 && ent[3 * (j - 1)] > key) { ent[3 * j] = ent[3 * (j - 1)]; ent[3 * j + 1] = ent[3 * (j - 1) + 1]; ent[3 * j + 2] = ent[3 * (j - 1) + 2]; j--; }
          ent[3 * j] = key; ent[3 * j + 1] = vv; ent[3 * j + 2] = qq;
        }
      }
      wave::sync();
    }
    for (int i0 = 0; i0 < n_ent; i0 += 64) {  // one lane per record; every effect is an addition
      const int i = i0 + lane;
      const bool has = i < n_ent;
      const int nb = (n_ent - i0) < 64 ? (n_ent - i0) : 64;
      const double r = KD(use_buffer_rng) ? mt_draw_batch(L.mt_buf, idx_buf, has ? lane : -1, nb, buf_twisted) : 0.0;
      if (has) {
        const int v = ent[3 * i + 1], q = ent[3 * i + 2];
        const int rpi = T.v_route_base[v] + V_POS(v);
        const int p = T.route_port[rpi], pc = T.route_cidx[rpi];
        wave::lds_add(&FV(VA_FULL, v), -q);
        wave::lds_add(&FV(VA_REMAINING_SPACE, v), q);
        wave::lds_add(&FOVC(v, pc), -q);
        const int b = KD(use_buffer_rng) ? (int)ceil(apply_noise(T.er_base[p], T.er_noise[p], r)) : T.er_delay[p];
        if (b == 0) {  // immediate RETURN_EMPTY
          wave::lds_add(&FP(PA_EMPTY, p), q);
        } else {
          wave::lds_add(&FP(PA_ON_CONSIGNEE, p), q);
          if (b > 0) { const int sl = slot + b >= H ? slot + b - H : slot + b; wave::lds_add(&RING_EMPTY(sl, p), q); }
        }
      }
    }
    wave::sync();
  }
  if (lane < P) {  // RETURN_EMPTY :695-706
    const int q = RING_EMPTY(slot, lane);
    if (q) {
      FP(PA_ON_CONSIGNEE, lane) -= q;
      FP(PA_EMPTY, lane) += q;
      RING_EMPTY(slot, lane) = 0;
    }
  }
  wave::sync();
  prof.mark(PF_DEPART_RETURNS);

  // ---------------- B3. orders (:448-497) — one BUFFER draw per order, in generation order; one lane per pair
  {
    int32_t* pre = L.misc;  // inclusive prefix of the order quantities over all pairs (ent[] is dead by now)
    int carry = 0;
    bool any_imm = false;
    if (lane < P) L.srcn[lane] = 0;  // per-port sum of immediately returned containers
#pragma unroll
    for (int k0 = 0; k0 < NT; k0 += 64) {
      const int k = k0 + lane;
      const int q = k < NT ? OQ_GET(k0 / 64, k) : 0;
      const bool has = q > 0;
      int b = 0;
      if (KD(use_buffer_rng)) {
        const uint64_t m = wave::ballot(has);
        const int rank = has ? __builtin_popcountll(m & ((1ull << lane) - 1ull)) : -1;
        const double r = mt_draw_batch(L.mt_buf, idx_buf, rank, __builtin_popcountll(m), buf_twisted);
        if (has) { const int src = MRX_PAIR_SRC(k0, k); b = (int)ceil(apply_noise(T.fr_base[src], T.fr_noise[src], r)); }
      } else if (has) {
        b = T.fr_delay[MRX_PAIR_SRC(k0, k)];
      }
      const int incl = wave::scan_incl_add(q) + carry;
      carry = wave::bcast(incl, 63);
      if (k < NT) { pre[k] = incl; if (has) OQ_SET(k0 / 64, k, q | ((b < 0 ? 0 : (b > 126 ? 127 : b + 1)) << 24)); }
      any_imm = any_imm || (wave::ballot(has && b == 0) != 0);
    }
    wave::sync();
    // exec_j = min(q_j, max(0, empty0 - sum of the port's earlier orders)): the sequential hand-out of :462-478
#pragma unroll
    for (int k0 = 0; k0 < NT; k0 += 64) {
      const int k = k0 + lane;
      const int qb = k < NT ? OQ_GET(k0 / 64, k) : 0;
      const int q = qb & 0xffffff;
      if (q > 0) {
        const int src = MRX_PAIR_SRC(k0, k);
        const int off = T.tgt_off[src];
        const int excl = pre[k] - q - (off > 0 ? pre[off - 1] : 0);
        const int avail = FP(PA_EMPTY, src) - excl;
        const int exec = avail <= 0 ? 0 : (q < avail ? q : avail);
        const int b = ((qb >> 24) & 0x7f) - 1;  // -1 = scheduled into the past: the containers never come back
        if (b == 0) { FOPK(k) += exec; wave::lds_add(&L.srcn[src], exec); }  // RETURN_FULL right away (:494-497)
        else if (b > 0) {
          int sl = slot + b;
          sl = sl >= H ? sl - H : sl;
#ifdef MRX_LEAN
#pragma unroll
          for (int h = 0; h < RR_H; h++) rr.rf[h][k0 / 64] += (sl == h) ? exec : 0;
#else
          RING_FULL(sl, k) += exec;
#endif
        }
      }
    }
    wave::sync();
    if (lane < P) {  // port totals in closed form
      const int p = lane;
      const int off = T.tgt_off[p], cnt = T.tgt_off[p + 1] - off;
      if (cnt > 0) {
        const int sumq = pre[off + cnt - 1] - (off > 0 ? pre[off - 1] : 0);
        if (sumq > 0) {
          const int empty0 = FP(PA_EMPTY, p);
          const int short_ = sumq > empty0 ? sumq - empty0 : 0;
          const int exec = sumq - short_;
          const int imm = any_imm ? L.srcn[p] : 0;
          const int booking = FP(PA_BOOKING, p) + sumq, shortage = FP(PA_SHORTAGE, p) + short_;
          FP(PA_BOOKING, p) = booking; FP(PA_ACC_BOOKING, p) += sumq;
          FP(PA_SHORTAGE, p) = shortage; FP(PA_ACC_SHORTAGE, p) += short_;
          FP(PA_FULFILLMENT, p) = booking - shortage;  // port.py:88-97
          FP(PA_EMPTY, p) = empty0 - exec;
          FP(PA_ON_SHIPPER, p) += exec - imm;
          FP(PA_FULL, p) += imm;
        }
      }
    }
  }
#undef MRX_PAIR_SRC
#undef OQ_GET
#undef OQ_SET
  wave::sync();
  prof.mark(PF_ORDERS);

  // ---------------- B4. arrivals + full loading, in vessel order (:600-632, :524-598); lane i = i-th next stop
  if (arr_mask) {
    int a_idx = 0;
    if (__builtin_popcountll(arr_mask) > 4) {  // the discharge pass re-used the prefetch registers: fetch group 0 again
      tick_prefetch_arrivals(K, env, L, arr_mask, pf);
      tick_prefetch_land(pf);
    }
    // per-vessel words as rows (lane = vessel): one LDS trip, then register reads per arriving vessel (an arrival only
    // changes its own vessel's words)
    const int lvv = lane < V ? lane : 0;
    const int r_k = FV(VA_NEXT_LOC_IDX, lvv), r_pos = V_POS(lvv), r_krl = V_KRL(lvv), r_cap = FV(VA_CAPACITY, lvv),
              r_full = FV(VA_FULL, lvv), r_empty = FV(VA_EMPTY, lvv), r_len = T.v_route_len[lvv], r_rb = T.v_route_base[lvv];
    for (uint64_t m = arr_mask; m; m &= m - 1, a_idx++) {  // wave-uniform over the arriving vessels
      if (a_idx == 4) {
        tick_prefetch_arrivals(K, env, L, m, pf);
        tick_prefetch_land(pf);
        a_idx = 0;
      }
      const int v = __builtin_ctzll(m);
      const int k = wave::readlane(r_k, v), pos = wave::readlane(r_pos, v), krl = wave::readlane(r_krl, v),
                cap = wave::readlane(r_cap, v);
      int full = wave::readlane(r_full, v), empty = wave::readlane(r_empty, v);
      const int full0 = full, empty0 = empty;
      const int Lr = wave::readlane(r_len, v), rb = wave::readlane(r_rb, v), RL = Lr + 1;
      const int p = T.route_port[rb + pos];
      const int ns = wave::bcast(pf.ns, a_idx);  // prefetched by lane a_idx (tick_prefetch_arrivals)
      const uint32_t st_k = (uint32_t)wave::bcast((int)pf.stk, a_idx);
      const uint32_t st_k1 = (uint32_t)wave::bcast((int)pf.stk1, a_idx);
      prof.mark(10);
      // lane i: the i-th stop after this one — route position, port, compact matrix column, predicted tick
      const bool act = lane < Lr;
      int xi = pos + lane, xn = pos + 1 + lane;  // leg out of stop i-1, position of stop i
      xi = act ? (xi >= Lr ? xi - Lr : xi) : 0;   // lanes < Lr stay below 2 Lr: a conditional subtraction, no integer division
      xn = act ? (xn >= Lr ? xn - Lr : xn) : 0;
      const int leg = act ? T.leg_time[T.leg_off[v] + xi] : 0;
      const int tick_i = t + wave::scan_incl_add(leg);  // vessel_future_stops_prediction.py:49-85
      const int port_i = T.route_port[rb + xn], c_i = T.route_cidx[rb + xn];
      bool dup_later = false, dup_earlier = false;  // the route may visit a port twice ...
      const int n_distinct = (v + 1 < V ? (int)T.v_cbase[v + 1] : KD(NC)) - (int)T.v_cbase[v];
      if (n_distinct != Lr) {  // ... (wave-uniform) but most routes do not: as many plan cells as stops
        for (int j = 0; j < Lr; j++) {
          const int cj = wave::bcast(c_i, j);
          dup_later = dup_later || (j > lane && cj == c_i);
          dup_earlier = dup_earlier || (j < lane && cj == c_i);
        }
      }
      if (lane < Lr && !dup_later) PLANC(v, c_i) = tick_i;  // vessel_sailing_plan_wrapper.py:24-28 (later stops overwrite)
      prof.mark(11);
      // load full (:551-587): the sequential hand-out of `acceptable` over the next Lr stops is a clamped prefix sum;
      // a second visit of the same port within the window gets nothing (first visit took all, or space ran out)
      const int acceptable = (int)floor((double)(cap - full * KD(vol)) / (double)KD(vol));
      const bool lv = lane < Lr && (k + 1 + lane) < ns && !dup_earlier;  // python slice truncation at the end of the stop list
      // order pair (p -> port_i), if the port ships there at all: prefetched (tick_prefetch_arrivals)
      const int kk = a_idx == 0 ? pf.kk[0] : a_idx == 1 ? pf.kk[1] : a_idx == 2 ? pf.kk[2] : pf.kk[3];
      const int pend = (lv && kk >= 0) ? FOPK(kk) : 0;
      const int incl = wave::scan_incl_add(pend);
      int l = 0;
      if (acceptable > 0 && pend > 0) { const int room = acceptable - (incl - pend); l = room <= 0 ? 0 : (pend < room ? pend : room); }
      const int loaded_total = wave::bcast(incl < acceptable ? incl : (acceptable > 0 ? acceptable : 0), 63);
      if (l > 0) {
        FOPK(kk) = pend - l;
        FOVC(v, c_i) += l;
        int row = krl + 1 + lane;  // (k + 1 + lane) mod RL
        if (row >= RL) row -= RL;
        g_rec[T.rec_off[v] + row * RL + krl] = l;  // cell is empty: (dst stop, load stop) pairs are unique
      }
      full += loaded_total;
      int early = 0;
      if ((long long)(full + empty) * KD(vol) > (long long)cap) {
        early = (full + empty) - (int)ceil((double)cap / (double)KD(vol));
        empty -= early;
      }
      wave::sync();
      if (lane == 0) {
        FV(VA_LAST_LOC_IDX, v) = k;
        FV(VA_IS_PARKING, v) = 1;
        FV(VA_LOC_PORT_IDX, v) = p;
        FP(PA_FULL, p) -= loaded_total;
        FP(PA_EMPTY, p) += early;
        FV(VA_FULL, v) = full;
        FV(VA_EMPTY, v) = empty;
        FV(VA_EARLY_DISCHARGE, v) = early;
        FV(VA_REMAINING_SPACE, v) += (full0 + empty0) - (full + empty);  // vessel.py:113-120: total_space - full - empty
        V_EVT(v) = t + stop_parking(st_k);
        V_NEXT(v) = k + 1 < ns ? stop_arrival(st_k1) : 0x7fffffff;
      }
      wave::sync();
    }
  }
  prof.mark(PF_ARRIVALS);
  return arr_mask;
}

// snapshot of the LDS frame into the env's ring (np_backend.pyx:481-518: slot = fi mod S)
MRX_DEV void take_snapshot(const CimParams& K, const CimObs& O, int env, Lds& L, int fi) {
  const int s = fi % KD(S);
  copy_words(K.ring + ((size_t)env * KD(S) + s) * KD(FW), L.frame, KD(FW));
  if (wave::lane() == 0) K.ring_fi[(size_t)env * KD(S) + s] = fi;
  if (O.hist_n > 0 && fi < O.hist_frames) {  // per-attribute retention (mrx_cim_set_port_history): this frame's rows
    int32_t* row = O.hist + ((size_t)env * O.hist_frames + fi) * O.hist_n * KD(P);
    for (int a = 0; a < O.hist_n; a++)
      if (wave::lane() < KD(P)) row[a * KD(P) + wave::lane()] = FP(O.hist_attr[a], wave::lane());
  }
}

// ==========================================================================================
// DEVICE AGENT (CimObs::agent_*, mrx_cim_set_device_agent): the answer to the decision this step has raised, written by the wave /
// lane that wrote the decision row.  Same rule and same draw as mrx_k_cim_random_policy (cim_engine.hip): h = mix(seed, key);
// even and load > 0 -> LOAD h' % (load + 1), else DISCHARGE h' % (discharge + 1).
MRX_DEV unsigned long long mix64(unsigned long long seed, unsigned long long step) {
  unsigned long long x = seed * 0x9E3779B97F4A7C15ull + step * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}
MRX_DEV void agent_answer(const CimParams& K, const CimObs& O, int env, long long seed, int t, int port, int vessel, int load, int dis) {
  const unsigned long long key = O.agent_key >= 0 ? (unsigned long long)O.agent_key : (((unsigned long long)(unsigned)t << 8) | (unsigned)vessel) + 0x100000000ull;
  const unsigned long long x = mix64((unsigned long long)seed, key), r = x >> 1;
  int32_t* a = O.agent_actions + (size_t)env * KD(max_actions) * 4;
  a[0] = vessel; a[1] = port;
  if ((x & 1ull) == 0 && load > 0) { a[2] = (int)(r % (unsigned long long)(load + 1)); a[3] = 0; }   // MRX_ACTION_LOAD
  else { a[2] = (int)(r % (unsigned long long)(dis + 1)); a[3] = 1; }                               // MRX_ACTION_DISCHARGE
  O.agent_n_actions[env] = 1;
  if (O.agent_count) wave::global_add_nr(&O.agent_count[env], 1);
}
MRX_DEV void agent_none(const CimObs& O, int env) { O.agent_n_actions[env] = 0; }   // no decision to answer (episode over)

// ==========================================================================================
// FAST PATH of a step that cannot advance time: another vessel of the current tick is still waiting for its
// decision (2.2 vessels arrive per tick on global_trade.22p, so this is more than half of all env-steps).
// Such a step touches a dozen words — the acting port / vessel, one plan cell, the next vessel.  It never stages
// the env in LDS, and it needs ONE trip to HBM: the rows those words live in (FastRows, ~1.5 KB, lane = node) are
// requested together with the private header at kernel entry, before it is known which path the step takes; the
// words themselves are then picked out of registers (readlane).  Only the (vessel, port) -> plan-cell table lookup
// follows, and that table is a few KB shared by every env (L1/L2 resident).
struct FastRows {
  int hdr;                  // private header word PH_<lane> (lane < 16)
  int pe, tc;               // ports:   empty, transfer_cost            (lane = port)
  int ve, rs, ed, lp, per;  // vessels: empty, remaining_space, early_discharge, loc_port_idx, vessel period (lane = vessel)
  int pl[4];                // vessel_plans cells lane + 64 k (only when the compact plan block has <= 256 cells)
  int vo[8];                // fused observation (CimObs): rows of the requested vessel attributes
};

template <bool OBS>
MRX_DEV void fast_rows_request(const CimParams& K, const CimObs& O, int env, FastRows& R) {
  const int lane = wave::lane();
  const int32_t* g_live = K.live + (size_t)env * KD(FW);
  const int32_t* g_priv = K.priv + (size_t)env * KD(PW);
  const int p = lane < KD(P) ? lane : 0, v = lane < KD(V) ? lane : 0;
  R.hdr = lane < PH_COUNT ? g_priv[lane] : 0;
  R.pe = g_live[KD(f_ports) + PA_EMPTY * KD(P) + p];
  R.tc = g_live[KD(f_ports) + PA_TRANSFER_COST * KD(P) + p];
  R.ve = g_live[KD(f_vessels) + VA_EMPTY * KD(V) + v];
  R.rs = g_live[KD(f_vessels) + VA_REMAINING_SPACE * KD(V) + v];
  R.ed = g_live[KD(f_vessels) + VA_EARLY_DISCHARGE * KD(V) + v];
  R.lp = g_live[KD(f_vessels) + VA_LOC_PORT_IDX * KD(V) + v];
  R.per = g_priv[KD(pv_period) + v];
  for (int k = 0; k < 4; k++) R.pl[k] = (KD(NC) <= 256 && lane + 64 * k < KD(NC)) ? g_live[KD(f_plans) + lane + 64 * k] : 0;
  if constexpr (OBS) {
#pragma unroll
    for (int a = 0; a < 8; a++) R.vo[a] = a < OD(nv) ? g_live[KD(f_vessels) + ODA(va, a) * KD(V) + v] : 0;
  }
}

MRX_DEV double port_attr_value(int attr, int raw) { return attr == PA_TRANSFER_COST ? (double)bits_f(raw) : (double)raw; }

// Returns false if the step needs the full path.
template <bool OBS>
MRX_DEV bool fast_step(const CimParams& K, const CimObs& O, int env, const FastRows& R, int n_act, int a0v, int a0p, int a0q, int a0t,
                       int32_t* dec_out, long long* met_out, uint8_t* done_out) {
  const int lane = wave::lane();
  const int P = KD(P), V = KD(V);
  const int hdr = R.hdr;
  const int flags = wave::bcast(hdr, PH_FLAGS);
  if (flags & FL_FINISHED) {  // reference: (None, None, True) once the generator is exhausted (core.py:128-133)
    if (lane < 8) dec_out[lane] = lane == 7 ? -1 : 0;
    if (lane < 3) met_out[lane] = 0;
    if (lane == 0) { *done_out = 1; if (O.agent_mode) agent_none(O, env); }
    return true;
  }
  if ((flags & FL_FRESH) || MRX_UNALIGNED_FRAMES) return false;
  const long long agent_seed = O.agent_mode ? K.seed[env] : 0;   // (requested with the step's other loads, used at its end)
  const uint64_t pend = ((uint64_t)(uint32_t)wave::bcast(hdr, PH_PEND_HI) << 32) | (uint32_t)wave::bcast(hdr, PH_PEND_LO);
  const int cur = wave::bcast(hdr, PH_CUR_VESSEL);
  const uint64_t pend_after = pend & ~(1ull << (cur & 63));
  if (!pend_after || n_act > 1) return false;
  int32_t* g_live = K.live + (size_t)env * KD(FW);
  int32_t* g_priv = K.priv + (size_t)env * KD(PW);
  const int t = wave::bcast(hdr, PH_TICK);
  long long opnum = ((long long)wave::bcast(hdr, PH_OPNUM_HI) << 32) | (uint32_t)wave::bcast(hdr, PH_OPNUM_LO);
  const long long acc_b = ((long long)wave::bcast(hdr, PH_ACCB_HI) << 32) | (uint32_t)wave::bcast(hdr, PH_ACCB_LO);
  const long long acc_s = ((long long)wave::bcast(hdr, PH_ACCS_HI) << 32) | (uint32_t)wave::bcast(hdr, PH_ACCS_LO);
  int status = 0;
  const int v2 = __builtin_ctzll(pend_after);  // the vessel whose decision comes next
#define GP(a, p) g_live[KD(f_ports) + (a) * P + (p)]
#define GV(a, v) g_live[KD(f_vessels) + (a) * V + (v)]
  bool act = n_act == 1;
  const int av = a0v, ap = a0p, q = a0q, ty = a0t;
  if (act && (av < 0 || av >= V || ap < 0 || ap >= P || q < 0 || (ty != 0 && ty != 1))) { status |= 1; act = false; }
  const int sv = act ? av : 0, sp = act ? ap : 0;  // safe lanes when there is no (valid) action
  const int c = act ? K.cidx_dense[sv * P + sp] : -1;
  const int pe = wave::bcast(R.pe, sp), tc = wave::bcast(R.tc, sp);
  const int ve = wave::bcast(R.ve, sv), rs = wave::bcast(R.rs, sv);
  const int period = wave::bcast(R.per, sv);
  const int lp2 = wave::bcast(R.lp, v2);
  int ve2 = wave::bcast(R.ve, v2), rs2 = wave::bcast(R.rs, v2);
  const int ed2 = wave::bcast(R.ed, v2);
  int pe2 = wave::bcast(R.pe, lp2);
  int pl;
  {
    const int cc = c >= 0 ? c : 0;
    const int p0 = wave::bcast(R.pl[0], cc & 63), p1 = wave::bcast(R.pl[1], cc & 63), p2 = wave::bcast(R.pl[2], cc & 63),
              p3 = wave::bcast(R.pl[3], cc & 63);
    pl = cc < 64 ? p0 : cc < 128 ? p1 : cc < 192 ? p2 : p3;
    if (KD(NC) > 256) pl = g_live[KD(f_plans) + cc];  // large plan blocks: a second, dependent trip
  }
  // ---- the action (business_engine.py:708-748)
  bool applied = false;
  int o_pe = 0, o_tc = 0, o_ve = 0, o_rs = 0;  // new values of the acting port / vessel (fused observation)
  if (act) {
    int npe = pe, nve = ve;
    bool ok = true;
    if (ty == 1) { if (q > ve) ok = false; else { npe = pe + q; nve = ve - q; } }
    else { if (q > (pe < rs ? pe : rs)) ok = false; else { npe = pe - q; nve = ve + q; } }
    if (!ok) status |= 1;
    else {
      const int nrs = rs - (nve - ve);
      applied = true; o_pe = npe; o_tc = f_bits((float)((double)bits_f(tc) + (double)q)); o_ve = nve; o_rs = nrs;
      if (lane == 0) {
        GP(PA_EMPTY, ap) = npe;
        GP(PA_TRANSFER_COST, ap) = f_bits((float)((double)bits_f(tc) + (double)q));
        GV(VA_EMPTY, av) = nve;
        GV(VA_REMAINING_SPACE, av) = nrs;
        if (c >= 0) g_live[KD(f_plans) + c] = pl + period;
      }
      if (c < 0) status |= 32;  // MRX_ENV_OFFROUTE_ACTION
      opnum += q;
      if (ap == lp2) pe2 = npe;
      if (av == v2) { ve2 = nve; rs2 = nrs; }
    }
  }
#undef GP
#undef GV
  // ---- fused observation of the next decision (vessel v2).  The ports block was written by the previous step of
  // this env, which paused at the same tick; since then only this action changed anything, so it is patched in place.
  if constexpr (OBS) {
    if (applied && lane == 0) {
      if (OD(i_empty) >= 0) O.ports[((size_t)env * P + ap) * OD(np) + OD(i_empty)] = (double)o_pe;
      if (OD(i_tc) >= 0) O.ports[((size_t)env * P + ap) * OD(np) + OD(i_tc)] = (double)bits_f(o_tc);
    }
#pragma unroll
    for (int a = 0; a < 8; a++) {
      if (a < OD(nv)) {
        int raw = wave::bcast(R.vo[a], v2);
        if (applied && av == v2) { if (ODA(va, a) == VA_EMPTY) raw = o_ve; else if (ODA(va, a) == VA_REMAINING_SPACE) raw = o_rs; }
        if (lane == 0) O.vessel[(size_t)env * OD(nv) + a] = (double)raw;
      }
    }
  }
  // ---- next decision of the same tick (action_scope :247-260; metrics unchanged since the last tick)
  if (lane == 0) {
    dec_out[0] = t; dec_out[1] = lp2; dec_out[2] = v2;
    dec_out[3] = pe2 < rs2 ? pe2 : rs2;
    dec_out[4] = ve2; dec_out[5] = ed2;
    dec_out[6] = (t - KD(start_tick)) / KD(resolution); dec_out[7] = 1;
    if (O.agent_mode) agent_answer(K, O, env, agent_seed, t, lp2, v2, pe2 < rs2 ? pe2 : rs2, ve2);
    met_out[0] = acc_b; met_out[1] = acc_s; met_out[2] = opnum;
    *done_out = 0;
    g_priv[PH_PEND_LO] = (int32_t)(uint32_t)(pend_after & 0xffffffffull);
    g_priv[PH_PEND_HI] = (int32_t)(uint32_t)(pend_after >> 32);
    g_priv[PH_CUR_VESSEL] = v2;
    g_priv[PH_OPNUM_LO] = (int32_t)(uint32_t)((unsigned long long)opnum & 0xffffffffull);
    g_priv[PH_OPNUM_HI] = (int32_t)(opnum >> 32);
    K.hint[env] = (pend_after & ~(1ull << v2)) ? 0 : ((flags & FL_LONG) ? 2 : 1);  // answering v2 leaves another decision of this tick pending: fast again
    if (status) wave::global_or(&K.status[env], status);  // fire-and-forget: a read-modify-write would wait for every store in flight
  }
  return true;
}

// ==========================================================================================
// FAST PATH, one env per LANE (persistent step kernel): the same transition as fast_step above, written as plain
// per-thread code — 64 envs of the sorted order list per wave, three dependent gathers (header + action; the acting
// port / vessel and the next vessel; that vessel's port and the plan cell).  No wave collectives: lanes may diverge.
// Returns false if the env needs the full path (more than one action, or its hint was stale).
template <bool OBS>
MRX_DEV bool fast_step_lane(const CimParams& K, const CimObs& O, int env, const int32_t* act, int n_act, int32_t* dec_out,
                            long long* met_out, uint8_t* done_out) {
  const int P = KD(P), V = KD(V);
  int32_t* g_live = K.live + (size_t)env * KD(FW);
  int32_t* g_priv = K.priv + (size_t)env * KD(PW);
  // ---- round trip 1: the 64-byte private header and the first action
  const wave::v4i h0 = wave::ld16(g_priv), h1 = wave::ld16(g_priv + 4), h3 = wave::ld16(g_priv + 12);
  wave::v4i a4 = {0, 0, 0, 0};
  if (act) a4 = wave::ld16(act);
  const int flags = h0.y;  // PH_FLAGS
  if (flags & FL_FINISHED) {  // reference: (None, None, True) once the generator is exhausted (core.py:128-133)
    for (int j = 0; j < 8; j++) dec_out[j] = j == 7 ? -1 : 0;
    for (int j = 0; j < 3; j++) met_out[j] = 0;
    *done_out = 1;
    if (O.agent_mode) agent_none(O, env);
    return true;
  }
  if ((flags & FL_FRESH) || MRX_UNALIGNED_FRAMES) return false;
  const long long agent_seed = O.agent_mode ? K.seed[env] : 0;
  const uint64_t pend = ((uint64_t)(uint32_t)h0.w << 32) | (uint32_t)h0.z;  // PH_PEND_HI, PH_PEND_LO
  const int cur = h1.x;                                                       // PH_CUR_VESSEL
  const uint64_t pend_after = pend & ~(1ull << (cur & 63));
  if (!pend_after || n_act > 1) return false;
  const int t = h0.x;  // PH_TICK
  long long opnum = ((long long)h1.z << 32) | (uint32_t)h1.y;               // PH_OPNUM_HI, PH_OPNUM_LO
  const long long acc_b = ((long long)h3.y << 32) | (uint32_t)h3.x, acc_s = ((long long)h3.w << 32) | (uint32_t)h3.z;
  int status = 0;
  const int v2 = __builtin_ctzll(pend_after);  // the vessel whose decision comes next
#define GP(a, p) g_live[KD(f_ports) + (a) * P + (p)]
#define GV(a, v) g_live[KD(f_vessels) + (a) * V + (v)]
  bool do_act = n_act == 1;
  const int av = a4.x, ap = a4.y, q = a4.z, ty = a4.w;
  if (do_act && (av < 0 || av >= V || ap < 0 || ap >= P || q < 0 || (ty != 0 && ty != 1))) { status |= 1; do_act = false; }
  const int sv = do_act ? av : 0, sp = do_act ? ap : 0;
  // ---- round trip 2
  const int c = do_act ? K.cidx_dense[sv * P + sp] : -1;
  const int pe = GP(PA_EMPTY, sp), tc = GP(PA_TRANSFER_COST, sp);
  const int ve = GV(VA_EMPTY, sv), rs = GV(VA_REMAINING_SPACE, sv);
  const int period = g_priv[KD(pv_period) + sv];
  const int lp2 = GV(VA_LOC_PORT_IDX, v2), ed2 = GV(VA_EARLY_DISCHARGE, v2);
  int ve2 = GV(VA_EMPTY, v2), rs2 = GV(VA_REMAINING_SPACE, v2);
  int vo[8];
  if constexpr (OBS) {
#pragma unroll
    for (int a = 0; a < 8; a++) vo[a] = a < OD(nv) ? GV(ODA(va, a), v2) : 0;
  }
  // ---- round trip 3
  int pe2 = GP(PA_EMPTY, lp2 >= 0 ? lp2 : 0);
  const int pl = c >= 0 ? g_live[KD(f_plans) + c] : 0;
  // ---- the action (business_engine.py:708-748)
  bool applied = false;
  int o_pe = 0, o_tc = 0, o_ve = 0, o_rs = 0;
  if (do_act) {
    int npe = pe, nve = ve;
    bool ok = true;
    if (ty == 1) { if (q > ve) ok = false; else { npe = pe + q; nve = ve - q; } }
    else { if (q > (pe < rs ? pe : rs)) ok = false; else { npe = pe - q; nve = ve + q; } }
    if (!ok) status |= 1;
    else {
      const int nrs = rs - (nve - ve);
      applied = true; o_pe = npe; o_tc = f_bits((float)((double)bits_f(tc) + (double)q)); o_ve = nve; o_rs = nrs;
      GP(PA_EMPTY, ap) = npe;
      GP(PA_TRANSFER_COST, ap) = o_tc;
      GV(VA_EMPTY, av) = nve;
      GV(VA_REMAINING_SPACE, av) = nrs;
      if (c >= 0) g_live[KD(f_plans) + c] = pl + period;
      else status |= 32;  // MRX_ENV_OFFROUTE_ACTION
      opnum += q;
      if (ap == lp2) pe2 = npe;
      if (av == v2) { ve2 = nve; rs2 = nrs; }
    }
  }
#undef GP
#undef GV
  if constexpr (OBS) {  // see fast_step: the ports block of this tick is patched, the deciding vessel's row rewritten
    if (applied) {
      if (OD(i_empty) >= 0) O.ports[((size_t)env * P + ap) * OD(np) + OD(i_empty)] = (double)o_pe;
      if (OD(i_tc) >= 0) O.ports[((size_t)env * P + ap) * OD(np) + OD(i_tc)] = (double)bits_f(o_tc);
    }
#pragma unroll
    for (int a = 0; a < 8; a++) {
      if (a < OD(nv)) {
        int raw = vo[a];
        if (applied && av == v2) { if (ODA(va, a) == VA_EMPTY) raw = o_ve; else if (ODA(va, a) == VA_REMAINING_SPACE) raw = o_rs; }
        O.vessel[(size_t)env * OD(nv) + a] = (double)raw;
      }
    }
  }
  dec_out[0] = t; dec_out[1] = lp2; dec_out[2] = v2;
  dec_out[3] = pe2 < rs2 ? pe2 : rs2;
  dec_out[4] = ve2; dec_out[5] = ed2;
  dec_out[6] = (t - KD(start_tick)) / KD(resolution); dec_out[7] = 1;
  if (O.agent_mode) agent_answer(K, O, env, agent_seed, t, lp2, v2, pe2 < rs2 ? pe2 : rs2, ve2);
  met_out[0] = acc_b; met_out[1] = acc_s; met_out[2] = opnum;
  *done_out = 0;
  g_priv[PH_PEND_LO] = (int32_t)(uint32_t)(pend_after & 0xffffffffull);
  g_priv[PH_PEND_HI] = (int32_t)(uint32_t)(pend_after >> 32);
  g_priv[PH_CUR_VESSEL] = v2;
  g_priv[PH_OPNUM_LO] = (int32_t)(uint32_t)((unsigned long long)opnum & 0xffffffffull);
  g_priv[PH_OPNUM_HI] = (int32_t)(opnum >> 32);
  K.hint[env] = (pend_after & ~(1ull << v2)) ? 0 : ((flags & FL_LONG) ? 2 : 1);
  if (status) wave::global_or(&K.status[env], status);
  return true;
}

// ==========================================================================================
// STEP: Env.step(action)
//   decision[8] = (tick, port, vessel, scope.load, scope.discharge, early_discharge, frame_index, valid)
// Decision events consumed by a step (core.py:349-366): Sequential answers the current vessel's event; Joint finishes
// every pending event of the tick (answered or not); JointWithSequentialAction the first n_answered in event order.
MRX_DEV uint64_t consume_decisions(const CimParams& K, uint64_t pend, int cur, int n_answered) {
  if (KD(decision_mode) == 0) return pend & ~(1ull << (cur & 63));
  if (KD(decision_mode) == 1 || n_answered < 0) return 0ull;
  for (int k = 0; k < n_answered && pend; k++) pend &= pend - 1;
  return pend;
}

// one env's arguments / outputs of a step
struct StepIO {
  const int32_t* actions;  // this env's action rows [max_actions][4]; nullptr: none
  int n_act, n_answered;
  int32_t* dec_out;
  long long* met_out;
  uint8_t* done_out;
};
// the batch-level arguments of one step launch (kernel arguments)
struct StepBatch {
  const int32_t *actions, *n_actions, *n_answered;
  int32_t* decisions;
  long long* metrics;
  uint8_t* done;
};
MRX_DEV StepIO step_io(const CimParams& K, const StepBatch& B, int env, bool counts = true) {
  StepIO io;
  io.actions = B.actions ? B.actions + (size_t)env * KD(max_actions) * 4 : nullptr;
  io.n_act = (counts && B.actions && B.n_actions) ? B.n_actions[env] : 0;
  io.n_answered = (counts && B.n_answered) ? B.n_answered[env] : -1;
  io.dec_out = B.decisions + (size_t)env * (KD(decision_mode) ? (size_t)KD(V) * 8 : 8);  // Joint modes: one row per vessel
  io.met_out = B.metrics + (size_t)env * 3;
  io.done_out = B.done + env;
  return io;
}
struct StepEnd { bool store, ord_dirty, buf_dirty; };

// HBM -> LDS of one env's state by LDS-DMA (complete with wave::lds_dma_wait()), and the write-back.  The RNG state
// arrays only go back when a twist regenerated them (a plain draw moves the cursor, which is a private-header word).
template <bool PG>
MRX_DEV void state_load_async(const CimParams& K, Lds& L, int env) {
  const uint32_t* g_mt = K.mt + (size_t)env * MTS_COUNT * MT_WORDS;
  copy_in_async(L.frame, K.live + (size_t)env * KD(FW), KD(FW));
  copy_in_async(L.priv, K.priv + (size_t)env * KD(PW), KD(PWH));
#ifndef MRX_LEAN
  copy_in_async(L.rfull, K.priv + (size_t)env * KD(PW) + KD(pv_rfull), KD(PW) - KD(pv_rfull));
#endif
  // The RNG states are requested unconditionally: whether a tick will run is only known once the private
  // state has arrived, and a second LDS-DMA round trip would serialise behind every later LDS access.
  if (!PG && KD(use_order_rng)) copy_in_async((int32_t*)L.mt_ord, (const int32_t*)(g_mt + MTS_ORDER * MT_WORDS), MT_WORDS);
  if (KD(use_buffer_rng)) copy_in_async((int32_t*)L.mt_buf, (const int32_t*)(g_mt + MTS_BUFFER * MT_WORDS), MT_WORDS);
}
template <bool PG>
MRX_DEV void state_store(const CimParams& K, Lds& L, int env, const StepEnd& e) {
  uint32_t* g_mt = K.mt + (size_t)env * MTS_COUNT * MT_WORDS;
  copy_words(K.live + (size_t)env * KD(FW), L.frame, KD(FW));
  copy_words(K.priv + (size_t)env * KD(PW), L.priv, KD(PWH));
#ifndef MRX_LEAN
  copy_words(K.priv + (size_t)env * KD(PW) + KD(pv_rfull), L.rfull, KD(PW) - KD(pv_rfull));
#endif
  if (!PG && KD(use_order_rng) && e.ord_dirty) copy_words((int32_t*)(g_mt + MTS_ORDER * MT_WORDS), (const int32_t*)L.mt_ord, MT_WORDS);
  if (KD(use_buffer_rng) && e.buf_dirty) copy_words((int32_t*)(g_mt + MTS_BUFFER * MT_WORDS), (const int32_t*)L.mt_buf, MT_WORDS);
}

// lean builds: the pending full returns of "my" pairs, HBM <-> registers (plain 4-byte accesses, lane = pair: coalesced rows)
MRX_DEV void ring_load(const CimParams& K, int env, RingRegs& rr) {
#ifdef MRX_LEAN
  const int lane = wave::lane();
  const int32_t* g = K.priv + (size_t)env * KD(PW) + KD(pv_rfull);
#pragma unroll
  for (int h = 0; h < RR_H; h++)
#pragma unroll
    for (int b = 0; b < RR_NB; b++) {
      const int k = b * 64 + lane;
      const int x = g[h * KD(NT) + (k < KD(NT) ? k : 0)];   // branch-free: clamped address, masked value
      rr.rf[h][b] = k < KD(NT) ? x : 0;
    }
#endif
}
MRX_DEV void ring_store(const CimParams& K, int env, const RingRegs& rr) {
#ifdef MRX_LEAN
  const int lane = wave::lane();
  int32_t* g = K.priv + (size_t)env * KD(PW) + KD(pv_rfull);
#pragma unroll
  for (int h = 0; h < RR_H; h++)
#pragma unroll
    for (int b = 0; b < RR_NB; b++) {
      const int k = b * 64 + lane;
      if (k < KD(NT)) g[h * KD(NT) + k] = rr.rf[h][b];
    }
#endif
}

// ------------------------------------------------------------------------------------------
// The FULL PATH of a step on state that is already staged in LDS (frame, private state, RNG states, tables), as STAGES,
// so that a pipelined caller can place its own memory traffic between them (gfx950 has ONE in-order vmcnt for loads and
// stores: a wait for a load also waits for every store issued before it, so loads must be issued BEFORE store batches
// and outputs are held in registers until a point where nothing will be waited for soon):
//   body_open  header out of LDS; the first tick's inputs are requested (global loads)
//   body_act   the action(s) of the pending decision (LDS only)
//   [caller]   tick_prefetch_land(c.pf)
//   body_run   ticks until the next decision event or the end of the episode; outputs -> registers (StepOut);
//              updated private header -> LDS.  The caller moves the state back to HBM (StepEnd::store).
//   body_emit  the outputs -> HBM
struct StepCtx {
  int t;
  bool fresh;
  uint64_t pend;
  int idx_ord, idx_buf;
  long long opnum;
  int status;
  int a0v, a0p, a0q, a0t, n_act, n_answered;
  TickPf pf;
};
struct StepOut {
  int kind;            // 0 nothing, 1 a regular step, 2 the episode was already over: (None, None, True)
  int dec[8];          // Sequential mode: the decision row (lane 0)
  long long met[3];
  int done, hint, tick, status;
  bool obs_on;         // fused observation written by this step
  double obs[8];       // elements lane + 64 j of the [P][np] block
  double vobs[8];      // the deciding vessel's row (lane 0)
};

// plan cell of (vessel v, port p) out of the staged route tables: v_cbase[v] + route_cidx of p's position on the route;
// -1 if p is not on the vessel's route (== CimParams::cidx_dense[v * P + p], without the trip to global memory)
MRX_DEV int plan_cell(const Lds& L, int v, int p) {
  const Tabs& T = L.tab;
  const int lane = wave::lane();
  const int Lr = T.v_route_len[v], rb = T.v_route_base[v];
  const bool hit = lane < Lr && (int)T.route_port[rb + (lane < Lr ? lane : 0)] == p;
  const uint64_t m = wave::ballot(hit);
  if (!m) return -1;
  return (int)T.v_cbase[v] + (int)T.route_cidx[rb + __builtin_ctzll(m)];
}

template <bool PG>
MRX_DEV bool body_open(const CimParams& K, int env, Lds& L, StepCtx& c, StepOut& out, Lean& Z) {
#ifdef MRX_LEAN2
  vrows_load(K, L, Z.w);  // (issued before the header words are waited for: one round trip for both)
#endif
  const int flags0 = U(L.priv[PH_FLAGS]);
  out.kind = 0;
  out.obs_on = false;
  c.pf = TickPf{};
  if (flags0 & FL_FINISHED) {  // reference: (None, None, True) once the generator is exhausted
    out.kind = 2;
    return false;
  }
  if (c.n_act > KD(max_actions)) c.n_act = KD(max_actions);
  c.t = L.priv[PH_TICK];
  c.fresh = (flags0 & FL_FRESH) != 0;
  c.pend = ((uint64_t)(uint32_t)L.priv[PH_PEND_HI] << 32) | (uint32_t)L.priv[PH_PEND_LO];
  c.idx_ord = L.priv[PH_IDX_ORDER];
  c.idx_buf = L.priv[PH_IDX_BUFFER];
  c.opnum = ((long long)L.priv[PH_OPNUM_HI] << 32) | (uint32_t)L.priv[PH_OPNUM_LO];
  c.status = 0;
  // A tick will run in this step iff no other vessel of the current tick is still waiting for its decision.
  // Its inputs (order row / order count, arrival records, noise tables) are requested NOW, so that
  // second memory round trip overlaps with the action handling.
  const uint64_t pend_after = c.fresh ? 0ull : consume_decisions(K, c.pend, L.priv[PH_CUR_VESSEL], c.n_answered);
  const int tn = c.fresh ? c.t : c.t + 1;
  if (!pend_after && tn < KD(T)) {
#ifdef MRX_LEAN2
    tick_prefetch_lean(K, env, L, tn, c.pf, Z);
#else
    if constexpr (!PG) tick_prefetch_static(K, c.pf, true);
    tick_prefetch<PG>(K, env, L, tn, c.pf);
#endif
  }
  return true;
}

// actions for the pending decision (core.py:301-315 -> business_engine.py:708-748)
MRX_DEV void body_act(const CimParams& K, Lds& L, const int32_t* actions, StepCtx& c, const Lean& Z) {
  const int P = KD(P), V = KD(V);
  if (c.fresh) return;
  for (int i = 0; i < c.n_act; i++) {  // wave-uniform
    const int32_t* a = actions + 4 * i;
    const int v = U(i == 0 ? c.a0v : a[0]), p = U(i == 0 ? c.a0p : a[1]), q = U(i == 0 ? c.a0q : a[2]), ty = U(i == 0 ? c.a0t : a[3]);
    if (v < 0 || v >= V || p < 0 || p >= P || q < 0 || (ty != 0 && ty != 1)) { c.status |= 1; continue; }
    // every word this action reads, requested back to back (one LDS round trip instead of one per word)
#ifdef MRX_LEAN2
    // ... including the vessel's route (lane = route position: port and compact matrix column), so that the plan cell of (v, p)
    // needs no lookup chain of its own
    const Tabs& T = L.tab;
    const int a_vs0 = wave::readlane(Z.s.vs0, v), a_Lr = a_vs0 & 63, a_rb = (int)((unsigned)a_vs0 >> 12), a_cb = wave::readlane(Z.s.vs2, v);
    const int a_lr = wave::lane() < a_Lr ? wave::lane() : 0;
    const int a_rp = T.route_port[a_rb + a_lr], a_rc = T.route_cidx[a_rb + a_lr], a_per = V_PERIOD(v);
#endif
    const int pe_v = FP(PA_EMPTY, p), ve_v = FV(VA_EMPTY, v), rs_v = FV(VA_REMAINING_SPACE, v), tc_v = FP(PA_TRANSFER_COST, p);
    const int pe = U(pe_v), ve = U(ve_v), rs = U(rs_v), tc = U(tc_v);
    int npe, nve;
    if (ty == 1) {  // DISCHARGE
      if (q > ve) { c.status |= 1; continue; }
      npe = pe + q; nve = ve - q;
    } else {
      if (q > (pe < rs ? pe : rs)) { c.status |= 1; continue; }
      npe = pe - q; nve = ve + q;
    }
    FP(PA_EMPTY, p) = npe;
    FV(VA_EMPTY, v) = nve;
    FV(VA_REMAINING_SPACE, v) = rs - (nve - ve);  // total_space - full - empty
    c.opnum += q;
    FP(PA_TRANSFER_COST, p) = f_bits((float)((double)bits_f(tc) + (double)q));
    {  // vessel_plans[v, p] += period (:748): the compact plan cell of (v, p), -1 if p is not on the vessel's route
#ifdef MRX_LEAN2
      const uint64_t hit = wave::ballot(wave::lane() < a_Lr && a_rp == p);
      const int cc = hit ? a_cb + wave::readlane(a_rc, __builtin_ctzll(hit)) : -1;
      if (cc >= 0) { const int pl = U(L.frame[KD(f_plans) + cc]); L.frame[KD(f_plans) + cc] = pl + U(a_per); }
#else
      const int cc = plan_cell(L, v, p);
      if (cc >= 0) { const int pl = U(L.frame[KD(f_plans) + cc]); L.frame[KD(f_plans) + cc] = pl + U(V_PERIOD(v)); }
#endif
      else c.status |= 32;  // MRX_ENV_OFFROUTE_ACTION
    }
  }
  c.pend = consume_decisions(K, c.pend, L.priv[PH_CUR_VESSEL], c.n_answered);
  wave::sync();
}

template <bool PG, bool OBS>
MRX_DEV StepEnd body_run(const CimParams& K, const CimObs& O, int env, Lds& L, const StepIO& io, StepCtx& c, StepOut& out, Prof& prof, RingRegs& rr, Lean& Z) {
  const int lane = wave::lane();
  const int P = KD(P), V = KD(V);
  StepEnd end = {true, false, false};
  int t = c.t;
  bool fresh = c.fresh, finished = false;
  uint64_t pend = c.pend;
  int dec_v = -1;
  // ---- advance to the next decision event or the end of the episode (core.py:329-381)
  for (;;) {
    if (pend) {
      dec_v = __builtin_ctzll(pend);
      break;
    }
    if (!fresh) {
      // C. post_step (business_engine.py:201-224)
      if ((t + 1) % KD(resolution) == 0) {
        if (lane < P) FP(PA_ACC_FULFILLMENT, lane) = FP(PA_ACC_BOOKING, lane) - FP(PA_ACC_SHORTAGE, lane);
        wave::sync();
        take_snapshot(K, O, env, L, (t - KD(start_tick)) / KD(resolution));
        wave::sync();
        if (lane < P) { FP(PA_SHORTAGE, lane) = 0; FP(PA_BOOKING, lane) = 0; FP(PA_FULFILLMENT, lane) = 0; FP(PA_TRANSFER_COST, lane) = 0; }
        wave::sync();
      }
      if (t + 1 == KD(T)) {
        if ((t + 1) % KD(resolution) != 0) take_snapshot(K, O, env, L, (t - KD(start_tick)) / KD(resolution));  // core.py:376-378
        finished = true;
        break;
      }
      t += 1;
    }
    prof.mark(PF_POST_STEP);
    fresh = false;
#ifdef MRX_LEAN2
    pend = run_tick_lean(K, env, L, t, c.pf, c.idx_buf, c.status, prof, end.buf_dirty, rr, Z);
#else
    pend = run_tick<PG>(K, env, L, t, c.pf, c.idx_ord, c.idx_buf, c.status, prof, end.ord_dirty, end.buf_dirty, rr);
#endif
    if (!pend && t + 1 < KD(T)) {  // another tick follows: its inputs, landed before that tick's snapshot stores are issued
#ifdef MRX_LEAN2
      tick_prefetch_lean(K, env, L, t + 1, c.pf, Z);
#else
      tick_prefetch<PG>(K, env, L, t + 1, c.pf);
#endif
      tick_prefetch_land(c.pf);
    }
  }

  if (MRX_UNALIGNED_FRAMES && !finished) {  // the pre-decision snapshot of core.py:345, where it is not overwritten later
    const int fi0 = (t - KD(start_tick)) / KD(resolution), lo = KD(start_tick) + fi0 * KD(resolution);
    if (t > lo + (KD(resolution) - 1 - lo % KD(resolution))) {  // the frame's post_step snapshot (tick + 1) % resolution == 0 is already behind
      wave::sync();
      take_snapshot(K, O, env, L, fi0);
      wave::sync();
    }
  }
  // ---- outputs (into registers; body_emit stores them)
  long long acc_b = 0, acc_s = 0;
  if (lane < P) { acc_b = FP(PA_ACC_BOOKING, lane); acc_s = FP(PA_ACC_SHORTAGE, lane); }
  acc_b = wave::reduce_add(acc_b);
  acc_s = wave::reduce_add(acc_s);
  const int fi = (t - KD(start_tick)) / KD(resolution);
  out.kind = 1;
  if (KD(decision_mode) != 0) {
    // Joint modes: one row per pending decision event, in event (= vessel) order; dec_out is [V][8] (stored right here)
    int32_t* dec_out = io.dec_out;
    const uint64_t pm = finished ? 0ull : pend;
    const bool mine = lane < V && ((pm >> lane) & 1ull);
    if (mine) {
      const int r = __builtin_popcountll(pm & ((1ull << lane) - 1ull));
      const int v = lane, p = FV(VA_LOC_PORT_IDX, v);
      const int pe = FP(PA_EMPTY, p), rs = FV(VA_REMAINING_SPACE, v);
      int32_t* d = dec_out + 8 * r;
      d[0] = t; d[1] = p; d[2] = v; d[3] = pe < rs ? pe : rs; d[4] = FV(VA_EMPTY, v); d[5] = FV(VA_EARLY_DISCHARGE, v);
      d[6] = fi; d[7] = 1;
    }
    const int cnt = __builtin_popcountll(pm);
    if (lane < V && lane >= cnt) {
      int32_t* d = dec_out + 8 * lane;
      for (int j = 0; j < 8; j++) d[j] = 0;
      if (lane == 0) { d[0] = t; d[6] = fi; }
    }
  } else if (!finished) {
    // The pre-decision snapshot (core.py:345) is ALIASED, not copied: while an env is paused, frame index
    // fi(t) of its snapshot list is its live frame (mrx_cim_query resolves it); see DESIGN.md.
    {
      const int v = dec_v, p = FV(VA_LOC_PORT_IDX, v);
      const int pe = FP(PA_EMPTY, p), rs = FV(VA_REMAINING_SPACE, v);
      out.dec[0] = t; out.dec[1] = p; out.dec[2] = v;
      out.dec[3] = pe < rs ? pe : rs;  // action_scope :247-260
      out.dec[4] = FV(VA_EMPTY, v);
      out.dec[5] = FV(VA_EARLY_DISCHARGE, v);
      out.dec[6] = fi; out.dec[7] = 1;
    }
    // fused observation: the decision's frame is the live frame (aliased pre-decision snapshot); consecutive lanes
    // hold consecutive doubles of the [P][np] block
    if constexpr (OBS) {
      out.obs_on = true;
      const int tot = P * OD(np), npd = OD(np) > 0 ? OD(np) : 1;  // (vessel-only observation: tot = 0, no division by a constant 0)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int i = lane + 64 * j;
        out.obs[j] = 0.0;
        if (i < tot) {
          const int p = i / npd, a = i - p * npd;
          const int attr = (int)((OD(pa_packed) >> (4 * a)) & 15u);
          out.obs[j] = port_attr_value(attr, FP(attr, p));
        }
      }
#pragma unroll
      for (int a = 0; a < 8; a++) out.vobs[a] = a < OD(nv) ? (double)FV(ODA(va, a), dec_v) : 0.0;
    }
  } else {
    out.dec[0] = t; out.dec[1] = 0; out.dec[2] = 0; out.dec[3] = 0; out.dec[4] = 0; out.dec[5] = 0;
    out.dec[6] = fi; out.dec[7] = 0;
  }
  out.met[0] = acc_b; out.met[1] = acc_s; out.met[2] = c.opnum;  // get_metrics :270-282
  out.done = finished ? 1 : 0;
  out.tick = t;
  out.status = c.status;
  // the coming step of this env: fast path iff (Sequential mode and) answering dec_v leaves another decision of the
  // tick pending, or the episode is over (the fast path reports "finished"); else a tick will run -> full path
  out.hint = (KD(decision_mode) == 0 && !MRX_UNALIGNED_FRAMES && (finished || (pend & ~(1ull << (dec_v & 63))))) ? 0 : 1;
  // ... and how long that full-path step will be: it runs to the next arrival (every arrival raises a decision), which the vessel
  // schedule already holds — a sailing vessel's next event is its arrival, a parked one's next arrival is cached in V_NEXT.  Steps
  // of two ticks or more (one in nine on global_trade.22p) are put at the head of the launch (hint 2; mrx_k_cim_schedule), and the
  // flag travels in the header so the fast-path steps in between can hand it on.  Scheduling only: results do not depend on it.
  bool long_next = false;
  if (KD(start_tick) == 0 && !finished) {
#ifdef MRX_LEAN2
    const bool soon = lane < V && (vr_park(Z.w) ? Z.w.next : Z.w.evt) <= t + 1;
#else
    const bool soon = lane < V && (FV(VA_IS_PARKING, lane) ? V_NEXT(lane) : V_EVT(lane)) <= t + 1;
#endif
    long_next = wave::ballot(soon) == 0ull;
  }
  if (long_next && out.hint) out.hint = 2;
  if (lane == 0) {
    L.priv[PH_TICK] = t;
    L.priv[PH_FLAGS] = finished ? FL_FINISHED : (long_next ? FL_LONG : 0);
    L.priv[PH_PEND_LO] = (int32_t)(uint32_t)(pend & 0xffffffffull);
    L.priv[PH_PEND_HI] = (int32_t)(uint32_t)(pend >> 32);
    L.priv[PH_CUR_VESSEL] = dec_v;
    L.priv[PH_OPNUM_LO] = (int32_t)(uint32_t)((unsigned long long)c.opnum & 0xffffffffull);
    L.priv[PH_OPNUM_HI] = (int32_t)(c.opnum >> 32);
    L.priv[PH_IDX_ORDER] = c.idx_ord;
    L.priv[PH_IDX_BUFFER] = c.idx_buf;
    L.priv[PH_ACCB_LO] = (int32_t)(uint32_t)((unsigned long long)acc_b & 0xffffffffull); L.priv[PH_ACCB_HI] = (int32_t)(acc_b >> 32);
    L.priv[PH_ACCS_LO] = (int32_t)(uint32_t)((unsigned long long)acc_s & 0xffffffffull); L.priv[PH_ACCS_HI] = (int32_t)(acc_s >> 32);
  }
  wave::sync();
  prof.mark(PF_OUTPUT);
  return end;
}

template <bool OBS>
MRX_DEV void body_emit(const CimParams& K, const CimObs& O, int env, const StepIO& io, const StepOut& out, long long agent_seed) {
  const int lane = wave::lane();
  if (out.kind == 0) return;
  if (out.kind == 2) {  // the episode was over before this step: (None, None, True)
    if (lane < 8) io.dec_out[lane] = lane == 7 ? -1 : 0;
    if (lane < 3) io.met_out[lane] = 0;
    if (lane == 0) { *io.done_out = 1; if (O.agent_mode) agent_none(O, env); }
    return;
  }
  if (O.agent_mode && KD(decision_mode) == 0 && lane == 0) {   // the device agent answers the decision this step has raised
    if (out.dec[7] == 1 && !out.done) agent_answer(K, O, env, agent_seed, out.dec[0], out.dec[1], out.dec[2], out.dec[3], out.dec[4]);
    else agent_none(O, env);
  }
  if (KD(decision_mode) == 0) {
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 8; j++) io.dec_out[j] = out.dec[j];
    }
    if constexpr (OBS) {
      if (out.obs_on) {
        const int tot = KD(P) * OD(np);
        double* op = O.ports + (size_t)env * tot;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int i = lane + 64 * j;
          if (i < tot) op[i] = out.obs[j];
        }
#pragma unroll
        for (int a = 0; a < 8; a++)
          if (a < OD(nv) && lane == 0) O.vessel[(size_t)env * OD(nv) + a] = out.vobs[a];
      }
    }
  }
  if (lane == 0) {
    io.met_out[0] = out.met[0]; io.met_out[1] = out.met[1]; io.met_out[2] = out.met[2];
    *io.done_out = (uint8_t)out.done;
    K.tick[env] = out.tick;
    K.hint[env] = (uint8_t)out.hint;
    if (out.status) wave::global_or(&K.status[env], out.status);  // fire-and-forget: a read-modify-write would wait for every store in flight
  }
}

// the stages back to back (one env per workgroup kernels)
template <bool PG, bool OBS>
MRX_DEV StepEnd full_body(const CimParams& K, const CimObs& O, int env, Lds& L, const StepIO& io, int a0v, int a0p, int a0q, int a0t, Prof& prof, RingRegs& rr, Lean& Z) {
  StepCtx c;
  StepOut out;
  c.a0v = a0v; c.a0p = a0p; c.a0q = a0q; c.a0t = a0t;
  c.n_act = io.n_act; c.n_answered = io.n_answered;
  StepEnd end = {false, false, false};
  const long long agent_seed = O.agent_mode ? K.seed[env] : 0;   // (requested before the step's work, used by body_emit)
  if (body_open<PG>(K, env, L, c, out, Z)) {
    prof.mark(PF_LOAD);
    body_act(K, L, io.actions, c, Z);
    prof.mark(PF_ACTION);
    tick_prefetch_land(c.pf);  // before the snapshot stores of post_step, so that nothing later waits behind them
    prof.mark(PF_MT_LOAD);
    end = body_run<PG, OBS>(K, O, env, L, io, c, out, prof, rr, Z);
  }
  body_emit<OBS>(K, O, env, io, out, agent_seed);
  return end;
}

// which path a step of `env` takes
enum { PATH_PROBE = 0,  // unsorted launch: read the hint with the fast rows, then decide
       PATH_FAST = 1,   // sorted launch, fast-hinted: fast path, full path as the fallback
       PATH_FULL = 2 }; // sorted launch, full-hinted: no header round trip, straight to the state transfer

// One env per workgroup (one wave): fast path out of HBM, or LDS-DMA in -> full_body -> write-back.
// `lds`: this env's LDS block (l_ctab words); `lds_ctab`: where the topology tables are staged — behind the env's block, or
// (multi-wave workgroups of the specialised build) one copy behind the blocks of all the workgroup's envs.  Every full-path
// wave issues the copy itself and waits for its own transfer: waves of a workgroup write identical bytes, so they need no barrier.
template <bool PG, bool OBS>
MRX_DEV void step_env(const CimParams& K, const CimObs& O, int env, int32_t* lds, const StepIO& io, int path, int32_t* lds_ctab = nullptr) {
  Lds L = make_lds(K, lds);
  Prof prof;
  // The 64-byte private header and the first action decide which path this step takes.
  if (KD(decision_mode) == 0 && path != PATH_FULL) {  // Sequential
    FastRows rows;
    fast_rows_request<OBS>(K, O, env, rows);
    int hint = 0;
    if (path == PATH_PROBE) hint = K.hint[env];
    int f0v = 0, f0p = 0, f0q = 0, f0t = 0;
    if (io.actions) { f0v = io.actions[0]; f0p = io.actions[1]; f0q = io.actions[2]; f0t = io.actions[3]; }
    if (!U(hint) && fast_step<OBS>(K, O, env, rows, io.n_act > KD(max_actions) ? KD(max_actions) : io.n_act, f0v, f0p, f0q, f0t, io.dec_out,
                                   io.met_out, io.done_out)) {
#ifdef MRX_PROFILE_FAST_PATH  // tools build: fast-path cycles and count (its atomics perturb the full-path numbers)
      prof.mark(13); prof.mark(14, 1);
      prof.flush();
#endif
      return;
    }
    prof.mark(12);  // header round trip of the full path
  }
  // Everything this step needs from HBM is requested before the first wait: hot frame, private state and
  // topology tables by LDS-DMA, the action words into registers.
  state_load_async<PG>(K, L, env);
  stage_tables(K, L, lds_ctab ? lds_ctab : lds + KD(l_ctab));
  RingRegs rr;
  ring_load(K, env, rr);
  Lean Z;
#ifdef MRX_LEAN2
  lean_stat_load(K, Z.s);
#endif
  int a0v = 0, a0p = 0, a0q = 0, a0t = 0;
  if (io.actions) { a0v = io.actions[0]; a0p = io.actions[1]; a0q = io.actions[2]; a0t = io.actions[3]; }
  wave::lds_dma_wait();
  const StepEnd e = full_body<PG, OBS>(K, O, env, L, io, a0v, a0p, a0q, a0t, prof, rr, Z);
  if (e.store) { state_store<PG>(K, L, env, e); ring_store(K, env, rr); }
  prof.mark(PF_STORE);
  prof.flush();
}

// ==========================================================================================
// SPLIT STEP (launch form 4): the fast-hinted envs and the full-path envs of a step go to two kernels.
//  * fast_lanes_env: one env per LANE, no LDS at all (fast_step_lane) — a few dozen workgroups for the whole batch instead
//    of one LDS-carrying workgroup per fast env queueing behind the full-path ones for an LDS slot.  An env the fast path
//    cannot handle (several actions, stale hint) is appended to the full-path list of the step.
//  * step_loop: as many workgroups as the device holds at once walk the full-path list (order[0 .. sched[0])), one env
//    per iteration through step_env.
template <bool OBS>
MRX_DEV void fast_lanes_env(const CimParams& K, const CimObs& O, const StepBatch& B, const uint8_t* mask, int env) {
  if (env >= K.n_envs || (mask && !mask[env]) || K.hint[env]) return;
  const StepIO io = step_io(K, B, env);
  const int n_act = io.n_act > KD(max_actions) ? KD(max_actions) : io.n_act;
  if (KD(decision_mode) == 0 && fast_step_lane<OBS>(K, O, env, io.actions, n_act, io.dec_out, io.met_out, io.done_out)) return;
  K.order[wave::global_add(&K.sched[0], 1)] = env | MRX_ORDER_TICK;   // the general path after all (launched next, in stream order)
}

template <bool PG, bool OBS>
MRX_DEV void step_loop(const CimParams& K, const CimObs& O, int32_t* lds, int w, int W, const StepBatch& B) {
  const int n_tick = U(wave::ld_uniform_v(K.sched + 0));
  for (int i = w; i < n_tick; i += W) {  // wave-uniform
    const int env = U(wave::ld_uniform_v(K.order + i)) & (MRX_ORDER_TICK - 1);
    step_env<PG, OBS>(K, O, env, lds, step_io(K, B, env), PATH_FULL);
    wave::sync();
  }
}

// ==========================================================================================
// QUERY: snapshot_list[node][ticks:nodes:attrs] -> float64 (np_backend.pyx:520-549)
MRX_DEV int attr_slots(const CimParams& K, int node_type, int a) {
  if (node_type == 0) return 1;
  if (node_type == 1) {
    if (a == VA_PAST_STOP_LIST || a == VA_PAST_STOP_TICK_LIST) return KD(past_n);
    if (a == VA_FUTURE_STOP_LIST || a == VA_FUTURE_STOP_TICK_LIST) return KD(future_n);
    return 1;
  }
  return a == MA_FULL_ON_PORTS ? KD(P) * KD(P) : KD(V) * KD(P);
}

MRX_DEV int frame_word(const CimParams& K, int node_type, int a, int node, int s) {
  if (node_type == 0) return KD(f_ports) + a * KD(P) + node;
  if (node_type == 1) return KD(f_vessels) + a * KD(V) + node;  // scalar attributes (a < VA_PAST_STOP_LIST); the lists: stop_list_value
  if (a == MA_FULL_ON_PORTS) {
    const int k = K.pair_dense[s];  // dense cell s = src * P + dst
    return k < 0 ? -1 : KD(f_fop) + k;
  }
  const int c = K.cidx_dense[s];  // dense cell s = vessel * P + port
  if (c < 0) return -1;          // port not on the vessel's route: constant cell
  return (a == MA_FULL_ON_VESSELS ? KD(f_fov) : KD(f_plans)) + c;
}

// Slot s of a vessel's stop-list attribute `a`, as of `frame` (a frame of `env`'s snapshot list or its live frame).
//   past lists (vessel_past_stops_wrapper.py:23-38; business_engine.py:634-656 appends the stop that is being left): after n
//   departures (n = next_loc_idx) the list holds stops n - past_n .. n - 1 of the vessel's unrolled stop table, -1 where that
//   index is negative: (port of route position (start + j) mod L, arrival tick of stop j).  Stop 0 is "arrived at" tick 0 by
//   construction of the episode (business_engine.py:371-379) — except for a vessel whose first departure fell before
//   start_tick (zombie_mask), which reads the table's own entry.
//   future lists (vessel_future_stops_prediction.py:49-85, refreshed at every arrival and at reset): the future_n stops after
//   stop k = last_loc_idx, predicted with the noise-free legs from that stop's arrival tick (0 for k = 0).
MRX_DEV int stop_list_value(const CimParams& K, int env, const int32_t* frame, int a, int v, int s) {
  const int Lr = K.v_route_len[v], rb = K.v_route_base[v], start = K.v_start[v];
  const uint32_t* srow = K.stops + ((size_t)env * KD(V) + v) * KD(SMAX);
  if (a == VA_PAST_STOP_LIST || a == VA_PAST_STOP_TICK_LIST) {
    const int j = frame[KD(f_vessels) + VA_NEXT_LOC_IDX * KD(V) + v] - KD(past_n) + s;
    if (j < 0) return -1;
    if (a == VA_PAST_STOP_LIST) return K.route_port[rb + (start + j) % Lr];
    if (j == 0) {
      const int32_t* hdr = K.priv + (size_t)env * KD(PW);
      const uint64_t zm = ((uint64_t)(uint32_t)hdr[PH_ZOMBIE_HI] << 32) | (uint32_t)hdr[PH_ZOMBIE_LO];
      if (!((zm >> v) & 1ull)) return 0;
    }
    return stop_arrival(srow[j < KD(SMAX) ? j : KD(SMAX) - 1]);
  }
  const int k = frame[KD(f_vessels) + VA_LAST_LOC_IDX * KD(V) + v];
  int tick = k == 0 ? 0 : stop_arrival(srow[k < KD(SMAX) ? k : KD(SMAX) - 1]);
  int x = (start + k) % Lr;
  const int lo = K.leg_off[v];
  for (int i = 0; i <= s; i++) {
    tick += K.leg_time[lo + x];
    x = (x + 1 == Lr) ? 0 : x + 1;
  }
  return a == VA_FUTURE_STOP_LIST ? K.route_port[rb + x] : tick;
}

// The frame a snapshot query for frame index `fi` of `env` reads, or nullptr when the ring does not hold it.
// While an env is paused at a decision its current frame index is aliased to the live frame (the
// reference's pre-decision take_snapshot, core.py:345), which also evicts whatever the slot held.
MRX_DEV const int32_t* frame_of(const CimParams& K, int env, int fi) {
  const int s = (fi < 0 ? 0 : fi) % KD(S);
  const int32_t* hdr = K.priv + (size_t)env * KD(PW);
  // three independent loads, then selects: no dependent chain of memory latencies
  const int flags = hdr[PH_FLAGS], tick = hdr[PH_TICK], held = K.ring_fi[(size_t)env * KD(S) + s];
  const bool paused = (flags & (FL_FRESH | FL_FINISHED)) == 0;
  const int cur_fi = (tick - KD(start_tick)) / KD(resolution);
  const int32_t* frame = nullptr;  // padding for missing frames :541-545
  if (held == fi) frame = K.ring + ((size_t)env * KD(S) + s) * KD(FW);
  if (paused && s == cur_fi % KD(S)) frame = fi == cur_fi ? K.live + (size_t)env * KD(FW) : nullptr;
  return fi < 0 ? nullptr : frame;
}

// one output element; `row` = env*nt*nn + ti*nn + ni, `col` in [0, row_slots)
MRX_DEV double query_elem(const CimParams& K, int node_type, const int32_t* ticks, int nt, int ticks_per_env,
                          const int32_t* nodes, int nn, int nodes_per_env, const int32_t* attrs, int na, long long row, int col) {
  const int ni = (int)(row % nn);
  const int ti = (int)((row / nn) % nt);
  const int env = (int)(row / ((long long)nn * nt));
  const int fi = ticks[(size_t)env * ticks_per_env + ti];  // ticks_per_env = row stride in int32 elements (0: one shared row)
  const int32_t* frame = frame_of(K, env, fi);
  if (!frame) return 0.0;
  int a = 0, slot = col;
  for (int i = 0; i < na; i++) {
    const int ns = attr_slots(K, node_type, attrs[i]);
    if (slot < ns) { a = attrs[i]; break; }
    slot -= ns;
  }
  const int node = nodes[(size_t)env * nodes_per_env + ni];  // nodes_per_env = row stride (0: one shared row)
  const int n_nodes = node_type == 0 ? KD(P) : node_type == 1 ? KD(V) : 1;
  if (node < 0 || node >= n_nodes) return 0.0;  // e.g. the -1 padding of stop lists used as a node index
  if (node_type == 1 && a >= VA_PAST_STOP_LIST) return (double)stop_list_value(K, env, frame, a, node, slot);
  const int w = frame_word(K, node_type, a, node, slot);
  if (w < 0) return a == MA_VESSEL_PLANS ? -1.0 : 0.0;  // never-written cells: plans are initialised to -1 (business_engine.py:344)
  const int32_t raw = frame[w];
  return (node_type == 0 && a == PA_TRANSFER_COST) ? (double)bits_f(raw) : (double)raw;
}

#undef FP
#undef FV
#undef FOPK
#undef FOVC
#undef PLANC
#undef V_EVT
#undef V_NEXT
#undef V_POS
#undef V_KRL
#undef V_PERIOD
#undef U
#undef RING_FULL
#undef RING_EMPTY

}  // namespace cim
