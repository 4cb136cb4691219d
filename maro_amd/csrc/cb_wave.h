// cb_wave.h — the WAVE-COOPERATIVE decision step of the citi_bike engine (one env per wave64).
//
// At the size the reference ships (ny.*: ~800 stations, the filter chain distance 80 -> requirements 40 -> trip_window 20 over
// 10 windows, topologies/ny.201801/config.yml:21-26) a decision tick raises 150-300 decision events per env, so more than 99 %
// of all env-steps never leave their tick: they apply one action, pick the next pending station and evaluate its action scope
// (decision_strategy.py:253-293) — ~4000 dependent compare/swap iterations of the per-lane selection sort in cb_device.h, on a
// batch that is only n_envs / 64 waves wide.  This file is that step written across the lanes of a wave instead:
//   * candidate neighbours live two per lane in registers (<= 128 after the leading distance filters),
//   * each filter RANKS them by counting (rank_i = number of better candidates: one broadcast per candidate, keys are distinct so
//     the order is total and equals the reference's stable sort) and compacts the survivors through 1.5 KB of LDS,
//   * the trip-window sums (decision_strategy.py:88-163) are one batch of independent loads per candidate, the per-frame row
//     offsets computed by one lane per frame,
//   * the pending-decision masks are one word per lane: "is there another decision in this tick" is a ballot.
// A batch of n envs is n waves (4096 envs fill the chip) instead of n / 64.  Whatever this step cannot do without replaying
// events — no further decision in the tick, a fresh / finished env, several actions, more than 128 candidates — it leaves
// UNTOUCHED and flags in `todo`; the general kernel (cb::step_env, one env per lane) then runs for exactly those envs.
// Same results by construction; pinned by the goldens and the batch-vs-oracle tests on both paths (tests/emu runs this file on
// the 64-fiber wave emulator).  Written against wave::{lane, sync, ballot, bcast}; state accesses go to HBM (SoA [word][env]).
#pragma once
#include <math.h>
#include <stdint.h>

#include "cb_params.h"

namespace cb {

#define W_HDR(w) K.hdr[CB_IX(CD(aos), CD(stride), CH_WORDS, (w), e)]
#define W_ST(a, s) K.live[CB_IX(CD(aos), CD(stride), CD(FW), ((size_t)(a) * CD(S) + (size_t)(s)), e)]
#define W_DMK(i) K.decmask[CB_IX(CD(aos), CD(stride), (2 * CD(mask_words)), (i), e)]
#define W_BKT(i) K.bkt[CB_IX(CD(aos), CD(stride), (2 * CB_LAND_SLOTS + CB_LAND_SLOTS / 32), (i), e)]
#define W_POOL(i, w) K.pool[CB_IX(CD(aos), CD(stride), (CD(pool_cap) * CB_POOL_WORDS), ((size_t)(i) * CB_POOL_WORDS + (w)), e)]

// What a wave learns about env `e` before it commits to the in-tick step: nothing is written while this is filled in.
struct WavePre {
  int h;              // header word `lane` (lanes >= CH_WORDS: 0)
  uint32_t sup, dem;  // pending-decision mask words `lane`, the answered decision already popped
  int s0, t, s1, type;
};

// Does this Env.step of env `e` stay inside its tick, within what decision_step_wave does across the lanes?  Reads only; the
// same answer on every lane.  mrx_k_cb_classify asks it once per env and batch step, so that the in-tick kernel and the replay
// kernel can run side by side on disjoint envs; decision_step_wave asks it again for the values.
MRX_DEV bool decision_step_wave_pre(const CbParams& K, int e, int n_actions, WavePre& P) {
  const int lane = wave::lane();
  const int MW = CD(mask_words);
  // ---- header (lane w holds word w) and the pending-decision masks (lane w holds words w of both masks)
  P.h = lane < CH_WORDS ? W_HDR(lane) : 0;
  const int flags = wave::bcast(P.h, CH_FLAGS);
  if (!(flags & CFL_PENDING) || (flags & (CFL_FINISHED | CFL_STASH)) || n_actions > 1 || MW > 64) return false;
  P.sup = lane < MW ? W_DMK(lane) : 0u;
  P.dem = lane < MW ? W_DMK(MW + lane) : 0u;
  P.s0 = wave::bcast(P.h, CH_CUR_STATION);
  P.t = wave::bcast(P.h, CH_TICK);
  if (lane == (P.s0 >> 5)) { P.sup &= ~(1u << (P.s0 & 31)); P.dem &= ~(1u << (P.s0 & 31)); }  // pop the answered decision (apply_actions)
  const uint64_t pend = wave::ballot((P.sup | P.dem) != 0u);
  if (!pend) return false;  // last decision of the tick: events have to be replayed
  // ---- the next decision (next_decision: lowest station index, Supply wins) and its candidate list
  const int l1 = __builtin_ctzll(pend);
  const uint32_t w_any = (uint32_t)wave::bcast((int)(P.sup | P.dem), l1), w_sup = (uint32_t)wave::bcast((int)P.sup, l1);
  const int j1 = __builtin_ctz(w_any);
  P.s1 = l1 * 32 + j1;
  P.type = (w_sup >> j1 & 1u) ? MRX_CB_SUPPLY : MRX_CB_DEMAND;
  return scope_wave_ok(K, P.s1, P.t);
}

// Env.step of env `e` when the step stays inside its tick.  Returns true when the env was handled (outputs written); false:
// nothing was touched, the general path must run.
MRX_DEV bool decision_step_wave(const CbParams& K, int e, const int32_t* actions, int n_actions, int32_t* dec, int32_t* out, int64_t* met,
                                uint8_t* done, int32_t* scr) {
  const int lane = wave::lane();
  const int S = CD(S), MW = CD(mask_words);
  WavePre P;
  WProf WP;
  if (!decision_step_wave_pre(K, e, n_actions, P)) return false;
  WP.count(K, e, 13);
  WP.mark(K, e, 8);
  const int h = P.h, s0 = P.s0, t = P.t, s1 = P.s1, type = P.type;
  const uint32_t sup = P.sup, dem = P.dem;
  const int fi_cur = (t - CD(start_tick)) / CD(res);
  // ---- from here on the env is handled.  The action (_on_action_received :521-559): one action at most
  int status = 0, tt_pos = wave::bcast(h, CH_TT_POS), tail = wave::bcast(h, CH_POOL_TAIL), minland = wave::bcast(h, CH_POOL_MINLAND),
      late = wave::bcast(h, CH_LATE);
  const int head = wave::bcast(h, CH_POOL_HEAD);
  int patch_s = -1, patch_b = 0, patch_min = 0;  // the one station whose bikes (and min_bikes) this step changed (the scope below must see the new values)
  if (n_actions == 1) {
    const int frm = actions[0], to = actions[1], number = actions[2];
    if (frm >= 0 && to >= 0) {
      if (frm >= S || to >= S) {
        status |= MRX_CB_ENV_INVALID_ACTION;
      } else {
        const int b = W_ST(LV_BIKES, frm), mb = W_ST(LV_MIN_BIKES, frm);
        wave::sync();  // every lane has read the station's bikes before lane 0 overwrites them below
        const int ex = b < number ? b : number;
        if (ex > 0) {
          patch_s = frm; patch_b = b - ex; patch_min = b - ex < mb ? b - ex : mb;
          if (lane == 0) {  // station.py:71-75
            W_ST(LV_BIKES, frm) = b - ex;
            if (b - ex < W_ST(LV_MIN_BIKES, frm)) W_ST(LV_MIN_BIKES, frm) = b - ex;
          }
          int tt = 1;
          if (tt_pos < CD(tt_cap)) tt = K.tt[CB_IX(CD(aos), CD(stride), CD(tt_cap), tt_pos, e)];
          else status |= MRX_CB_ENV_TRANSFER_TIMES_OUT;
          tt_pos++;
          if (!(tt < 0 || t + tt >= CD(max_tick))) {  // (a delivery into the past / after the episode never runs)
            if (tt == 0) late++;   // (another decision follows in the tick's list: the list tail is not stale, event_linked_list.py:86-108)
            if (tail - head >= CD(pool_cap) || tt >= CB_LAND_SLOTS) {
              status |= MRX_CB_ENV_DELIVERY_OVERFLOW;
            } else {
              const int idx = tail % CD(pool_cap);
              if (lane == 0) {  // pool_push: append, and link behind the last entry landing at the same tick
                W_POOL(idx, 0) = t + tt; W_POOL(idx, 1) = t; W_POOL(idx, 2) = frm; W_POOL(idx, 3) = to; W_POOL(idx, 4) = ex; W_POOL(idx, 5) = -1;
                const int slot = (t + tt) & (CB_LAND_SLOTS - 1);
                const int last = W_BKT(CB_LAND_SLOTS + slot);
                if (last >= 0) W_POOL(last, 5) = idx;
                else { W_BKT(slot) = idx; W_BKT(2 * CB_LAND_SLOTS + (slot >> 5)) |= (int32_t)(1u << (slot & 31)); }
                W_BKT(CB_LAND_SLOTS + slot) = idx;
              }
              tail++;
              if (t + tt < minland) minland = t + tt;
            }
          }
        }
      }
    }
  }
  // ---- state write-back: the two mask words that changed, the header words that changed
  if (lane == (s0 >> 5)) { W_DMK(lane) = sup; W_DMK(MW + lane) = dem; }
  {
    const int ndec = wave::bcast(h, CH_NDEC) + 1;
    const int st_old = wave::bcast(h, CH_STATUS);
    if (lane == 0) {
      W_HDR(CH_CUR_STATION) = s1; W_HDR(CH_CUR_TYPE) = type; W_HDR(CH_NDEC) = ndec; W_HDR(CH_TT_POS) = tt_pos; W_HDR(CH_POOL_TAIL) = tail;
      W_HDR(CH_POOL_MINLAND) = minland; W_HDR(CH_LATE) = late;
      if (status) W_HDR(CH_STATUS) = st_old | status;
    }
  }
  WP.mark(K, e, 9);
  // ---- action scope of (s1, type) at tick t (decision_strategy.py:253-293), across the lanes; the bikes this step just took
  // from `patch_s` come out of registers (lane 0's store above is not ordered against other lanes' loads)
  const int n_rows = scope_wave(K, s1, type, t, scr, out,
                                [&](int st) { return st == patch_s ? patch_b : W_ST(LV_BIKES, st); },
                                [&](int slot, int* fi, int* tk) { *fi = K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), slot, e)]; *tk = K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), slot, e)]; },
                                [&](int slot, int fi, int tk) { K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), slot, e)] = fi; K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), slot, e)] = tk; },
                                [&](int i, int st) {   // the fused observation's row i (mrx_cb_set_observation, scope rows)
                                  if (K.obs) write_scope_observation_row(K, e, i, st, t, [&](int lv, int s2) {
                                    return (s2 == patch_s && lv == LV_BIKES) ? patch_b : (s2 == patch_s && lv == LV_MIN_BIKES) ? patch_min : (int)W_ST(lv, s2); });
                                });
  WP.mark(K, e, 10);
  const int m0 = wave::bcast(h, CH_TRIPS), m1 = wave::bcast(h, CH_SHORT), m2 = wave::bcast(h, CH_OPER);
  if (lane == 0) {
    dec[0] = t; dec[1] = s1; dec[2] = type; dec[3] = fi_cur; dec[4] = n_rows; dec[5] = 1; dec[6] = 0; dec[7] = 0;
    met[0] = m0; met[1] = m1; met[2] = m2;
    *done = 0;
  }
  return true;
}

// mrx_cb_set_replay_period: a call that the general kernel does not follow.  The env keeps its state; if it stands at a decision the
// answer it was just given is put aside (CbParams::stash, CFL_STASH: the general step applies it when it runs), and its row says
// "no decision yet" — exactly what a step budget's unfinished env reports.  Later answers to that row are ignored, as documented.
MRX_DEV void defer_env_wave(const CbParams& K, int e, const int32_t* actions, int n_actions, int32_t* dec, int32_t* out, int64_t* met, uint8_t* done) {
  const int lane = wave::lane();
  const int h = lane < CH_WORDS ? W_HDR(lane) : 0;
  const int flags = wave::bcast(h, CH_FLAGS), t = wave::bcast(h, CH_TICK);
  const bool finished = (flags & CFL_FINISHED) != 0;
  if ((flags & CFL_PENDING) && !(flags & CFL_STASH) && !finished) {
    const int na = n_actions < 0 ? 0 : n_actions > CB_STASH_MAX ? CB_STASH_MAX : n_actions;
    if (lane < na * 3) K.stash[CB_IX(CD(aos), CD(stride), (CB_STASH_MAX * 3), lane, e)] = actions[lane];
    if (lane == 0) { W_HDR(CH_FLAGS) = flags | CFL_STASH; W_HDR(CH_RES1) = na; }
  }
  const int m0 = wave::bcast(h, CH_TRIPS), m1 = wave::bcast(h, CH_SHORT), m2 = wave::bcast(h, CH_OPER);
  if (lane == 0) {
    dec[0] = t; dec[1] = -1; dec[2] = -1; dec[3] = (t - CD(start_tick)) / CD(res); dec[4] = 0; dec[5] = 0; dec[6] = 0; dec[7] = 0;
    met[0] = m0; met[1] = m1; met[2] = m2;
    *done = finished ? 1 : 0;
  }
  for (int i = lane; i < CD(scope_cap); i += 64) {
    out[2 * i] = -1; out[2 * i + 1] = -1;
    if (K.obs) write_scope_observation_row(K, e, i, -1, t, [&](int, int) { return 0; });
  }
}

#undef W_HDR
#undef W_ST
#undef W_DMK
#undef W_POOL
#undef W_BKT

}  // namespace cb
