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

enum { CBW_MAX = 128 };  // candidates ranked in registers (two per lane)

#define W_HDR(w) K.hdr[(size_t)(w) * CD(stride) + e]
#define W_ST(a, s) K.live[((size_t)(a) * CD(S) + (size_t)(s)) * CD(stride) + e]
#define W_DMK(i) K.decmask[(size_t)(i) * CD(stride) + e]
#define W_POOL(i, w) K.pool[((size_t)(i) * CB_POOL_WORDS + (w)) * CD(stride) + e]

// rank[a] = how many of the n candidates come before candidate (a, lane) — mode 0 / 2: (v, key) descending, 1: ascending
MRX_DEV void cbw_rank(int n, int mode, const int* v, const int* key, int* rank) {
  rank[0] = rank[1] = 0;
  for (int j = 0; j < n; j++) {  // wave-uniform
    const int vj = wave::bcast(j < 64 ? v[0] : v[1], j & 63), kj = wave::bcast(j < 64 ? key[0] : key[1], j & 63);
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const bool before = mode == 1 ? (vj < v[a] || (vj == v[a] && kj < key[a])) : (vj > v[a] || (vj == v[a] && kj > key[a]));
      rank[a] += before ? 1 : 0;
    }
  }
}

// keep the n_out best of n candidates, best first: survivors move to position = rank through the LDS scratch
MRX_DEV void cbw_select(int32_t* scr, int n, int n_out, int mode, const int* v, int* key, int* val) {
  const int lane = wave::lane();
  int rank[2];
  cbw_rank(n, mode, v, key, rank);
#pragma unroll
  for (int a = 0; a < 2; a++)
    if (a * 64 + lane < n && rank[a] < n_out) { scr[rank[a]] = key[a]; scr[CBW_MAX + rank[a]] = val[a]; }
  wave::sync();
#pragma unroll
  for (int a = 0; a < 2; a++) {
    const int i = a * 64 + lane;
    key[a] = i < n_out ? scr[i] : -1;
    val[a] = i < n_out ? scr[CBW_MAX + i] : 0;
  }
  wave::sync();
}

// Env.step of env `e` when the step stays inside its tick.  Returns true when the env was handled (outputs written); false:
// nothing was touched, the general path must run.
MRX_DEV bool decision_step_wave(const CbParams& K, int e, const int32_t* actions, int n_actions, int32_t* dec, int32_t* out, int64_t* met,
                                uint8_t* done, int32_t* scr) {
  const int lane = wave::lane();
  const int S = CD(S), MW = CD(mask_words);
  // ---- header (lane w holds word w) and the pending-decision masks (lane w holds words w of both masks)
  const int h = lane < CH_WORDS ? W_HDR(lane) : 0;
  const int flags = wave::bcast(h, CH_FLAGS);
  if (!(flags & CFL_PENDING) || (flags & CFL_FINISHED) || n_actions > 1 || MW > 64) return false;
  uint32_t sup = lane < MW ? W_DMK(lane) : 0u, dem = lane < MW ? W_DMK(MW + lane) : 0u;
  const int s0 = wave::bcast(h, CH_CUR_STATION), t = wave::bcast(h, CH_TICK);
  if (lane == (s0 >> 5)) { sup &= ~(1u << (s0 & 31)); dem &= ~(1u << (s0 & 31)); }  // pop the answered decision (apply_actions)
  const uint64_t pend = wave::ballot((sup | dem) != 0u);
  if (!pend) return false;  // last decision of the tick: events have to be replayed
  // ---- the next decision (next_decision: lowest station index, Supply wins) and its candidate list
  const int l1 = __builtin_ctzll(pend);
  const uint32_t w_any = (uint32_t)wave::bcast((int)(sup | dem), l1), w_sup = (uint32_t)wave::bcast((int)sup, l1);
  const int j1 = __builtin_ctz(w_any), s1 = l1 * 32 + j1;
  const int type = (w_sup >> j1 & 1u) ? MRX_CB_SUPPLY : MRX_CB_DEMAND;
  int n = K.nb_cnt[s1], nf0 = 0;
  while (nf0 < CD(n_filters) && CDA(f_type, nf0) == MRX_CB_FILTER_DISTANCE) {  // leading distance filters: a prefix of the list
    n = CDA(f_num, nf0) < n ? CDA(f_num, nf0) : n;
    nf0++;
  }
  if (n > CBW_MAX) return false;
  const int fi_cur = (t - CD(start_tick)) / CD(res);
  int tw_count = 0;
  for (int f = nf0; f < CD(n_filters); f++)
    if (CDA(f_type, f) == MRX_CB_FILTER_TRIP_WINDOW) {
      const int avail = fi_cur + 1 < CD(ring_slots) ? fi_cur + 1 : CD(ring_slots);
      const int aw = CDA(f_win, f) < avail ? CDA(f_win, f) : avail;
      tw_count = aw > 0 ? aw : avail;
      if (tw_count > 64) return false;
    }
  // ---- from here on the env is handled.  The action (_on_action_received :521-559): one action at most
  int status = 0, tt_pos = wave::bcast(h, CH_TT_POS), tail = wave::bcast(h, CH_POOL_TAIL), minland = wave::bcast(h, CH_POOL_MINLAND),
      late = wave::bcast(h, CH_LATE);
  const int head = wave::bcast(h, CH_POOL_HEAD);
  int patch_s = -1, patch_b = 0;  // the one station whose bikes this step changed (the scope below must see the new value)
  if (n_actions == 1) {
    const int frm = actions[0], to = actions[1], number = actions[2];
    if (frm >= 0 && to >= 0) {
      if (frm >= S || to >= S) {
        status |= MRX_CB_ENV_INVALID_ACTION;
      } else {
        const int b = W_ST(LV_BIKES, frm);
        wave::sync();  // every lane has read the station's bikes before lane 0 overwrites them below
        const int ex = b < number ? b : number;
        if (ex > 0) {
          patch_s = frm; patch_b = b - ex;
          if (lane == 0) {  // station.py:71-75
            W_ST(LV_BIKES, frm) = b - ex;
            if (b - ex < W_ST(LV_MIN_BIKES, frm)) W_ST(LV_MIN_BIKES, frm) = b - ex;
          }
          int tt = 1;
          if (tt_pos < CD(tt_cap)) tt = K.tt[(size_t)tt_pos * CD(stride) + e];
          else status |= MRX_CB_ENV_TRANSFER_TIMES_OUT;
          tt_pos++;
          if (!(tt < 0 || t + tt >= CD(max_tick))) {  // (a delivery into the past / after the episode never runs)
            if (tt == 0) late++;   // (another decision follows in the tick's list: the list tail is not stale, event_linked_list.py:86-108)
            if (tail - head >= CD(pool_cap)) {
              status |= MRX_CB_ENV_DELIVERY_OVERFLOW;
            } else {
              const int idx = tail % CD(pool_cap);
              if (lane == 0) { W_POOL(idx, 0) = t + tt; W_POOL(idx, 1) = t; W_POOL(idx, 2) = frm; W_POOL(idx, 3) = to; W_POOL(idx, 4) = ex; }
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
  // ---- action scope of (s1, type) at tick t (decision_strategy.py:253-293): candidate i = a * 64 + lane
  int key[2], val[2];
#pragma unroll
  for (int a = 0; a < 2; a++) {
    const int i = a * 64 + lane;
    const int nb = K.nb[(size_t)s1 * CD(nb_stride) + (i < n ? i : 0)];
    const int bk = nb == patch_s ? patch_b : W_ST(LV_BIKES, nb);
    key[a] = i < n ? nb : -1;
    val[a] = type == MRX_CB_SUPPLY ? K.capacity[nb] - bk : (int)floor((double)bk * K.scope_high);
  }
  for (int f = nf0; f < CD(n_filters); f++) {  // wave-uniform
    const int n_out = CDA(f_num, f) < n ? CDA(f_num, f) : n;
    if (CDA(f_type, f) == MRX_CB_FILTER_REQUIREMENTS) {
      const int v0[2] = {val[0], val[1]};
      cbw_select(scr, n, n_out, 0, v0, key, val);
    } else {
      // TripsWindowFilter :88-163 — see cb::action_scope: a frame's value is frozen at the tick it was last read as the
      // current frame; the newest frame is re-read now (with windows == 0, Python's lst[-0:], only when it is new)
      const int avail = fi_cur + 1 < CD(ring_slots) ? fi_cur + 1 : CD(ring_slots);
      const int aw = CDA(f_win, f) < avail ? CDA(f_win, f) : avail;
      const int cnt = aw > 0 ? aw : avail;
      int hi_o = 0, lo_o = 0;
      if (lane < cnt) {  // lane k: frame fi_cur - k
        const int fi = fi_cur - lane, slot = fi % CD(ring_slots);
        int tag_fi = K.twc_fi[(size_t)slot * CD(stride) + e], tag_t = K.twc_tick[(size_t)slot * CD(stride) + e];
        if (lane == 0 && (aw > 0 || tag_fi != fi)) {
          tag_fi = fi; tag_t = t;
          K.twc_fi[(size_t)slot * CD(stride) + e] = fi; K.twc_tick[(size_t)slot * CD(stride) + e] = t;
        }
        const int tb = tag_fi == fi ? tag_t : snapshot_tick(K, fi);
        int w0 = tb / CD(res) * CD(res);
        if (w0 < CD(start_tick)) w0 = CD(start_tick);
        hi_o = (tb + 1 - CD(start_tick)) * S;
        lo_o = (w0 - CD(start_tick)) * S;
      }
      int trips[2] = {0, 0};
      for (int k = 0; k < cnt; k++) {  // wave-uniform
        const int hi = wave::bcast(hi_o, k), lo = wave::bcast(lo_o, k);
#pragma unroll
        for (int a = 0; a < 2; a++) {
          const int x = key[a] >= 0 ? key[a] : 0;
          trips[a] += K.req_cum[hi + x] - K.req_cum[lo + x];
        }
      }
      cbw_select(scr, n, n_out, type == MRX_CB_DEMAND ? 2 : 1, trips, key, val);
    }
    n = n_out;
  }
  // ---- outputs
#pragma unroll
  for (int a = 0; a < 2; a++) {
    const int i = a * 64 + lane;
    if (i < n) { out[2 * i] = key[a]; out[2 * i + 1] = val[a]; }
  }
  {
    const int bs = s1 == patch_s ? patch_b : W_ST(LV_BIKES, s1);
    if (lane == 0) {
      out[2 * n] = s1;
      out[2 * n + 1] = type == MRX_CB_SUPPLY ? (int)floor((double)bs * K.scope_low_keep) : K.capacity[s1] - bs;
    }
  }
  for (int i = n + 1 + lane; i < CD(scope_cap); i += 64) { out[2 * i] = -1; out[2 * i + 1] = -1; }
  const int m0 = wave::bcast(h, CH_TRIPS), m1 = wave::bcast(h, CH_SHORT), m2 = wave::bcast(h, CH_OPER);
  if (lane == 0) {
    dec[0] = t; dec[1] = s1; dec[2] = type; dec[3] = fi_cur; dec[4] = n + 1; dec[5] = 1; dec[6] = 0; dec[7] = 0;
    met[0] = m0; met[1] = m1; met[2] = m2;
    *done = 0;
  }
  return true;
}

#undef W_HDR
#undef W_ST
#undef W_DMK
#undef W_POOL

}  // namespace cb
