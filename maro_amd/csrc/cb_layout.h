// cb_layout.h — host-side planning of the citi_bike engine's HBM workspace and shared tables.
// Plain C++ (no HIP): used by the C-ABI implementation (cb_engine.hip) and the CPU harness in tests/emu/.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "../../include/maro_amd_citi_bike.h"
#include "cb_params.h"

struct CbHostPlan {
  CbParams kp;  // pointers valid after cb_plan_bind()
  mrx_cb_layout layout;
  std::vector<uint8_t> const_blob;
  int64_t const_off = 0, workspace_bytes = 0;
  std::vector<std::pair<size_t, int64_t>> binds;  // (offset of a pointer field inside kp, byte offset in the workspace)
};

namespace cb_layout_detail {
inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
template <class T>
inline int64_t blob_put(std::vector<uint8_t>& blob, const std::vector<T>& v) {
  size_t off = (blob.size() + 255) / 256 * 256;
  blob.resize(off + sizeof(T) * (v.size() ? v.size() : 1), 0);
  if (!v.empty()) memcpy(blob.data() + off, v.data(), sizeof(T) * v.size());
  return (int64_t)off;
}
}  // namespace cb_layout_detail

inline int cb_plan(const mrx_cb_topology* t, const mrx_cb_config* c, CbHostPlan* pl, std::string* err) {
  using namespace cb_layout_detail;
  auto bad = [&](const char* m, int code) { if (err) *err = m; return code; };
  if (!t || !c) return bad("null topology/config", MRX_ERR_INVALID_ARG);
  if (c->n_envs <= 0 || c->durations <= 0 || c->snapshot_resolution <= 0 || c->start_tick < 0)
    return bad("n_envs, durations and snapshot_resolution must be positive", MRX_ERR_INVALID_ARG);
  const int S = t->n_stations;
  if (S <= 0 || S > 4096) return bad("engine limit: 1..4096 stations", MRX_ERR_UNSUPPORTED);
  if (t->resolution <= 0) return bad("decision resolution must be positive", MRX_ERR_INVALID_ARG);
  if (t->n_filters < 0 || t->n_filters > MRX_CB_MAX_FILTERS) return bad("at most 4 neighbour filters", MRX_ERR_UNSUPPORTED);
  const int max_tick = c->start_tick + c->durations;
  if (max_tick > t->n_ticks) return bad("start_tick + durations exceeds the topology's tick range", MRX_ERR_INVALID_ARG);
  for (int s = 0; s < S; s++)
    if (t->capacity[s] <= 0 || t->init_bikes[s] < 0) return bad("station capacity must be positive", MRX_ERR_INVALID_ARG);
  bool reordered = false;
  for (int f = 0; f < t->n_filters; f++) {
    if (t->filter_type[f] == MRX_CB_FILTER_DISTANCE) {
      // DistanceFilter indexes its input by the nearest neighbours (decision_strategy.py:40-47): after a reordering
      // filter dropped some of them the reference raises KeyError
      if (reordered) return bad("a distance filter after a requirements / trip-window filter is not supported", MRX_ERR_UNSUPPORTED);
    } else if (t->filter_type[f] == MRX_CB_FILTER_REQUIREMENTS || t->filter_type[f] == MRX_CB_FILTER_TRIP_WINDOW) {
      reordered = true;
    } else {
      return bad("unknown neighbour filter type", MRX_ERR_INVALID_ARG);
    }
    if (t->filter_num[f] < 0 || t->filter_windows[f] < 0) return bad("negative filter option", MRX_ERR_INVALID_ARG);
  }
  if (c->decision_mode < 0 || c->decision_mode > 2) return bad("decision_mode must be 0 (Sequential), 1 (Joint) or 2 (JointWithSequentialAction)", MRX_ERR_INVALID_ARG);
  CbParams& k = pl->kp;
  memset(&k, 0, sizeof(k));
  k.decision_mode = c->decision_mode;
  k.n_envs = c->n_envs;
  k.stride = (int)align_up(c->n_envs, 64);
  // per-env state env-major for the plans that run one env per WAVE (cb_params.h CB_IX); MRX_CB_AOS = 0 / 1 forces it (tests)
  k.aos = (c->decision_mode == 0 && c->start_tick % c->snapshot_resolution == 0 && S >= 96 && (S + 31) / 32 <= 64) ? 1 : 0;
  if (const char* ev = getenv("MRX_CB_AOS")) k.aos = atoi(ev) ? 1 : 0;
  k.S = S; k.start_tick = c->start_tick; k.max_tick = max_tick; k.res = c->snapshot_resolution;
  const int total_frames = (c->durations + c->snapshot_resolution - 1) / c->snapshot_resolution;
  k.ring_slots = c->max_snapshots > 0 ? c->max_snapshots : total_frames;
  k.max_actions = c->max_actions > 0 ? c->max_actions : 1;
  k.dres = t->resolution; k.extra_cost_mode = t->extra_cost_mode; k.n_filters = t->n_filters;
  for (int f = 0; f < t->n_filters; f++) { k.f_type[f] = t->filter_type[f]; k.f_num[f] = t->filter_num[f]; k.f_win[f] = t->filter_windows[f]; }
  k.supply_wm = t->supply_water_mark_ratio; k.demand_wm = t->demand_water_mark_ratio;
  k.scope_low_keep = 1 - t->scope_low_ratio;  // decision_strategy.py:287: floor(bikes * (1 - scope_low_ratio))
  k.scope_high = t->scope_high_ratio;
  k.FW = LV_COUNT * S;  // (trips_adj is not per-env state: see adj_off below)
  k.mask_words = (S + 31) / 32;
  k.pool_cap = c->delivery_capacity > 0 ? c->delivery_capacity : 4 * S + 4;
  k.tt_cap = c->transfer_times_cap > 0 ? c->transfer_times_cap : S * (c->durations / t->resolution + 1);

  // ---- neighbours sorted by distance (stable; decision_strategy.py:381-391)
  int nb_max = 0;
  std::vector<std::vector<int>> nbs(S);
  for (int s = 0; s < S; s++) {
    for (int i = 0; i < S; i++) if (t->distance[(size_t)s * S + i] != 0.0) nbs[s].push_back(i);
    std::stable_sort(nbs[s].begin(), nbs[s].end(), [&](int a, int b) { return t->distance[(size_t)s * S + a] < t->distance[(size_t)s * S + b]; });
    nb_max = std::max(nb_max, (int)nbs[s].size());
  }
  k.nb_stride = nb_max > 0 ? nb_max : 1;
  std::vector<int32_t> nb((size_t)S * k.nb_stride, -1), nb_cnt(S);
  for (int s = 0; s < S; s++) { nb_cnt[s] = (int)nbs[s].size(); for (size_t i = 0; i < nbs[s].size(); i++) nb[(size_t)s * k.nb_stride + i] = nbs[s][i]; }
  int keep = nb_max;
  for (int f = 0; f < t->n_filters; f++) keep = std::min(keep, t->filter_num[f]);
  k.scope_cap = keep + 1;

  // ---- trips of [start_tick, max_tick), CSR by tick
  const int D = c->durations;
  int lo = 0, hi = 0;
  for (int i = 0; i < t->n_trips; i++) {
    if (i > 0 && t->trip_tick[i] < t->trip_tick[i - 1]) return bad("trips must be sorted by tick", MRX_ERR_INVALID_ARG);
    if (t->trip_tick[i] < c->start_tick) lo = i + 1;
    if (t->trip_tick[i] < max_tick) hi = i + 1;
    if (t->trip_src[i] < 0 || t->trip_src[i] >= S || t->trip_dst[i] < 0 || t->trip_dst[i] >= S || t->trip_duration[i] < 0)
      return bad("trip with an invalid station index or a negative duration", MRX_ERR_INVALID_ARG);
  }
  const int n = hi - lo;
  std::vector<int32_t> trip_off(D + 1, 0), ttick(n), tsrc(n), tdst(n);
  for (int i = 0; i < n; i++) { ttick[i] = t->trip_tick[lo + i]; tsrc[i] = t->trip_src[lo + i]; tdst[i] = t->trip_dst[lo + i]; trip_off[ttick[i] - c->start_tick + 1]++; }
  for (int d = 0; d < D; d++) trip_off[d + 1] += trip_off[d];
  // returns landing at each tick, in insertion order = trip order; same-tick (duration 0) ones last (ret_mid)
  std::vector<int32_t> ret_off(D + 1, 0), ret_mid(D, 0), ret_trip;
  {
    std::vector<std::vector<int32_t>> by_tick(D);
    int maxdur = 0;
    for (int i = 0; i < n; i++) {
      const int dur = t->trip_duration[lo + i], land = ttick[i] + dur - c->start_tick;
      if (land < D) { by_tick[land].push_back(i); maxdur = std::max(maxdur, dur); }
    }
    for (int d = 0; d < D; d++) {
      ret_off[d] = (int32_t)ret_trip.size();
      int mid = (int)by_tick[d].size();
      for (size_t j = 0; j < by_tick[d].size(); j++) {
        if (ttick[by_tick[d][j]] - c->start_tick == d && mid == (int)by_tick[d].size()) mid = (int)j;
        ret_trip.push_back(by_tick[d][j]);
      }
      ret_mid[d] = ret_off[d] + mid;
    }
    ret_off[D] = (int32_t)ret_trip.size();
    // bit ring over trip indices: must cover every trip that can still be in flight
    int span = 1;
    for (int d = 0; d < D; d++) span = std::max(span, trip_off[d + 1] - trip_off[std::max(d - maxdur, 0)]);
    int w = 32;
    while (w < span) w *= 2;
    k.w_mask = w - 1; k.w_words = w / 32;
  }
  k.lds_words = k.FW + S + k.w_words + 2 * k.mask_words + (k.aos ? 0 : 3 * S) + CH_WORDS + (k.ring_slots <= CB_TWC_LDS ? 2 * k.ring_slots : 0) +
                (k.aos ? CB_POOL_STAGE_WORDS : 0) + CB_EV_BLOCK * 4;  // cb_device.h: LDS_CAP .. LDS_TWC + the event block (env-major plans keep the scope scratch in HBM)
  k.lsh_plan = -1;
  if (!k.aos && (int64_t)k.lds_words * 4 <= MRX_CB_LDS_BYTES) {  // (what mrx_cb_step's launch computes for the automatic choice)
    int lanes = cb_auto_lanes(c->n_envs);
    while (lanes > 1 && (int64_t)k.lds_words * 4 * lanes > MRX_CB_LDS_BYTES) lanes /= 2;
    k.lsh_plan = 0;
    while ((1 << k.lsh_plan) < lanes) k.lsh_plan++;
  }
  std::vector<int32_t> tick_day(D), cal((size_t)std::max(t->n_days, 1) * 4, 0);
  for (int d = 0; d < D; d++) {
    tick_day[d] = t->tick_day[c->start_tick + d];
    if (tick_day[d] < 0 || tick_day[d] >= t->n_days) return bad("tick_day out of range", MRX_ERR_INVALID_ARG);
  }
  for (int d = 0; d < t->n_days; d++) { cal[d * 4 + 0] = t->day_weekday[d]; cal[d * 4 + 1] = t->day_temperature[d]; cal[d * 4 + 2] = t->day_weather[d]; cal[d * 4 + 3] = t->day_holiday[d]; }

  // ---- shared tables -> const blob
  std::vector<uint8_t>& blob = pl->const_blob;
  blob.clear();
  pl->binds.clear();
  std::vector<std::pair<size_t, int64_t>> crel;
  auto put = [&](const int32_t* CbParams::*field, const std::vector<int32_t>& v) {
    crel.push_back({(size_t)((const char*)&(k.*field) - (const char*)&k), blob_put(blob, v)});
  };
  put(&CbParams::trip_off, trip_off);
  {
    // The ONE event stream every env replays (cb_device.h::step_env): per tick, in the reference's execution order, the
    // returns scheduled by earlier ticks, the trips, the rebalance check of a decision tick, the zero-duration returns, and
    // an end-of-tick record where something happens at the end (frame end, decision tick, last tick).  Ticks with
    // nothing to do have no record at all.  4 words per record: (tick - start_tick) << 3 | kind, then
    //   RET / RETZ: trip index, its scheduling tick, src | dst << 16      TRIP: trip index, src      TICK_END: flags
    // (+2 blocks of padding: the look-ahead of cb_device.h::EvWin reads past the last record)
    std::vector<int32_t> ev;
    auto rec = [&](int d, int kind, int a, int b, int c2) { ev.push_back(d << 3 | kind); ev.push_back(a); ev.push_back(b); ev.push_back(c2); };
    for (int d = 0; d < D; d++) {
      const int tk = c->start_tick + d;
      for (int r = ret_off[d]; r < ret_mid[d]; r++) { const int i = ret_trip[r]; rec(d, CB_EV_RET, i, ttick[i], tsrc[i] | tdst[i] << 16); }
      for (int i = trip_off[d]; i < trip_off[d + 1]; i++) rec(d, CB_EV_TRIP, i, tsrc[i], 0);
      const bool decision_tick = (tk + 1) % t->resolution == 0, frame_end = (tk + 1) % c->snapshot_resolution == 0, last = d + 1 == D;
      if (decision_tick) rec(d, CB_EV_REBAL, 0, 0, 0);
      for (int r = ret_mid[d]; r < ret_off[d + 1]; r++) { const int i = ret_trip[r]; rec(d, CB_EV_RETZ, i, ttick[i], tsrc[i] | tdst[i] << 16); }
      if (decision_tick || frame_end || last) rec(d, CB_EV_TICK_END, (frame_end ? 1 : 0) | (last ? 2 : 0), 0, 0);
    }
    if (ev.size() / 4 > (size_t)0x7fffff00) return bad("event stream too long", MRX_ERR_UNSUPPORTED);
    ev.resize(ev.size() + (2 * CB_EV_BLOCK + CB_EVW_RECS) * 4, (int32_t)((D << 3) | CB_EV_TICK_END));   // (look-ahead reads run past the last record)
    put(&CbParams::ev_rec, ev);
  }
  {
    // trips_adj[src][dst] counts EVERY RequireBike of the episode so far, fulfilled or not (business_engine.py:412), and is
    // never reset inside an episode: it is a function of (shared trip table, tick) only.  Kept as trip indices grouped by
    // (src, dst) pair: the matrix of a frame taken at tick t = per pair, how many of its indices are < trip_off[t + 1].
    std::vector<int32_t> adj_off((size_t)S * S + 1, 0), adj_idx(n);
    for (int i = 0; i < n; i++) adj_off[(size_t)tsrc[i] * S + tdst[i] + 1]++;
    for (size_t c2 = 0; c2 < (size_t)S * S; c2++) adj_off[c2 + 1] += adj_off[c2];
    std::vector<int32_t> fill(adj_off.begin(), adj_off.end() - 1);
    for (int i = 0; i < n; i++) adj_idx[fill[(size_t)tsrc[i] * S + tdst[i]]++] = i;
    put(&CbParams::adj_off, adj_off); put(&CbParams::adj_idx, adj_idx);
  }
  {
    bool has_window = false;
    for (int f = 0; f < t->n_filters; f++) has_window |= t->filter_type[f] == MRX_CB_FILTER_TRIP_WINDOW;
    std::vector<int32_t> req_cum(has_window ? (size_t)(D + 1) * S : 1, 0);
    if (has_window)
      for (int d = 0; d < D; d++) {
        memcpy(&req_cum[(size_t)(d + 1) * S], &req_cum[(size_t)d * S], sizeof(int32_t) * S);
        for (int i = trip_off[d]; i < trip_off[d + 1]; i++) req_cum[(size_t)(d + 1) * S + tsrc[i]]++;
      }
    put(&CbParams::req_cum, req_cum);
  }
  put(&CbParams::capacity, std::vector<int32_t>(t->capacity, t->capacity + S));
  put(&CbParams::init_bikes, std::vector<int32_t>(t->init_bikes, t->init_bikes + S));
  put(&CbParams::station_id, std::vector<int32_t>(t->station_id, t->station_id + S));
  put(&CbParams::nb, nb); put(&CbParams::nb_cnt, nb_cnt); put(&CbParams::tick_day, tick_day); put(&CbParams::cal, cal);

  // ---- workspace arena: [const blob | per-env SoA arrays]
  int64_t top = 0;
  auto take = [&](int64_t words) { int64_t o = align_up(top, 256); top = o + words * 4 * (int64_t)k.stride; return o; };
  pl->const_off = 0;
  top = (int64_t)blob.size();
  for (auto& b : crel) pl->binds.push_back({b.first, pl->const_off + b.second});
  auto env_arr = [&](auto CbParams::*field, int64_t words) {
    int64_t o = take(words);
    pl->binds.push_back({(size_t)((const char*)&(k.*field) - (const char*)&k), o});
    return o;
  };
  mrx_cb_layout& L = pl->layout;
  memset(&L, 0, sizeof(L));
  L.off_hdr = env_arr(&CbParams::hdr, CH_WORDS);
  L.off_live = env_arr(&CbParams::live, k.FW);
  L.off_ring = env_arr(&CbParams::ring, (int64_t)k.ring_slots * (k.FW + 1));
  L.off_ring_fi = env_arr(&CbParams::ring_fi, k.ring_slots);
  env_arr(&CbParams::twc_fi, k.ring_slots);
  env_arr(&CbParams::twc_tick, k.ring_slots);
  env_arr(&CbParams::pool, (int64_t)k.pool_cap * CB_POOL_WORDS);
  env_arr(&CbParams::bkt, 2 * CB_LAND_SLOTS + CB_LAND_SLOTS / 32);
  L.off_transfer_times = env_arr(&CbParams::tt, k.tt_cap);
  env_arr(&CbParams::scratch, 3 * (int64_t)S);
  env_arr(&CbParams::fulfilled, k.w_words);
  env_arr(&CbParams::decmask, 2 * (int64_t)k.mask_words);
  L.off_prof = env_arr(&CbParams::prof, 16);
  env_arr(&CbParams::stash, CB_STASH_MAX * 3);
  env_arr(&CbParams::todo, 1);   // (one byte per env is used: written by the wave-cooperative decision kernel, read as the general kernel's mask)
  pl->workspace_bytes = align_up(top, 256);
  L.n_envs = k.n_envs; L.env_stride = k.stride; L.env_major = k.aos; L.n_stations = S; L.frame_words = k.FW; L.ring_slots = k.ring_slots;
  L.scope_cap = k.scope_cap; L.delivery_capacity = k.pool_cap; L.transfer_times_cap = k.tt_cap;
  L.workspace_bytes = pl->workspace_bytes;
  return MRX_OK;
}

// Point kp's pointers into a workspace located at `base` (device or host address).
inline void cb_plan_bind(CbHostPlan* pl, void* base) {
  for (auto& b : pl->binds) {
    void* p = (uint8_t*)base + b.second;
    memcpy((uint8_t*)&pl->kp + b.first, &p, sizeof(void*));
  }
}
