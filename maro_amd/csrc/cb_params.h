// cb_params.h — POD kernel-argument block of the citi_bike engine, shared by host (layout / C-ABI) and device code.
#pragma once
#include <stdint.h>

// Station attribute ids (the ids of mrx_cb_attr_id), in the reference's schema order (citi_bike/station.py:8-39).
enum { SA_BIKES, SA_SHORTAGE, SA_TRIP_REQUIREMENT, SA_FULFILLMENT, SA_CAPACITY, SA_ID, SA_WEEKDAY, SA_TEMPERATURE,
       SA_WEATHER, SA_HOLIDAY, SA_EXTRA_COST, SA_TRANSFER_COST, SA_FAILED_RETURN, SA_MIN_BIKES, SA_COUNT };
enum { CB_MA_TRIPS_ADJ, CB_MA_COUNT };

// Live-frame rows (what actually varies per env); capacity / id / calendar attributes are shared tables.
enum { LV_BIKES, LV_SHORTAGE, LV_TRIP_REQUIREMENT, LV_FULFILLMENT, LV_EXTRA_COST, LV_TRANSFER_COST, LV_FAILED_RETURN,
       LV_MIN_BIKES, LV_COUNT };

// Per-env header words (hdr[w][env]).
enum { CH_TICK, CH_FLAGS, CH_CUR_STATION, CH_CUR_TYPE, CH_TT_POS, CH_TRIPS, CH_SHORT, CH_OPER, CH_POOL_HEAD, CH_POOL_TAIL,
       CH_POOL_MINLAND, CH_LATE, CH_NDEC, CH_STATUS, CH_EV_POS, CH_RES1, CH_WORDS };  // CH_EV_POS: cursor into ev_rec
enum { CFL_FRESH = 1, CFL_FINISHED = 2, CFL_PENDING = 4, CFL_STASH = 8 };  // CFL_STASH: the answer to the pending decision waits in CbParams::stash (CH_RES1 actions)
#define CB_STASH_MAX 4           /* actions an env's stash holds (mrx_cb_set_replay_period needs max_actions <= this) */
enum { CB_POOL_WORDS = 6 };  // land tick, scheduling tick, from, to, number (<0: executed), next entry landing at the same tick (-1: last)
#define CB_LAND_SLOTS 128  /* landing-tick buckets of the delivery pool (power of two): a transfer may take at most 127 ticks */
#define CB_NO_LAND 0x7fffffff
#define CB_POOL_STAGE 256        /* pool entries (from the ring's head on) the wave replay kernel keeps a copy of in LDS (cb_device.h::pool_rd) */
#define CB_BKT_WORDS (2 * CB_LAND_SLOTS + CB_LAND_SLOTS / 32)
#define CB_EVW_RECS 64           /* event records the wave replay kernel fetches ahead, one per lane (cb_device.h::EvWinW) */
#define CB_EVW_WORDS (2 + 3 + CB_EVW_RECS * 4) /* window bounds, alignment slack, records */
#define CB_POOL_STAGE_WORDS (CB_BKT_WORDS + 1 + CB_POOL_STAGE * CB_POOL_WORDS + CB_EVW_WORDS) /* env-major plans' LDS column: buckets, window anchor, entries; the event window */
#define CB_TWC_LDS 32            /* the trip-window filter's per-slot words ride in the LDS column when the ring has at most this many slots */
#define CB_TWC_REG 12            /* trip-window frames whose table rows are kept in registers (cb_device.h::action_scope) */
#define CB_EV_BLOCK 8            /* event records per look-ahead block (cb_device.h::EvWin) */
#define MRX_CB_LDS_BYTES 65536   /* LDS one workgroup (= one wave) of the step kernel may take */
// envs per wave of the step kernel when the caller does not say: a wave runs the union of its lanes' control flow, so a
// small batch is spread over about a thousand waves (four per CU); below 8 envs per wave the extra waves stop paying
// (profiles/r02_citi_bike.md: 4096 envs 62.5 / 64.1 / 66.7 / 64.4 / 54.2 M env-steps/s at 32 / 16 / 8 / 4 / 1 envs per wave;
// 32768 envs 363 M at 64, 381 M at 32)
static inline int cb_auto_lanes(int n_envs) {
  int lanes = 64;
  while (lanes > 8 && n_envs / lanes < 1024) lanes /= 2;
  return lanes;
}
enum { CB_EV_RET, CB_EV_TRIP, CB_EV_REBAL, CB_EV_RETZ, CB_EV_TICK_END };  // kinds of CbParams::ev_rec records

// Word w of env e in a per-env array of W words.  Two layouts (CbParams::aos, fixed at creation):
//   0  struct-of-arrays [word][env stride]: the general kernels own one env per LANE — whenever the lanes of a wave agree on the
//      word (every env replays the same trip table) an access is one contiguous run over the envs;
//   1  env-major [env][W]: the wave-cooperative kernels own one env per WAVE — the env's rows are contiguous, so moving its state
//      HBM <-> LDS, its snapshots and its snapshot queries are coalesced (with [word][env] every word of one env sits in its
//      own 64-byte sector: measured 1.3 MB of traffic to move 40 KB of state).  Chosen for plans that run the wave-cooperative
//      path (cb_layout.h: Sequential mode, aligned frames, >= 96 stations).
#define CB_IX(aos, stride, W, w, e) ((aos) ? (size_t)(e) * (size_t)(W) + (size_t)(w) : (size_t)(w) * (size_t)(stride) + (size_t)(e))

struct CbParams {
  // ---- dimensions / options
  int32_t n_envs, stride, aos, S, start_tick, max_tick, res, ring_slots, max_actions;
  int32_t dres, extra_cost_mode, n_filters, f_type[4], f_num[4], f_win[4];
  int32_t decision_mode;  // 0 Sequential, 1 Joint, 2 JointWithSequentialAction (core.py:349-366)
  int32_t FW, w_mask, w_words, pool_cap, tt_cap, scope_cap, mask_words, nb_stride;
  int32_t lds_words;  // a lane's LDS column in the specialised step kernel: frame, capacities, bit words, scope scratch, event block
  int32_t lsh;        // per launch: log2(envs per wave)
  int32_t lsh_plan;   // plan constant: log2 of the envs per wave the automatic choice gives this plan's batch (cb_auto_lanes, capped by
                      // the LDS column) — a plan-specialised step kernel folds it into its LDS addresses and is only launched with
                      // that many envs per wave; -1: env-major plans (their wave kernels run with lsh 0), lsh stays a kernel argument
  int32_t step_budget;  // per launch: records an env may replay in one step call before it reports "no decision yet" (0: no limit)
  int32_t defer;        // per launch (mrx_k_cb_step_wave): no general kernel follows this call — an env that leaves its tick stashes its answer and reports "no decision yet"
  int32_t pool_stage;   // per launch (mrx_k_cb_replay_wave only; else 0): the env's delivery buckets and this many entries (<= CB_POOL_STAGE) from its pool ring's head on are staged in LDS
  double supply_wm, demand_wm, scope_low_keep, scope_high;
  // ---- per-env struct-of-arrays state: X[word][stride]
  int32_t* hdr;       // [CH_WORDS]
  int32_t* live;      // [FW]  LV_* x S (attr-major)
  int32_t* ring;      // [ring_slots][FW + 1]  (+1: tick the snapshot was taken at)
  int32_t* ring_fi;   // [ring_slots]
  int32_t* twc_fi;    // [ring_slots] TripsWindowFilter cache: the frame a slot was read for ...
  int32_t* twc_tick;  // [ring_slots] ... and the tick it was last read at (the values come from req_cum)
  int32_t* pool;      // [pool_cap][CB_POOL_WORDS]  in-flight DeliverBike events, insertion order
  int32_t* bkt;       // [2 * CB_LAND_SLOTS + CB_LAND_SLOTS / 32] first / last pool entry per landing-tick slot, occupancy mask
  int32_t* tt;        // [tt_cap] transfer times
  int32_t* scratch;   // [3 * S] action-scope work arrays
  uint32_t* fulfilled;  // [w_words]  bit ring over trip index: RequireBike got a bike
  int32_t* prof;        // [16] phase cycle counters (MRX_CB_PROFILE builds only)
  uint32_t* decmask;    // [2 * mask_words] stations with a pending Supply / Demand decision this tick
  uint8_t* todo;        // [n_envs] written by the wave-cooperative decision kernel (cb_wave.h): 1 = the general step must run for this env
  int32_t* stash;       // [CB_STASH_MAX * 3] the actions a deferred env was answered with (mrx_cb_set_replay_period): applied when its replay runs
  // ---- observation fused into the step (mrx_cb_set_observation; runtime configuration, not part of a specialised plan):
  // obs [n_envs][S][obs_n] float64 = snapshot_list["stations"][decision frame :: obs_attr] of the env's new decision
  double* obs;
  int32_t obs_n, obs_attr[8];
  // ---- shared tables (trips restricted to [start_tick, max_tick), re-indexed from 0)
  const int32_t* trip_off;  // [durations + 1] CSR offsets of the trips by tick (trips_adj bound of a frame)
  const int32_t* ev_rec;    // [n_events + 16][4] the event stream every env replays, see cb_layout.h
  const int32_t* req_cum;   // [durations + 1][S] trips with src = s before relative tick d (running trip_requirement)
  const int32_t *adj_off, *adj_idx;  // trips_adj, shared: trip indices grouped by (src, dst); adj_off [S * S + 1]
  const int32_t *capacity, *init_bikes, *station_id, *nb, *nb_cnt;
  const int32_t *tick_day, *cal;  // tick_day [durations] (relative to start_tick) -> cal [n_days][4] weekday, temperature, weather, holiday
};

// The integer fields of CbParams that are constant for one (topology, config, batch size) plan: what a specialised build
// (cb_spec.hip) receives as MRXC_<field> macros (CD() in cb_device.h); the filter arrays become MRXC_f_type(i) etc.
#define MRX_CB_DIM_FIELDS(X) \
  X(stride) \
  X(aos) \
  X(S) \
  X(start_tick) \
  X(max_tick) \
  X(res) \
  X(ring_slots) \
  X(max_actions) \
  X(dres) \
  X(extra_cost_mode) \
  X(n_filters) \
  X(FW) \
  X(w_mask) \
  X(w_words) \
  X(pool_cap) \
  X(tt_cap) \
  X(scope_cap) \
  X(mask_words) \
  X(nb_stride) \
  X(lds_words) \
  X(decision_mode) \
  X(lsh_plan)
#define MRX_CB_DIM_ARRAYS(X) X(f_type) X(f_num) X(f_win)
