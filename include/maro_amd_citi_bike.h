/*
 * maro_amd_citi_bike.h — C ABI of the MI355X batched rollout engine for MARO's citi_bike simulator
 * (SURVEY.md §8 row a20).  Same conventions as maro_amd.h: plain C, caller-owned device pointers,
 * all engine state inside the caller's workspace, asynchronous on `stream`, status codes, no throws.
 *
 * The reference has no FFI for this path; each entry point cites the Python interface it replaces.
 */
#ifndef MARO_AMD_CITI_BIKE_H_
#define MARO_AMD_CITI_BIKE_H_

#include "maro_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Extra per-environment status bits (same status word convention as MRX_ENV_*). */
enum {
  MRX_CB_ENV_INVALID_ACTION = 1,     /* station index out of range — action skipped */
  MRX_CB_ENV_DELIVERY_OVERFLOW = 2,  /* more in-flight DeliverBike events than delivery_capacity */
  MRX_CB_ENV_TRANSFER_TIMES_OUT = 4, /* the env consumed more transfer times than were supplied */
  MRX_CB_ENV_TRANSFER_TOO_LONG = 8   /* (unused: any positive transfer time is representable) */
};

enum { MRX_CB_SUPPLY = 0, MRX_CB_DEMAND = 1 }; /* DecisionType, citi_bike/common.py:57-65 */
enum { MRX_CB_FILTER_DISTANCE = 0, MRX_CB_FILTER_REQUIREMENTS = 1, MRX_CB_FILTER_TRIP_WINDOW = 2 };
enum { MRX_CB_MAX_FILTERS = 4 };

/*
 * Flat citi_bike topology: what CitibikeBusinessEngine reads from a topology folder
 * (citi_bike/business_engine.py:205-260): trips.bin through BinaryReader/ItemTickPicker
 * (data_lib/binary_reader.py:80-112), station_meta.csv, distance_adj.csv, KNYC_daily.bin, config.yml.
 * All arrays are HOST pointers, copied at create.  Trips must be sorted by tick (file order inside a tick).
 */
typedef struct mrx_cb_topology {
  int32_t n_stations, n_trips, n_ticks, n_days;
  const int32_t* trip_tick;     /* [n_trips] minutes from the data start */
  const int32_t* trip_src;      /* [n_trips] station index */
  const int32_t* trip_dst;      /* [n_trips] */
  const int32_t* trip_duration; /* [n_trips] ticks */
  const int32_t* capacity;      /* [S] */
  const int32_t* init_bikes;    /* [S] */
  const int32_t* station_id;    /* [S] */
  const double* distance;       /* [S][S]; 0.0 = not a neighbour (decision_strategy.py:381-391) */
  const int32_t* tick_day;      /* [n_ticks] day index of each tick (business_engine.py:367-369) */
  const int16_t* day_weekday;   /* [n_days] values as the int16 frame attributes store them */
  const int16_t* day_holiday;
  const int16_t* day_weather;
  const int16_t* day_temperature;
  /* decision strategy options (decision_strategy.py:181-211) */
  int32_t resolution;
  double supply_water_mark_ratio, demand_water_mark_ratio, scope_low_ratio, scope_high_ratio;
  int32_t extra_cost_mode; /* 0 source, 1 target, 2 target_neighbors (ExtraCostMode) */
  int32_t n_filters;
  int32_t filter_type[MRX_CB_MAX_FILTERS], filter_num[MRX_CB_MAX_FILTERS], filter_windows[MRX_CB_MAX_FILTERS];
} mrx_cb_topology;

/* Env(...) constructor arguments (core.py:42-56) plus engine capacities. */
typedef struct mrx_cb_config {
  int32_t n_envs, device, start_tick, durations, snapshot_resolution;
  int32_t max_snapshots;      /* <=0 -> ceil(durations/resolution) */
  int32_t max_actions;        /* A: actions accepted per decision per step (>=1) */
  int32_t delivery_capacity;  /* in-flight DeliverBike events per env; <=0 -> 4*S+4.  Size it as
                                 S * (ceil(longest transfer time / decision resolution) + 2): overflow is flagged
                                 (MRX_CB_ENV_DELIVERY_OVERFLOW) and the delivery dropped */
  int32_t transfer_times_cap; /* transfer times per env; <=0 -> S*(durations/resolution+1) */
  int32_t decision_mode;      /* DecisionMode, maro/simulator/abs_core.py:14-22: 0 Sequential (mrx_cb_step), 1 Joint,
                                 2 JointWithSequentialAction (mrx_cb_step_joint) */
} mrx_cb_config;

/* Layout of the per-env arrays inside the workspace.  env_major = 0: every per-env array is struct-of-arrays
 * `int32 [words][env_stride]` (word-major, env-minor: the general kernels own one env per lane, a wave's 64 envs touch 256
 * contiguous bytes).  env_major = 1 (plans that step one env per WAVE: Sequential mode, aligned frames, >= 96 stations): every
 * per-env array is `int32 [env][words]` — an env's rows are contiguous; read `[words][stride]` below as `[env][words]` then. */
typedef struct mrx_cb_layout {
  int32_t n_envs, env_stride, n_stations, frame_words, ring_slots, scope_cap, delivery_capacity, transfer_times_cap;
  int32_t env_major, reserved0;
  int64_t off_hdr;     /* int32 [16][stride]: tick, flags, ..., status (MRX_CB_HDR_*) */
  int64_t off_live;    /* int32 [frame_words][stride]: the 8 per-env station attrs x S (attr-major); trips_adj is a shared table */
  int64_t off_ring;    /* int32 [ring_slots][frame_words + 1][stride] (last word: tick of the snapshot) */
  int64_t off_ring_fi; /* int32 [ring_slots][stride] frame index held by each slot, -1 = empty */
  int64_t off_transfer_times; /* int32 [transfer_times_cap][stride] */
  int64_t workspace_bytes;
  int64_t off_prof;    /* int32 [16][stride]: per-env phase cycle counters, written only by -DMRX_CB_PROFILE builds (tools) */
} mrx_cb_layout;
enum { MRX_CB_HDR_TICK = 0, MRX_CB_HDR_FLAGS = 1, MRX_CB_HDR_STATUS = 13, MRX_CB_HDR_WORDS = 16 };

typedef struct mrx_cb_engine* mrx_cb_handle;

int64_t mrx_cb_workspace_bytes(const mrx_cb_topology* topo, const mrx_cb_config* cfg);
/* Replaces Env.__init__ for N envs (core.py:42-90, citi_bike/business_engine.py:40-99, 205-366). */
int mrx_cb_create(const mrx_cb_topology* topo, const mrx_cb_config* cfg, void* d_workspace, int64_t workspace_bytes,
                  mrx_cb_handle* out);
int mrx_cb_destroy(mrx_cb_handle h);
int mrx_cb_get_layout(mrx_cb_handle h, mrx_cb_layout* out);

/*
 * Replaces Env.reset (core.py:143-170, citi_bike/business_engine.py:164-190) for every env whose mask byte is
 * non-zero (NULL = all).  d_transfer_times int32 [n_envs][n_times] (NULL = keep what the workspace holds) is the
 * sequence BikeDecisionStrategy.transfer_time would yield for that env (`round(np.random.normal(mean, std))`,
 * global numpy RNG in the reference, decision_strategy.py:213-216; SURVEY.md §8c: pre-drawn, fed to the device).
 */
int mrx_cb_reset(mrx_cb_handle h, const int32_t* d_transfer_times, int32_t n_times, const uint8_t* d_env_mask,
                 void* stream);

/*
 * Replaces Env.step(action) in Sequential decision mode (core.py:92-133, 317-381) for every unmasked env:
 * applies the actions to the pending RebalanceBike decision (business_engine.py:521-559), then runs ticks
 * (…:101-147 and the handlers :398-519) until the next decision or the end of the episode.
 *   d_actions   int32 [n_envs][A][3] = (from_station_idx, to_station_idx, number); d_n_actions int32 [n_envs] (NULL = 0)
 *   d_decisions int32 [n_envs][8] = (tick, station_idx, MRX_CB_SUPPLY|DEMAND, frame_index, n_scope, valid, 0, 0)
 *   d_scope     int32 [n_envs][scope_cap][2] = the decision's action_scope as ordered (station, max) pairs,
 *               the deciding station last (decision_strategy.py:253-293); unused rows are (-1, -1)
 *   d_metrics   int64 [n_envs][3] = (trip_requirements, bike_shortage, operation_number)
 */
int mrx_cb_step(mrx_cb_handle h, const int32_t* d_actions, const int32_t* d_n_actions, const uint8_t* d_env_mask,
                int32_t* d_decisions, int32_t* d_scope, int64_t* d_metrics, uint8_t* d_done, void* stream);

/*
 * Replaces Env.step(actions) in the Joint decision modes (core.py:354-366; mrx_cb_config.decision_mode 1 / 2): every pending
 * decision event of the tick — one per station the rebalance check flagged, in station order — is reported at once.
 *   d_decisions  int32 [n_envs][S][8]: row i = the i-th pending event (layout as mrx_cb_step, plus [6] = number of events
 *                reported, [7] = i); rows beyond the events have valid = 0
 *   d_scope      int32 [n_envs][S][scope_cap][2]: row i = that event's action_scope, evaluated on the state at report time
 *                (the reference caches a payload's scope at its first read; an object layer on top re-serves it)
 *   d_actions    int32 [n_envs][S][A][3], d_n_actions int32 [n_envs][S]: the action list of the i-th reported event
 *   d_n_answered int32 [n_envs]: how many of the reported events the agent answered (the reference zips actions with events:
 *                the first len(actions)); the others are finished without effect (Joint) or stay pending and are reported
 *                again by this call (JointWithSequentialAction).  NULL = 0 answered.
 */
int mrx_cb_step_joint(mrx_cb_handle h, const int32_t* d_actions, const int32_t* d_n_actions, const int32_t* d_n_answered,
                      const uint8_t* d_env_mask, int32_t* d_decisions, int32_t* d_scope, int64_t* d_metrics, uint8_t* d_done, void* stream);

/*
 * How many envs share one 64-lane wave of the step kernel (no reference counterpart: the reference steps one env per process).
 * One lane owns one env; a wave executes the union of its lanes' control flow and env-steps differ a lot in length (0 to
 * 20+ ticks), so few envs per wave means less divergence but more waves.  lanes = 1, 2, 4, ..., 64; 0 = automatic (the
 * default: 64, lowered to 32 / 16 / 8 while the batch gives fewer than about four waves per CU, and — plan-specialised kernels — until the
 * lanes' LDS columns fit in 64 KB; env var MRX_CB_LANES overrides it at creation).  Results do not depend on it.
 */
int mrx_cb_set_lanes_per_wave(mrx_cb_handle h, int lanes);

/*
 * The wave-cooperative decision step (no reference counterpart; maro_amd/csrc/cb_wave.h).  At the size of the reference's ny.*
 * topologies (~800 stations, filter chain 80 -> 40 -> 20) a decision tick raises hundreds of decision events per env, so almost
 * every env-step stays inside its tick: apply one action, take the next pending station, evaluate its action scope
 * (decision_strategy.py:253-293).  With this on, mrx_cb_step first launches ONE WAVE PER ENV for exactly those steps — the
 * candidate neighbours are ranked across the lanes (ballot / broadcast / prefix rank) instead of a per-lane selection sort —
 * and then the general kernel for the envs that have to replay events.  Sequential mode, aligned frames, <= 2048 stations.
 * mode: 0 = automatic (on from 96 stations), 1 = on, -1 = off.  Returns 1 / 0 (in effect or not) or a negative mrx_status.
 * Results do not depend on it.  Plan-specialised step kernels: a code object built with the envs-per-wave shift compiled in
 * (MRXC_lsh_plan >= 0: plans whose state fits LDS one env per lane) contains the speculative wave kernel but NOT the wave replay
 * kernel; forcing the mode on wants the runtime-shift build (MRXC_lsh_plan -1) loaded first — the Python engine does that.
 */
int mrx_cb_set_wave_decisions(mrx_cb_handle h, int mode);

/*
 * The two wave kernels of a batch step side by side (no reference counterpart).  With plan-specialised kernels loaded and the
 * wave-cooperative step in effect, mrx_cb_step first reads off the state which envs leave their tick (mrx_k_cb_classify, one
 * wave per env, writes nothing but that flag), then runs the replay kernel for those envs on a stream of the engine's own while
 * the in-tick kernel steps all the others on the caller's stream: a batch step lasts as long as the longer of the two instead of
 * their sum.  The engine's stream is forked from the caller's and joined back into it by events inside the call, so the caller
 * sees one stream-ordered operation as before (also under stream capture).  on = 1 / 0 (the default): the replay kernel after
 * the in-tick kernel, on the caller's stream.  Results do not depend on it.
 */
int mrx_cb_set_replay_overlap(mrx_cb_handle h, int on);

/*
 * How often the general (replay) kernel of the wave-stepped path runs (no reference counterpart; plan-specialised wave kernels, with
 * mrx_cb_set_step_budget).  A call of the replay kernel costs a fixed part — the env's state into LDS and back, one wave's latency
 * chain — whatever the budget; with n > 1 it runs on every n-th mrx_cb_step only, with a budget n times as large the same amount of
 * replaying per call at 1 / n of that fixed cost.  On the calls in between an env that leaves its tick waits: the answer it was
 * just given is kept with the env (applied when its replay runs), and its row says "no decision yet" exactly as under a step budget
 * (decisions[e] = {tick, -1, -1, frame_index, 0, valid = 0, ..}, done as it stands); answers to such a row are ignored.  Every env
 * still sees exactly the reference's sequence of decisions, actions and snapshots.  n = 1 (the default): every call.  Needs
 * max_actions <= 4.  Ignored while mrx_cb_set_replay_overlap is on and on plans that are not wave-stepped.
 * phase (0 .. n - 1): counting this engine's mrx_cb_step calls c = 1, 2, ... from now, the replay kernel runs when (phase + c) % n == 0 —
 * several engines stepped in turn on streams of their own (env groups) take different phases, so that one group's replay kernel (a
 * few hundred waves, one lane busy in each) runs beside the other groups' in-tick kernels instead of beside their replay kernels.
 */
int mrx_cb_set_replay_period(mrx_cb_handle h, int n, int phase);

/*
 * Bounded steps (no reference counterpart).  mrx_cb_step returns when EVERY env of the batch has its next decision, so a call
 * lasts as long as the batch's longest env-step — and env-steps differ by two orders of magnitude (another station deciding at
 * the same tick: nothing to simulate; the last decision of a tick: twenty ticks of trips, two snapshots, a rebalance sweep).
 * With max_records > 0 an env replays at most about that many events per call; if it has not reached a decision by then its
 * row says so (decisions[e] = {tick, -1, -1, frame_index, 0, valid = 0, ..}, done = 0), the actions passed for it are ignored
 * and the next call continues where it stopped.  Every env still sees exactly the reference's sequence of decisions, actions
 * and snapshots; only the grouping into calls changes.  0 = off (the default: every call yields a decision or `done`).
 */
int mrx_cb_set_step_budget(mrx_cb_handle h, int max_records);

/*
 * Fuse an agent's per-decision snapshot slice into mrx_cb_step: d_obs float64 [n_envs][rows][n_attrs] is rewritten by every step with
 *   snapshot_list["stations"][frame_index of the env's new decision : nodes : station_attrs]   (citi_bike/business_engine.py:101-147;
 * the slice an agent reads after Env.step, examples/citi_bike) — exactly what mrx_cb_query("stations", decisions[:, 3], nodes,
 * station_attrs) returns, without the extra launch: the decision's frame is the live frame the step kernel holds.
 *   plans stepped one env per LANE (below 96 stations): rows = S, nodes = every station;
 *   plans stepped by the wave-cooperative kernels (mrx_cb_layout.env_major; needs the plan-specialised kernels): rows = scope_cap,
 *   nodes = the stations of the decision's action scope, d_scope[e][i][0] (the deciding station and its filtered neighbours: what an
 *   agent can act on; -1 padding rows are zeros) — at 800 stations the full slice would be 45 KB of float64 per env-step.
 * Rows of envs without a valid decision (finished; out of step budget) are zeros.  Sequential decision mode; n_attrs = 0 switches
 * it off.
 */
int mrx_cb_set_observation(mrx_cb_handle h, const int32_t* station_attrs, int n_attrs, double* d_obs);
/*
 * Rows per env of the fused observation on the step path currently in effect (read-only): scope_cap on wave-stepped plans, the
 * number of stations otherwise — the second dimension d_obs must have.  The buffer is sized for ONE row layout: while an
 * observation is configured, mrx_cb_set_wave_decisions / mrx_cb_load_step_kernels refuse a change that would switch the layout
 * (MRX_ERR_UNSUPPORTED; switch the observation off first, n_attrs = 0).  No reference counterpart (the reference returns
 * whatever nodes the caller slices: maro/backends/np_backend.pyx:520-549).
 */
int mrx_cb_observation_rows(mrx_cb_handle h);

/*
 * Replaces snapshot_list["stations" | "matrices"][ticks:nodes:attrs] (frame.pyx:754-801, np_backend.pyx:520-549).
 * node_type 0 = stations, 1 = matrices; arguments as mrx_cim_query.  trips_adj has S*S slots, every other attr 1.
 */
int mrx_cb_query(mrx_cb_handle h, int node_type, const int32_t* d_ticks, int nt, int ticks_per_env, const int32_t* d_nodes,
                 int nn, int nodes_per_env, const int32_t* attrs, int na, double* d_out, void* stream);

/*
 * Utility device policy (counter-based, replayed by the tests' oracle): for every env with a valid decision pick
 * the first other station of the scope and move hash-chosen 0..min(scope[self], scope[other]) bikes
 * (Supply: self -> other, Demand: other -> self).  Adds the number of valid decisions to *d_counter (may be NULL).
 */
int mrx_cb_random_policy(mrx_cb_handle h, const int32_t* d_decisions, const int32_t* d_scope, int64_t step,
                         int32_t* d_actions, int32_t* d_n_actions, uint64_t* d_counter, void* stream);

/*
 * Plan-specialised kernels (same scheme as mrx_cim_plan_defines / mrx_cim_load_step_kernels in maro_amd.h): the text returned by
 * mrx_cb_plan_defines (host only; here the batch size matters: the struct-of-arrays stride is one of the constants), saved as
 * cb_spec_dims.h, turns maro_amd/csrc/cb_spec.hip into a code object whose mrx_k_cb_reset / mrx_k_cb_step have every plan
 * dimension as a compile-time constant and keep frames of up to 128 words in registers (+50 % env-steps/s on toy.3s_4t); mrx_cb_load_step_kernels makes the handle use it.
 */
int64_t mrx_cb_plan_defines(const mrx_cb_topology* topo, const mrx_cb_config* cfg, char* buf, int64_t len);
int mrx_cb_load_step_kernels(mrx_cb_handle h, const void* image, int64_t bytes, const char* defines);

int mrx_cb_attr_id(int node_type, const char* name);
int mrx_cb_attr_slots(mrx_cb_handle h, int node_type, int attr_id);

#ifdef __cplusplus
}
#endif
#endif /* MARO_AMD_CITI_BIKE_H_ */
