/*
 * maro_amd.h — C ABI of the MI355X batched rollout engine for MARO's CIM simulator.
 *
 * This is the drop-in boundary for the hot path named by BASELINE.json:north_star:
 * thousands of independent CIM environments stepping on one GPU.  The reference has
 * no FFI for this path (its native seam is the per-scalar Cython BackendAbc,
 * maro/backends/backend.pxd:47-157, which is the wrong granularity for a batch engine),
 * so every entry point below cites the reference *Python* interface it replaces.
 *
 * Conventions
 *  - plain C, no torch / HIP types in signatures (`void* stream` is a hipStream_t);
 *  - every `d_*` pointer is a CALLER-OWNED DEVICE pointer (e.g. torch tensor data_ptr());
 *    the engine owns only its handle; all engine state lives in the caller-provided
 *    workspace (`d_workspace`), so PyTorch stays the device allocator;
 *  - all calls are asynchronous on `stream`, never synchronise, never throw; they
 *    return 0 or a negative mrx_status; mrx_last_error() describes the last failure
 *    on the calling thread;
 *  - one handle <-> one stream at a time (thread-compatible, no internal threads).
 */
#ifndef MARO_AMD_H_
#define MARO_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mrx_status {
  MRX_OK = 0,
  MRX_ERR_INVALID_ARG = -1,
  MRX_ERR_UNSUPPORTED = -2, /* topology exceeds engine limits (ports/vessels > 64 ...) */
  MRX_ERR_WORKSPACE = -3,   /* workspace too small / misaligned */
  MRX_ERR_HIP = -4,         /* a HIP runtime call failed */
  MRX_ERR_NO_DEVICE = -5
} mrx_status;

/* Per-environment status bits written by the kernels (mrx_cim_layout.off_status). */
enum {
  MRX_ENV_OK = 0,
  MRX_ENV_INVALID_ACTION = 1, /* reference: AssertionError in _on_action_received,
                                 cim/business_engine.py:731,736 — action skipped */
  MRX_ENV_STOP_OVERFLOW = 2,  /* route unrolling exceeded max_stops (engine limit) */
  MRX_ENV_OFFROUTE_ACTION = 32 /* an Action named a port that is not on the vessel's route: containers are moved
                                  as in the reference but the vessel_plans[v,p] += period update is not representable */
};

/* Action types, reference cim/common.py:18-22 (ActionType.LOAD / DISCHARGE). */
enum { MRX_ACTION_LOAD = 0, MRX_ACTION_DISCHARGE = 1 };

/*
 * Flat CIM topology: everything the reference parses out of topologies/<name>/config.yml
 * (maro/data_lib/cim/parsers.py:14-211, cim_data_generator.py:118-205).  Index order is
 * yml order (parsers.py:27-52, 151-156).  All arrays are HOST pointers, copied at create.
 * Numbers that the reference keeps as Python int/float and feeds into float arithmetic
 * are doubles here (exactly representable), so device fp64 reproduces CPython bit for bit.
 */
typedef struct mrx_cim_topology {
  int32_t n_ports, n_vessels, n_routes;
  int32_t n_targets;      /* sum over ports of len(order_distribution.targets) */
  int32_t n_route_points; /* sum over routes of len(route) */
  int32_t past_stop_number, future_stop_number; /* stop_number: [past, future] */
  int32_t container_volume;                     /* container_volumes[0] */
  int32_t total_containers;
  int32_t order_mode; /* 0 = fixed, 1 = unfixed (OrderGenerateMode, entities.py) */
  int64_t seed;       /* topology default seed */

  /* container_usage_proportion, after np.interp over one period (parsers.py:80-91) */
  int32_t period;
  double sample_noise;
  const double* order_dist; /* [period] */

  /* ports */
  const int32_t* port_capacity;   /* [P] */
  const int32_t* port_init_empty; /* [P] int(initial_container_proportion*total) parsers.py:190 */
  const double* empty_return_base; /* [P] empty_return.buffer_ticks */
  const double* empty_return_noise;
  const double* full_return_base; /* [P] full_return.buffer_ticks */
  const double* full_return_noise;
  const double* source_base; /* [P] order_distribution.source.proportion */
  const double* source_noise;
  const int32_t* target_offset; /* [P+1] CSR into target_* */
  const int32_t* target_port;   /* [n_targets] destination port index */
  const double* target_base;    /* [n_targets] */
  const double* target_noise;

  /* routes */
  const int32_t* route_offset; /* [R+1] CSR into route_* */
  const int32_t* route_port;   /* [n_route_points] */
  const double* route_dist;    /* [n_route_points] distance_to_next_port */

  /* vessels */
  const int32_t* vessel_capacity;     /* [V] */
  const int32_t* vessel_init_empty;   /* [V] vessels.<name>.empty (default 0) */
  const int32_t* vessel_route;        /* [V] route index */
  const int32_t* vessel_start_offset; /* [V] position of initial_port_name in its route */
  const double* vessel_speed;         /* [V] sailing.speed */
  const double* vessel_speed_noise;
  const double* vessel_duration; /* [V] parking.duration */
  const double* vessel_duration_noise;

  /* ---- data read from files instead of generated at reset (maro/data_lib/cim/cim_data_loader.py:360-450):
   * 1 = dump folder (data_from_dumps: stops.csv|bin, global_order_proportion.txt, vessel periods; orders are still
   *     drawn by the synthetic generator), 2 = real data files (data_from_files: stops + orders.csv|bin; no order
   *     generator, ports carry no order distribution).  In both modes every reset re-seeds the RNG registry with the
   *     data set's own seed (cim_data_container_helpers.py:79-85, 118-123), so seed commands are ignored. */
  int32_t data_mode;       /* 0 = generated from config.yml (everything above) */
  int32_t data_max_tick;   /* misc.yml max_tick: ticks covered by fixed_order_prop / fixed_orders */
  int32_t fixed_max_stops; /* row length of fixed_stops_* */
  const int32_t* fixed_n_stops;       /* [V] */
  const int32_t* fixed_stops_arrival; /* [V][fixed_max_stops]; stop k of a vessel is at route position (start + k) mod len */
  const int32_t* fixed_stops_leave;   /* [V][fixed_max_stops] */
  const int32_t* fixed_vessel_period; /* [V] */
  const int32_t* fixed_order_prop;    /* [data_max_tick] order_proportion (mode 1) */
  const int32_t* fixed_orders;        /* [data_max_tick][n_targets] quantity per order pair (mode 2), target_offset CSR
                                         order = the order the file lists a tick's orders in */
} mrx_cim_topology;

/* Env(...) constructor arguments, reference maro/simulator/core.py:42-56. */
typedef struct mrx_cim_config {
  int32_t n_envs;
  int32_t device;              /* HIP device ordinal */
  int32_t start_tick;          /* core.py:46 */
  int32_t durations;           /* core.py:47; max_tick = start_tick + durations */
  int32_t snapshot_resolution; /* core.py:48 */
  int32_t max_snapshots;       /* core.py:49; <=0 -> ceil(durations/resolution) (abs_business_engine.py:115-129) */
  int32_t max_actions;         /* A: actions accepted per decision event per step (>=1) */
  int32_t max_stops;           /* <=0 -> engine computes a safe bound per vessel */
  int32_t decision_mode;       /* DecisionMode, maro/simulator/abs_core.py:14-22: 0 Sequential (mrx_cim_step),
                                  1 Joint, 2 JointWithSequentialAction (mrx_cim_step_joint) */
  int32_t order_table;         /* 0 = auto, 1 = on, -1 = off.  In `fixed` order mode the orders of a tick are a pure
                                  function of (seed, tick) (cim_data_container.py:309-398), so mrx_cim_reset can draw
                                  the whole episode's order table ahead of time (int32 [durations][n_targets] per env)
                                  and the step kernel reads its tick's row instead of generating it.  auto = on
                                  whenever the topology allows it. */
} mrx_cim_config;

/* Word (4-byte) offsets describing the engine's HBM layout inside the workspace, so the
 * host can build zero-copy tensor views (live frames, snapshot ring, status words). */
typedef struct mrx_cim_layout {
  int32_t n_envs, n_ports, n_vessels;
  int32_t frame_words;      /* FW: words per env frame */
  int32_t ring_slots;       /* S: snapshot ring capacity per env */
  int32_t max_stops;        /* stop-table capacity per vessel */
  int32_t horizon;          /* H: pending-return ring depth in ticks */
  /* within one frame (word offsets): ports are [attr][port], vessels [attr][slot][vessel] */
  int32_t frame_off_ports;    /* 12 attrs x P */
  int32_t frame_off_vessels;  /* 24 words x V */
  int32_t frame_off_full_on_ports;   /* compact: one cell per (src, dst) order pair (target_offset CSR order) */
  int32_t frame_off_full_on_vessels; /* compact: one cell per (vessel, distinct port of its route); all other */
  int32_t frame_off_vessel_plans;    /* cells of the dense V*P matrices are constant (0 / -1); mrx_cim_query expands */
  /* byte offsets of the big arrays inside the workspace */
  int64_t off_live;    /* int32 [n_envs][FW] */
  int64_t off_ring;    /* int32 [n_envs][S][FW] */
  int64_t off_ring_fi; /* int32 [n_envs][S]  frame index held by each slot, -1 = empty */
  int64_t off_status;  /* int32 [n_envs] MRX_ENV_* bits */
  int64_t off_tick;    /* int32 [n_envs] current tick of each env */
  int64_t off_seed;    /* int64 [n_envs] base seed in use */
  int64_t off_stops;   /* uint32 [n_envs][V][max_stops]  (arrival<<8 | parking) */
  int64_t off_nstops;  /* int32 [n_envs][V] */
  int64_t off_order_prop; /* int32 [n_envs or 1][max_tick] */
  int64_t off_vessel_period; /* int32 [n_envs][V] vessel_period_without_noise */
  int64_t off_orders;        /* [n_envs][durations][order_row_words] pre-generated order quantities per (src, dst) pair in
                                target_offset CSR order, elements of order_elem_bytes bytes: uint16 when the plan proves every
                                quantity fits (every source / target base >= |noise|, so no noised ratio is negative and a
                                tick's orders never exceed its order proportion, itself <= 65535), else int32;
                                0 when the order table is off */
  int32_t order_row_words, order_table_on;   /* (order_row_words = elements per row) */
  int32_t order_elem_bytes, reserved0;
  int64_t workspace_bytes;
} mrx_cim_layout;

typedef struct mrx_cim_engine* mrx_handle;

/* Bytes of device workspace the caller must allocate (256-byte aligned). */
int64_t mrx_cim_workspace_bytes(const mrx_cim_topology* topo, const mrx_cim_config* cfg);

/* Replaces Env.__init__ for N envs (core.py:42-90, cim/business_engine.py:40-105).
 * Does not generate data: call mrx_cim_reset before the first step. */
int mrx_cim_create(const mrx_cim_topology* topo, const mrx_cim_config* cfg, void* d_workspace,
                   int64_t workspace_bytes, mrx_handle* out);
int mrx_cim_destroy(mrx_handle h);
int mrx_cim_get_layout(mrx_handle h, mrx_cim_layout* out);

/*
 * Replaces Env.reset / Env.set_seed (core.py:143-170,219-229;
 * cim_data_container_helpers.py:56-73; cim/business_engine.py:226-245) for every env whose
 * d_env_mask byte is non-zero (NULL = all).  d_seed_cmd[e] (NULL = all -1):
 *    >= 0 : set_seed(s) + reset(keep_seed=True)   -> regenerate with base seed s
 *    -1   : reset(keep_seed=True)                  -> same data, RNG streams rewound
 *    -2   : reset(keep_seed=False)                 -> new seed = route_init.randint(0,4095)
 * Route unrolling (cim_data_generator.py:18-115), order proportion (parsers.py:57-106),
 * MT19937 seeding (sim_random.py:35-63) and frame initialisation all run on the device.
 */
int mrx_cim_reset(mrx_handle h, const int64_t* d_seed_cmd, const uint8_t* d_env_mask, void* stream);

/*
 * Replaces Env.step(action) in Sequential decision mode (core.py:92-133, 317-381) for every
 * env whose d_env_mask byte is non-zero (NULL = all): applies the actions to the pending
 * decision event (cim/business_engine.py:708-748), then advances ticks (…:122-224 and the
 * handlers :448-706) until the next decision event or the end of the episode.
 *   d_actions   int32 [n_envs][A][4] = (vessel_idx, port_idx, quantity, MRX_ACTION_*)
 *   d_n_actions int32 [n_envs] number of valid actions per env (NULL = 0: action=None)
 *   d_decisions int32 [n_envs][8] = (tick, port_idx, vessel_idx, scope.load, scope.discharge,
 *               early_discharge, frame_index, valid)   valid=0 -> no decision (episode done)
 *   d_metrics   int64 [n_envs][3] = (order_requirements, container_shortage, operation_number)
 *   d_done      uint8 [n_envs]
 * Outputs of masked-out envs are left untouched.
 */
int mrx_cim_step(mrx_handle h, const int32_t* d_actions, const int32_t* d_n_actions,
                 const uint8_t* d_env_mask, int32_t* d_decisions, int64_t* d_metrics,
                 uint8_t* d_done, void* stream);

/*
 * How mrx_cim_step / mrx_cim_step_joint launch their work (no reference counterpart: the reference steps one env per
 * process, vector_env/env_process.py:26-67; this is pure scheduling — results are identical in every mode):
 *   1  one workgroup (one wave) per env in env order; the env's private header is read first to pick its path;
 *   2  the same kernels over the ORDER LIST of the step: every env records, at the end of its step, whether its next step
 *      runs a tick (full path: LDS-staged state, ~25 us) or only answers another decision of the same tick (fast path out
 *      of HBM); a small kernel (mrx_k_cim_schedule) puts the full-path envs first, so the long waves start first, the
 *      short ones fill the tail of the launch, and a full-path wave skips the header round trip;
 *   4  the step split in two kernels: the fast-path envs one per LANE in a small kernel without LDS, then the full-path list
 *      walked by as many workgroups as the device holds at once;
 *   5  the fast-path kernel of 4, then one workgroup per entry of the full-path list as in 2 (fast-hinted envs never occupy an
 *      LDS-carrying workgroup);
 *   0  automatic (default): 2 (measured fastest on MI355X at the benchmark's batch sizes).
 * (3 was round 2's persistent pipelined kernel — measured slower than 2 and removed; the value is accepted and means 2.)
 * Returns the mode the next step will actually use (>= 1), or a negative mrx_status.
 */
int mrx_cim_set_step_mode(mrx_handle h, int mode);

/*
 * Fuses an agent's per-decision snapshot slices into mrx_cim_step (Sequential mode): after this call every step also writes,
 * for each stepped env that pauses at a new decision,
 *   d_obs_ports  float64 [n_envs][n_ports][n_port_attrs] = snapshot_list["ports"][decision frame :: port_attrs]
 *   d_obs_vessel float64 [n_envs][n_vessel_attrs]        = snapshot_list["vessels"][decision frame : decision vessel : vessel_attrs]
 * i.e. what examples/cim/rl/env_sampler.py:21-31 and examples/hello_world read through frame.pyx:754-801 right after
 * Env.step returns — values identical to mrx_cim_query on decisions[:, 6] (the decision's frame is the live frame).
 * Attribute ids as mrx_cim_attr_id; single-slot attributes only, at most 8 each; HOST arrays (copied).  Rows of envs
 * that did not get a new decision (episode over, masked out) are left untouched.  n = 0 switches a part off.
 * The two buffers are engine state between steps: a step that stays inside the current tick (another vessel's decision)
 * only patches the cells its action changed, so do not write to the buffers.  The call may come at any time (it drains the
 * device): the next step of every env takes the full path and writes its whole block.
 */
int mrx_cim_set_observation(mrx_handle h, const int32_t* port_attrs, int n_port_attrs, const int32_t* vessel_attrs,
                            int n_vessel_attrs, double* d_obs_ports, double* d_obs_vessel);

/*
 * Per-attribute retention next to the snapshot ring: from this call on every snapshot the engine takes
 * (business_engine.py:215, core.py:376-378) also stores the listed port attributes of its frame into
 *   d_hist int32 [n_envs][frames][n_port_attrs][n_ports]     (frame f of env e at ((e * frames + f) * n_port_attrs + a) * n_ports)
 * so that a consumer which needs a few attributes over the WHOLE episode — the CIM RL example's delayed reward reads
 * ports[tick+1 .. tick+99]["fulfillment" | "shortage"] after its rollout loop (examples/cim/rl/env_sampler.py:65-80,
 * rl/rollout/env_sampler.py:516-526) — can run with a ring of a few frames (max_snapshots) instead of the full history
 * (15.4 KB per frame on global_trade.22p vs 176 B here).  Integer attributes only, at most 4; HOST array of ids
 * (mrx_cim_attr_id).  `frames` = rows per env (normally ceil(durations / snapshot_resolution)); frames beyond it are not
 * stored.  The caller owns and zeroes the buffer (rows of frames that were not reached stay as they are); n = 0 switches
 * it off.  Drains the device.
 */
int mrx_cim_set_port_history(mrx_handle h, const int32_t* port_attrs, int n_port_attrs, int32_t* d_hist, int64_t frames);

/*
 * Env.step in DecisionMode.Joint / JointWithSequentialAction (core.py:354-366; engines created with
 * mrx_cim_config.decision_mode 1 / 2): every pending decision event of the tick is reported at once.
 *   d_decisions  int32 [n_envs][n_vessels][8]: one row per pending event in event (= vessel index) order, same columns as
 *                mrx_cim_step; unused rows have valid = 0 (row 0 still carries tick and frame_index).  Rows are always
 *                evaluated on the live state; the reference re-yields the same DecisionEvent objects, which cache
 *                action_scope at their first read — keep the first row seen per (tick, vessel) to reproduce that.
 *   d_actions    int32 [n_envs][A][4]: the answered events' actions, flattened in event order (they are applied in this order,
 *                exactly as the reference runs each event's action list when the event is popped); d_n_actions int32 [n_envs]
 *   d_n_answered int32 [n_envs] (JointWithSequentialAction): how many of the pending events were answered; the others
 *                stay pending and are reported again.  NULL / negative = all.  In Joint mode unanswered events are
 *                finished without an action (core.py:364-366), so the value is ignored.
 */
int mrx_cim_step_joint(mrx_handle h, const int32_t* d_actions, const int32_t* d_n_actions, const int32_t* d_n_answered,
                       const uint8_t* d_env_mask, int32_t* d_decisions, int64_t* d_metrics, uint8_t* d_done, void* stream);

/*
 * Replaces snapshot_list[node][ticks:nodes:attrs] (frame.pyx:754-801, np_backend.pyx:520-549).
 *   node_type 0 = ports, 1 = vessels, 2 = matrices
 *   d_ticks   int32 frame indices: one row [nt] shared by all envs (ticks_per_env = 0), or one row per env with a
 *             row stride of `ticks_per_env` int32 elements (= nt for a dense [n_envs][nt] array; 8 with d_ticks =
 *             d_decisions + 6 slices "the frame of my pending decision" straight out of mrx_cim_step's output)
 *   d_nodes   int32 node indices (device): one row [nn] shared by all envs (nodes_per_env = 0) or one row per env
 *             with a row stride of `nodes_per_env` elements;  attrs int32 [na<=16] attribute ids (HOST array,
 *             mrx_cim_attr_id; copied into the kernel arguments)
 *   d_out     float64 [n_envs][nt][nn][sum(slots)] — flat order tick -> node -> attr -> slot,
 *             zeros for frame indices not held by the ring (np_backend.pyx:541-545).
 */
int mrx_cim_query(mrx_handle h, int node_type, const int32_t* d_ticks, int nt, int ticks_per_env,
                  const int32_t* d_nodes, int nn, int nodes_per_env, const int32_t* attrs, int na, double* d_out,
                  void* stream);

/*
 * Utility: the reference's hello-world random agent (examples/hello_world/cim/hello.py:22-37) as a
 * counter-based device policy, so rollouts need no host round trip: for every env with a valid
 * decision writes one legal action into d_actions[e][0] (and d_n_actions[e] = 1, else 0) from
 * hash(seed[e], step) — or, with step < 0, from hash(seed[e], tick, vessel) of the decision itself, so the call can
 * sit in a captured hipGraph; adds the number of valid decisions to *d_counter (may be NULL).
 */
int mrx_cim_random_policy(mrx_handle h, const int32_t* d_decisions, int64_t step, int32_t* d_actions,
                          int32_t* d_n_actions, uint64_t* d_counter, void* stream);
/*
 * The same agent ANSWERED INSIDE mrx_cim_step (Sequential mode): with mode 1 every step writes, for each env it leaves at a new
 * decision, the action mrx_cim_random_policy(step = key) would write for that decision into d_actions[e][0] / d_n_actions[e]
 * (0 for an env whose episode is over) and adds 1 to d_counts[e] (int32 [n_envs], may be NULL) — so a rollout with this agent is ONE
 * launch per batch step: pass the same d_actions / d_n_actions to the next mrx_cim_step.  `next_key` >= 0: the key of the first
 * answering step; every mrx_cim_step call that follows uses the next integer (call it again after a reset to restart the count);
 * < 0: every draw is keyed on its decision's (tick, vessel), as mrx_cim_random_policy(step < 0).  mode 0 switches it off.  Runtime
 * configuration (not part of a specialised plan); the buffers must stay valid while it is on.
 */
int mrx_cim_set_device_agent(mrx_handle h, int mode, int32_t* d_actions, int32_t* d_n_actions, int32_t* d_counts, int64_t next_key);

/*
 * Plan-specialised step kernels.  The generic kernels read the plan's ~65 integer dimensions / layout offsets from the kernel
 * arguments; a code object built from maro_amd/csrc/cim_spec.hip with those values as macros (the text this function returns,
 * saved as cim_spec_dims.h) has them as compile-time constants: 40 % fewer VGPRs, a third of the SGPR spill traffic, +12 % env-steps/s.
 *   mrx_cim_plan_defines       host only, no device needed: writes the "#define MRXC_<field> <value>" text of the plan that
 *                              (topo, cfg) produce — n_envs does not matter — and of the fused observation's configuration
 *                              (the attribute lists of mrx_cim_set_observation; 0 attributes = off) into buf; returns the
 *                              bytes needed incl. the NUL.  mrx_cim_set_observation with another configuration drops a loaded
 *                              code object (generic kernels again) until one built for the new text is loaded
 *   mrx_cim_load_step_kernels  loads a gfx950 code object (hipcc --genco of cim_spec.hip: the step-kernel pair of the plan's
 *                              order mode, mrx_k_cim_reset, mrx_k_cim_order_table) and uses it for every later mrx_cim_step* /
 *                              mrx_cim_reset; `defines` must equal the handle's own text, observation included (checked)
 */
int64_t mrx_cim_plan_defines(const mrx_cim_topology* topo, const mrx_cim_config* cfg, const int32_t* obs_port_attrs, int n_obs_port_attrs,
                             const int32_t* obs_vessel_attrs, int n_obs_vessel_attrs, char* buf, int64_t len);
int mrx_cim_load_step_kernels(mrx_handle h, const void* image, int64_t bytes, const char* defines);
/* Tools: copy out (and optionally zero) a __device__ global of the loaded code object — e.g. the phase-cycle counters
 * g_mrx_prof of a code object built with -DMRX_PROFILE_PHASES (tools/phase_profile.py --specialized). */
int mrx_cim_read_kernel_global(mrx_handle h, const char* name, void* out, int64_t bytes, int reset);

/*
 * On-device action selection of the CIM RL example — for every env with a valid decision:
 *   state   = CIMEnvSampler._get_global_and_agent_state_impl (examples/cim/rl/env_sampler.py:15-31):
 *             ports[ticks : [port] + future_stop_list : port_attrs] over ticks max(0, tick - rt), rt in range(look_back - 1),
 *             then vessels[tick : vessel : vessel_attrs], float32
 *   q       = the deciding port's own network, MyQNet (examples/cim/rl/algorithms/dqn.py:13-52) built from FullyConnected
 *             blocks (maro/rl/model/fc_block.py:72-133) in eval mode: BatchNorm folded into the linear maps, LeakyReLU
 *             between layers, no activation on the top layers (head=True), q = adv - mean(adv) + v; exact-f32 MFMA
 *   action  = argmax q (first maximum), translated as CIMEnvSampler._translate_to_env_action (env_sampler.py:33-64)
 * The network is a chain of dense layers dims[0] -> ... -> dims[n_layers]; a dueling net folds its two heads into the last
 * two layers (hidden layers side by side, top layers block-diagonal) so dims[n_layers] = n_actions + 1 (advantages, then V).
 */
enum { MRX_DQN_MAX_LAYERS = 8, MRX_DQN_MAX_WIDTH = 256, MRX_DQN_MAX_ACTIONS = 32 };
typedef struct mrx_cim_dqn_model {
  int32_t n_nets;                       /* one network per port: must equal n_ports */
  int32_t n_layers;                     /* 1..MRX_DQN_MAX_LAYERS dense layers */
  int32_t dims[MRX_DQN_MAX_LAYERS + 1]; /* dims[0] = state_dim; every width <= MRX_DQN_MAX_WIDTH */
  int32_t dueling;                      /* 1: dims[n_layers] = n_actions + 1, 0: = n_actions */
  int32_t n_actions;                    /* <= MRX_DQN_MAX_ACTIONS */
  float negative_slope;                 /* LeakyReLU slope (torch default 0.01) applied after every layer but the last */
  float epsilon;                        /* > 0: epsilon-greedy, counter-based on (env seed, tick, vessel); 0 = greedy */
  int32_t look_back;                    /* state_shaping_conf["look_back"] (examples/cim/rl/config.py) */
  int32_t n_port_attrs, port_attrs[8];  /* attribute ids as mrx_cim_attr_id */
  int32_t n_vessel_attrs, vessel_attrs[8];
  double action_space[MRX_DQN_MAX_ACTIONS]; /* examples/cim/rl/config.py:20-24: fraction per action index, negative = load */
  const float* d_weights;               /* device: n_nets consecutive blobs of mrx_cim_dqn_net_floats() floats */
} mrx_cim_dqn_model;

/* Floats in one network's packed blob (weights padded and permuted for the MFMA operand loads, then biases); < 0 on error. */
int64_t mrx_cim_dqn_net_floats(const mrx_cim_dqn_model* m);
/* HOST helper: packs one network.  weights[l] row-major float32 [dims[l]][dims[l+1]] (= torch Linear.weight transposed),
 * biases[l] float32 [dims[l+1]]; out = host buffer of mrx_cim_dqn_net_floats(m) floats (copy it to d_weights + net * that). */
int mrx_cim_dqn_pack_net(const mrx_cim_dqn_model* m, const float* const* weights, const float* const* biases, float* out);
/* Bytes of device scratch mrx_cim_dqn_act needs (per-port env lists and their counters).  The caller zeroes it ONCE after
 * allocation; every call leaves the counters zero again.  One scratch per handle and stream. */
int64_t mrx_cim_dqn_scratch_bytes(mrx_handle h);
/*
 *   d_decisions  int32 [n_envs][8] as written by mrx_cim_step (Sequential mode)
 *   d_actions    int32 [n_envs][A][4], d_n_actions int32 [n_envs]: ready for the next mrx_cim_step
 *   d_q          float32 [n_envs][n_actions] or NULL; d_state float32 [n_envs][state_dim] or NULL; d_choice int32 [n_envs]
 *                or NULL (the chosen action index) — rows of envs without a valid decision are left untouched
 *   d_counter    NULL, or a device counter the number of answered decisions is added to (as mrx_cim_random_policy)
 */
int mrx_cim_dqn_act(mrx_handle h, const mrx_cim_dqn_model* m, const int32_t* d_decisions, void* d_scratch, int32_t* d_actions,
                    int32_t* d_n_actions, float* d_q, float* d_state, int32_t* d_choice, uint64_t* d_counter, void* stream);

/*
 * One step of the batched EnvSampler's transition cache (replaces the per-env Python bookkeeping of
 * maro/rl/rollout/env_sampler.py:386-410, 472-512: `_append_cache_element` — the element of this decision, the previous
 * element's next state, the deciding agent's previous element's next agent state and terminal flag).  Call it between the
 * policy (mrx_cim_dqn_act, which produced d_state / d_choice / d_actions for d_decisions) and mrx_cim_step.  Per env e:
 *   eoe[e] |= done[e] first (d_done: the `done` output of the previous mrx_cim_step, or NULL when the caller keeps eoe current
 *   itself): the end-of-episode flag then needs no launch of its own between the steps;
 *   alive = prev_active[e] && !eoe[e]:  next_state[e][prev_j[e]] = alive ? state[e] : state element prev_j[e] itself (episode over)
 *   active = !eoe[e]:  element number q = count[e] goes to slot j = q & (cap - 1) <- (tick, deciding port, state[e], choice[e],
 *                      first action row), terminal = 0; the port's previous element last[e][port] (an element NUMBER, -1: none;
 *                      slot = number & (cap - 1)) gets next_agent_state = state[e], terminal = 0;
 *                      last[e][port] = q; count[e] += 1; n_actions[e] stays; inactive envs get n_actions[e] = 0
 *   prev_j[e] = j (a slot), prev_active[e] = active; d_interactions[e] += 1 (int64 [n_envs]: interactions each env has performed).
 * The cache is caller-owned device memory, a RING of `cap` slots per env (`cap` a power of two; count[e] only ever grows, the
 * caller tracks how many of the oldest elements it has consumed and must keep count - consumed <= cap): c_tick int32 [n][cap], c_agent / c_action int64
 * [n][cap], c_state / c_next_state / c_next_agent_state [n][cap][state_dim] of float32 (state_f64 = 0) or float64 (1),
 * c_env_action int32 [n][cap][4], c_terminal uint8 [n][cap].  first != 0: no previous step in this call (prev_* are only written).
 */
int mrx_cim_sampler_record(int32_t n_envs, int32_t n_ports, int32_t state_dim, int32_t cap, int32_t max_actions, int32_t state_f64, int32_t first,
                           const int32_t* d_decisions, const float* d_state, const int32_t* d_choice, const int32_t* d_actions, int32_t* d_n_actions,
                           uint8_t* d_eoe, const uint8_t* d_done, int64_t* d_count, int64_t* d_last, int64_t* d_prev_j, uint8_t* d_prev_active,
                           int32_t* c_tick, int64_t* c_agent, void* c_state, int64_t* c_action, int32_t* c_env_action, uint8_t* c_terminal,
                           void* c_next_state, void* c_next_agent_state, int64_t* d_interactions, int32_t device, void* stream);

/*
 * The batched EnvSampler's EMISSION: the loop after AbsEnvSampler.sample's inner loop (maro/rl/rollout/env_sampler.py:514-530:
 * every cached element old enough for its reward window is popped, `_get_reward` evaluated, the experience appended) together with
 * the CIM example's reward (examples/cim/rl/env_sampler.py:65-80), in one launch: for each of the n_rows envs d_rows[r] (NULL:
 * env r) the oldest d_n_emit[r] elements of its transition ring (element numbers d_tail[r] ..., slot = number & (cap - 1); the
 * cache arrays of mrx_cim_sampler_record) are copied to rows d_out_offset[r] ... of the compact outputs, in age order, with
 *   reward = float32( ff * sum_k decay[k] * fulfillment(port, tick + 1 + k) - sf * sum_k decay[k] * shortage(port, tick + 1 + k) ),
 * k = 0 .. window - 1, summed in float64, read from d_port_history = the int32 [n_envs][frames][2][n_ports] array
 * mrx_cim_set_port_history(fulfillment, shortage) maintains (ticks >= frames contribute zeros: the reference's snapshot padding).
 * The caller advances its tail afterwards.  o_state / o_next_state / o_next_agent_state [K][state_dim] (float32, or float64 when
 * state_f64), o_action int64 [K], o_env_action int32 [K][4], o_reward float32 [K], o_terminal uint8 [K], o_env_id / o_tick /
 * o_agent int32 [K], K = sum of d_n_emit.
 */
int mrx_cim_sampler_emit(int32_t n_rows, int32_t n_ports, int32_t state_dim, int32_t cap, int32_t frames, int32_t window, int32_t state_f64,
                         double fulfillment_factor, double shortage_factor, const double* d_decay, const int64_t* d_rows, const int64_t* d_tail,
                         const int64_t* d_n_emit, const int64_t* d_out_offset, const int32_t* d_port_history, const int32_t* c_tick,
                         const int64_t* c_agent, const void* c_state, const int64_t* c_action, const int32_t* c_env_action, const uint8_t* c_terminal,
                         const void* c_next_state, const void* c_next_agent_state, void* o_state, int64_t* o_action, int32_t* o_env_action,
                         float* o_reward, void* o_next_state, void* o_next_agent_state, uint8_t* o_terminal, int32_t* o_env_id, int32_t* o_tick,
                         int32_t* o_agent, int32_t device, void* stream);

/*
 * The batched EnvSampler's per-env transition cache — `_trans_cache` / `_agent_last_index` of AbsEnvSampler
 * (maro/rl/rollout/env_sampler.py:404-410, 438-537) for n_envs envs at once — as one argument block: caller-owned device arrays,
 * the same ones mrx_cim_sampler_record / mrx_cim_sampler_emit take one by one.  An env's cache is a ring of `cap` slots (a power of
 * two): element number q lives in slot q & (cap - 1); d_head counts the elements appended, d_tail the ones emitted or dropped.
 */
typedef struct mrx_cim_sampler_cache {
  int32_t n_envs, n_ports, state_dim, cap;
  int32_t state_f64;           /* element type of c_state / c_next_state / c_next_agent_state: 0 float32, 1 float64 */
  int32_t window;              /* reward window in ticks = reward_eval_delay (examples/cim/rl/config.py: time_window) */
  int32_t frames, reserved0;   /* frames per env of d_port_history */
  double fulfillment_factor, shortage_factor;
  const double* d_decay;       /* [window] time_decay ^ k */
  uint8_t* d_eoe;              /* [n] _end_of_episode */
  int64_t *d_head, *d_tail;    /* [n] */
  int64_t* d_last;             /* [n][n_ports] element number of each agent's last element (-1: none) */
  int64_t* d_prev_j;           /* [n] slot written by the env's previous interaction */
  uint8_t* d_prev_active;      /* [n] that element still waits for its next_state */
  int64_t* d_interactions;     /* [n] interactions performed */
  int32_t* c_tick;             /* [n][cap] */
  int64_t* c_agent;            /* [n][cap] */
  void* c_state;               /* [n][cap][state_dim] */
  int64_t* c_action;           /* [n][cap] model action */
  int32_t* c_env_action;       /* [n][cap][4] */
  uint8_t* c_terminal;         /* [n][cap] */
  void* c_next_state;          /* [n][cap][state_dim] */
  void* c_next_agent_state;    /* [n][cap][state_dim] */
  int32_t* d_port_history;     /* int32 [n][frames][2][n_ports]: mrx_cim_set_port_history(fulfillment, shortage) */
} mrx_cim_sampler_cache;

/*
 * n_steps interactions of every env, enqueued in ONE call (the inner loop of AbsEnvSampler.sample,
 * maro/rl/rollout/env_sampler.py:484-511): per interaction
 *   mrx_cim_dqn_act (binning + forward kernel) with the transition-cache update of mrx_cim_sampler_record folded into those two
 *   launches (the forward kernel appends each deciding env's transition from the state row it holds in LDS; the binning launch
 *   retires envs whose episode is over: eoe |= done, the last element's next_state = its own state), then mrx_cim_step
 * — three launches, no host round trip, no host work between interactions.  Envs whose episode ends sit the remaining steps out
 * (the caller rolls them over between calls).  d_prev_active must be zero for envs without a pending element (after a reset).
 * Results are those of the three separate calls, step by step.
 */
int mrx_cim_collect_steps(mrx_handle h, const mrx_cim_dqn_model* m, void* d_scratch, const mrx_cim_sampler_cache* cache, int32_t* d_actions,
                          int32_t* d_n_actions, int32_t* d_decisions, int64_t* d_metrics, uint8_t* d_done, int32_t n_steps, void* stream);

/*
 * The end of a batched EnvSampler call (env_sampler.py:512-530) for every env, on the device.
 * mrx_cim_sampler_finalize: eoe |= d_done; an env whose episode is over gets its last element's next_state (its own state); a
 *   paused env's current tick gets its port-history row from the live frame (the pre-decision snapshot, core.py:345);
 *   d_n_emit[e] = the env's cached elements, oldest first, with tick <= env.tick - window; d_out_offset = their exclusive prefix
 *   over the envs; d_info int64 [4] = (experiences to emit, most elements any env keeps cached afterwards, envs at the end of
 *   their episode, 0) — the call's one read-back.
 * mrx_cim_sampler_emit_all: every env's d_n_emit[e] oldest elements -> rows d_out_offset[e] ... of the outputs (those of
 *   mrx_cim_sampler_emit, K = d_info[0] rows) with the delayed reward; `_append_cache_element(None)` is applied on the way (an
 *   emitted element that is still its agent's last one: terminal = end_of_episode, next_agent_state = its own state,
 *   env_sampler.py:404-410); then the emitted prefix is popped (d_tail += d_n_emit, d_last entries below the new tail = -1).
 */
int mrx_cim_sampler_finalize(mrx_handle h, const mrx_cim_sampler_cache* cache, const uint8_t* d_done, int64_t* d_n_emit, int64_t* d_out_offset,
                             int64_t* d_info, void* stream);
int mrx_cim_sampler_emit_all(mrx_handle h, const mrx_cim_sampler_cache* cache, const int64_t* d_n_emit, const int64_t* d_out_offset, void* o_state,
                             int64_t* o_action, int32_t* o_env_action, float* o_reward, void* o_next_state, void* o_next_agent_state,
                             uint8_t* o_terminal, int32_t* o_env_id, int32_t* o_tick, int32_t* o_agent, void* stream);

/* Attribute name -> id and slot count for a node type; returns -1 for an unknown attribute
 * (reference raises BackendsInvalidAttributeException, frame.pyx:786-790). */
int mrx_cim_attr_id(int node_type, const char* name);
int mrx_cim_attr_slots(mrx_handle h, int node_type, int attr_id);

const char* mrx_last_error(void);
const char* mrx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MARO_AMD_H_ */
