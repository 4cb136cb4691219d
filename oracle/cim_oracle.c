/*
 * ORACLE — test infrastructure only.  Never imported, linked or executed by the product
 * (maro_amd/, libmaro_amd.so); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker / the timed CPU baseline.
 *
 * A single-environment, sequential, event-driven CPU restatement of the reference CIM
 * simulator path:  Env (maro/simulator/core.py) + EventBuffer (maro/event_buffer/) +
 * CimBusinessEngine (maro/simulator/scenarios/cim/business_engine.py) + the synthetic data
 * generator/containers (maro/data_lib/cim/) + SimRandom (maro/simulator/utils/sim_random.py) +
 * the NumPy snapshot list (maro/backends/np_backend.pyx).  It deliberately keeps the
 * reference's *structure* (per-tick event lists, cascade/immediate events, decision pause),
 * whereas the GPU engine uses a fixed-phase tick loop — so agreement between the two is
 * evidence, not tautology.  Each function cites the reference file:line it follows.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks this file against vectors produced by
 * the real reference (built from /root/reference; generator: oracle/gen_golden.py) and against
 * the reference's own known answers (tests/cim/test_cim_scenario.py, docs ...rst:152-165,293-303).
 */
#define _POSIX_C_SOURCE 199309L /* clock_gettime (cim_oracle_bench) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/maro_amd.h"
#include "mt19937.h"

/* ------------------------------------------------------------------ SimRandom */
/* maro/simulator/utils/sim_random.py:10-100 — ordered registry of named streams; stream i
 * (creation order) is seeded seed+i.  Keys: maro/data_lib/cim/utils.py:8-11. */
enum { K_ORDER_INIT = 0, K_ROUTE_INIT, K_ORDER_NUM, K_BUFFER_TICK, K_COUNT };

typedef struct {
  mt_state rand[K_COUNT];
  int created[K_COUNT];   /* creation index, -1 = not created */
  int64_t seed_of[K_COUNT]; /* _seed_dict */
  int n_created;
  int64_t seed; /* _seed */
} sim_random;

static void simrand_clear(sim_random* r) { /* sim_random.py:89-93 (time-based seed is always overridden before use) */
  for (int k = 0; k < K_COUNT; k++) r->created[k] = -1;
  r->n_created = 0;
  r->seed = 0;
}
static void simrand_seed(sim_random* r, int64_t s) { /* sim_random.py:35-54 */
  r->seed = s;
  for (int k = 0; k < K_COUNT; k++)
    if (r->created[k] >= 0) {
      r->seed_of[k] = s + r->created[k];
      mt_seed_int(&r->rand[k], r->seed_of[k]);
    }
}
static mt_state* simrand_get(sim_random* r, int key) { /* sim_random.py:56-71 */
  if (r->created[key] < 0) {
    r->seed_of[key] = r->seed + r->n_created;
    r->created[key] = r->n_created++;
    mt_seed_int(&r->rand[key], r->seed_of[key]);
  }
  return &r->rand[key];
}
static void simrand_reset_seed(sim_random* r, int key) { /* sim_random.py:73-87 */
  if (r->created[key] < 0) { simrand_get(r, key); return; }
  mt_seed_int(&r->rand[key], r->seed_of[key]);
}

/* data_lib/cim/utils.py:30-41 */
static double apply_noise(double value, double noise, mt_state* rng) { return value + mt_uniform(rng, -noise, noise); }

/* ------------------------------------------------------------------ events */
/* maro/simulator/scenarios/cim/events.py:7-21 + maro/event_buffer/maro_events.py:8-12 */
enum { EV_DEPARTURE, EV_RETURN_FULL, EV_RETURN_EMPTY, EV_ORDER, EV_ARRIVAL, EV_LOAD_FULL,
       EV_DISCHARGE_FULL, EV_PENDING_DECISION, EV_TAKE_ACTION };
enum { ST_PENDING, ST_EXECUTING, ST_FINISHED };

typedef struct event {
  int type, tick, state;
  int a, b, c, d;          /* payload ints (meaning depends on type) */
  int32_t* actions;        /* TAKE_ACTION payload: n x 4 */
  int n_actions;
  struct event* next;
  struct event *imm_head, *imm_tail; /* cascade immediate events, event.py:110-144 */
  int imm_count;
} event;

typedef struct { event *head, *tail; } ev_list;

/* ------------------------------------------------------------------ frame layout (oracle-private, AoS like np_backend) */
enum { PA_CAPACITY, PA_EMPTY, PA_FULL, PA_ON_SHIPPER, PA_ON_CONSIGNEE, PA_SHORTAGE, PA_ACC_SHORTAGE,
       PA_BOOKING, PA_ACC_BOOKING, PA_FULFILLMENT, PA_ACC_FULFILLMENT, PA_TRANSFER_COST, PA_COUNT };
enum { VA_CAPACITY, VA_EMPTY, VA_FULL, VA_REMAINING_SPACE, VA_EARLY_DISCHARGE, VA_IS_PARKING, VA_LOC_PORT_IDX,
       VA_ROUTE_IDX, VA_LAST_LOC_IDX, VA_NEXT_LOC_IDX, VA_PAST_STOP_LIST, VA_PAST_STOP_TICK_LIST,
       VA_FUTURE_STOP_LIST, VA_FUTURE_STOP_TICK_LIST, VA_COUNT };
enum { MA_FULL_ON_PORTS, MA_FULL_ON_VESSELS, MA_VESSEL_PLANS, MA_COUNT };

typedef struct { int index, arrival_tick, leave_tick, port_idx, vessel_idx; } stop_t; /* entities.py:20-26 */

typedef struct cim_oracle {
  /* topology (deep copy) */
  mrx_cim_topology t;
  void* topo_blob;
  int P, V, past_n, future_n, vw; /* vw = words per vessel row */
  int start_tick, durations, max_tick, resolution, max_snapshots_arg;
  /* data collection (cim_data_generator.py:118-205) */
  sim_random rnd;
  stop_t** stops; int* n_stops; int* stops_cap;
  int* vessel_period;
  int32_t* order_proportion; /* [max_tick] */
  int64_t data_seed;
  int is_need_reset_seed;             /* cim_data_container.py:74, 246-252 */
  int64_t wrapper_random_seed; int wrapper_has_seed; int re_init_flag; /* cim_data_container_helpers.py:46-73 */
  /* frame + snapshots (np_backend.pyx) */
  int fw; int32_t* frame; /* row 0 */
  int S; int32_t* snaps;  /* rows 1..S */
  int* slot_fi; int n_stored; int cur_index; /* _index2tick / _cur_index */
  int* fi_order; int n_fi_order;             /* insertion order of stored frame indices */
  int off_ports, off_vessels, off_fop, off_fov, off_plans;
  /* event buffer */
  ev_list* lists; int n_lists;
  /* env state (core.py) */
  int tick; int waiting_action; int finished; int stop_iteration;
  event* pending0;
  int64_t total_operate_num;
  int error;
} cim_oracle;

#define PORT(o, i, a) ((o)->frame[(o)->off_ports + (i) * PA_COUNT + (a)])
#define VES(o, v, a) ((o)->frame[(o)->off_vessels + (v) * (o)->vw + (a)])
#define FOP(o, s, d) ((o)->frame[(o)->off_fop + (s) * (o)->P + (d)])
#define FOV(o, v, p) ((o)->frame[(o)->off_fov + (v) * (o)->P + (p)])
#define PLAN(o, v, p) ((o)->frame[(o)->off_plans + (v) * (o)->P + (p)])

static int ves_list_off(const cim_oracle* o, int attr) {
  switch (attr) {
    case VA_PAST_STOP_LIST: return 10;
    case VA_PAST_STOP_TICK_LIST: return 10 + o->past_n;
    case VA_FUTURE_STOP_LIST: return 10 + 2 * o->past_n;
    case VA_FUTURE_STOP_TICK_LIST: return 10 + 2 * o->past_n + o->future_n;
    default: return attr;
  }
}

/* derived attributes: port.py:88-97, vessel.py:113-120 */
static void port_update_fulfillment(cim_oracle* o, int p) { PORT(o, p, PA_FULFILLMENT) = PORT(o, p, PA_BOOKING) - PORT(o, p, PA_SHORTAGE); }
static int vessel_total_space(const cim_oracle* o, int v) { return (int)floor((double)o->t.vessel_capacity[v] / (double)o->t.container_volume); }
static void vessel_update_remaining(cim_oracle* o, int v) { VES(o, v, VA_REMAINING_SPACE) = vessel_total_space(o, v) - VES(o, v, VA_FULL) - VES(o, v, VA_EMPTY); }

/* ------------------------------------------------------------------ topology deep copy */
static void* dup_arr(const void* p, size_t bytes) { void* q = malloc(bytes ? bytes : 1); memcpy(q, p, bytes); return q; }
static void topo_copy(cim_oracle* o, const mrx_cim_topology* s) {
  mrx_cim_topology* t = &o->t; *t = *s;
  int P = s->n_ports, V = s->n_vessels, R = s->n_routes, NT = s->n_targets, NR = s->n_route_points;
#define DUP(f, n, T) t->f = (const T*)dup_arr(s->f, sizeof(T) * (size_t)(n))
  DUP(order_dist, s->period, double);
  DUP(port_capacity, P, int32_t); DUP(port_init_empty, P, int32_t);
  DUP(empty_return_base, P, double); DUP(empty_return_noise, P, double);
  DUP(full_return_base, P, double); DUP(full_return_noise, P, double);
  DUP(source_base, P, double); DUP(source_noise, P, double);
  DUP(target_offset, P + 1, int32_t); DUP(target_port, NT, int32_t); DUP(target_base, NT, double); DUP(target_noise, NT, double);
  DUP(route_offset, R + 1, int32_t); DUP(route_port, NR, int32_t); DUP(route_dist, NR, double);
  DUP(vessel_capacity, V, int32_t); DUP(vessel_init_empty, V, int32_t); DUP(vessel_route, V, int32_t); DUP(vessel_start_offset, V, int32_t);
  DUP(vessel_speed, V, double); DUP(vessel_speed_noise, V, double); DUP(vessel_duration, V, double); DUP(vessel_duration_noise, V, double);
  if (s->data_mode) { /* dump folder / real data files: cim_data_loader.py:360-450 */
    DUP(fixed_n_stops, V, int32_t); DUP(fixed_vessel_period, V, int32_t);
    DUP(fixed_stops_arrival, (size_t)V * s->fixed_max_stops, int32_t); DUP(fixed_stops_leave, (size_t)V * s->fixed_max_stops, int32_t);
    if (s->data_mode == 1) DUP(fixed_order_prop, s->data_max_tick, int32_t);
    if (s->data_mode == 2) DUP(fixed_orders, (size_t)s->data_max_tick * NT, int32_t);
  }
#undef DUP
}
static int route_len(const cim_oracle* o, int v) { int r = o->t.vessel_route[v]; return o->t.route_offset[r + 1] - o->t.route_offset[r]; }
static int route_port_at(const cim_oracle* o, int v, int pos) { int r = o->t.vessel_route[v]; return o->t.route_port[o->t.route_offset[r] + pos]; }
static double route_dist_at(const cim_oracle* o, int v, int pos) { int r = o->t.vessel_route[v]; return o->t.route_dist[o->t.route_offset[r] + pos]; }

/* ------------------------------------------------------------------ data generation */
/* parsers.py:57-106 parse_global_order_proportion (np.interp already applied by the host parser) */
static void gen_order_proportion(cim_oracle* o) {
  const mrx_cim_topology* t = &o->t;
  /* NB: the wrapper always passes start_tick=0 (cim_data_container_helpers.py:22,57) */
  for (int tk = 0; tk < o->max_tick; tk++) {
    double orders = t->order_dist[tk % t->period];
    int32_t val = 0;
    if (orders != 0) {
      if (t->sample_noise != 0) orders = apply_noise(orders, t->sample_noise, simrand_get(&o->rnd, K_ORDER_INIT));
      double c = orders < 1 ? orders : 1; if (c < 0) c = 0; /* clip(0,1,orders) utils.py:14-27 */
      val = (int32_t)floor(c * (double)t->total_containers);
    }
    o->order_proportion[tk] = val;
  }
}

/* cim_data_generator.py:18-115 _extend_route */
static void extend_route(cim_oracle* o) {
  const mrx_cim_topology* t = &o->t;
  for (int v = 0; v < o->V; v++) {
    int L = route_len(o, v);
    int loc = t->vessel_start_offset[v];
    double speed = t->vessel_speed[v], sn = t->vessel_speed_noise[v];
    double duration = t->vessel_duration[v], dn = t->vessel_duration_noise[v];
    int tick = 0, period_no_noise = 0, extra = 0, idx = 0;
    o->n_stops[v] = 0;
    while (extra <= t->future_stop_number) {
      int port = route_port_at(o, v, loc);
      int parking = (int)ceil(apply_noise(duration, dn, simrand_get(&o->rnd, K_ROUTE_INIT)));
      if (parking <= 0) o->error |= 4; /* assert parking_duration > 0, :57 */
      if (o->n_stops[v] == o->stops_cap[v]) {
        o->stops_cap[v] = o->stops_cap[v] ? o->stops_cap[v] * 2 : 64;
        o->stops[v] = (stop_t*)realloc(o->stops[v], sizeof(stop_t) * (size_t)o->stops_cap[v]);
      }
      stop_t s = {idx, tick, tick + parking, port, v};
      o->stops[v][o->n_stops[v]++] = s;
      double dist = route_dist_at(o, v, loc);
      double noised_speed = apply_noise(speed, sn, simrand_get(&o->rnd, K_ROUTE_INIT));
      int sailing = (int)ceil(dist / noised_speed);
      tick += parking + sailing;
      int whole = (int)(duration + ceil(dist / speed));
      period_no_noise += (o->n_stops[v] <= L) ? whole : 0;
      loc = (loc + 1) % L;
      extra += (tick > o->max_tick) ? 1 : 0;
      idx++;
    }
    o->vessel_period[v] = period_no_noise;
  }
}

/* cim_data_generator.py:118-205 gen_cim_data (+ data_from_generator) */
static void gen_cim_data(cim_oracle* o, int64_t topology_seed) {
  if (o->t.data_mode) {
    /* data_from_dumps / data_from_files (cim_data_container_helpers.py:79-85, 118-123): the data set is (re)loaded from
     * its folder whatever the wrapper's seed says, then seed(data_collection.seed) re-seeds the registry */
    const mrx_cim_topology* t = &o->t;
    simrand_seed(&o->rnd, t->seed);
    o->data_seed = t->seed;
    for (int tk = 0; tk < o->max_tick; tk++) o->order_proportion[tk] = (t->data_mode == 1 && tk < t->data_max_tick) ? t->fixed_order_prop[tk] : 0;
    for (int v = 0; v < o->V; v++) {
      const int n = t->fixed_n_stops[v], L = route_len(o, v);
      if (o->stops_cap[v] < n) { o->stops_cap[v] = n; o->stops[v] = (stop_t*)realloc(o->stops[v], sizeof(stop_t) * (size_t)n); }
      for (int k = 0; k < n; k++) {
        stop_t st = {k, t->fixed_stops_arrival[(size_t)v * t->fixed_max_stops + k], t->fixed_stops_leave[(size_t)v * t->fixed_max_stops + k],
                     route_port_at(o, v, (t->vessel_start_offset[v] + k) % L), v};
        o->stops[v][k] = st;
      }
      o->n_stops[v] = n;
      o->vessel_period[v] = t->fixed_vessel_period[v];
    }
    o->is_need_reset_seed = 0;
    return;
  }
  simrand_seed(&o->rnd, topology_seed); /* :145 */
  o->data_seed = topology_seed;
  gen_order_proportion(o);              /* :157-162 */
  extend_route(o);                      /* :165-172 */
  o->is_need_reset_seed = 0;            /* fresh CimSyntheticDataContainer */
}

/* ------------------------------------------------------------------ stop wrappers */
/* vessel_future_stops_prediction.py:49-85 */
static void predict_future_stops(const cim_oracle* o, int v, int last_stop_idx, int n, int* ports, int* ticks) {
  double speed = o->t.vessel_speed[v], duration = o->t.vessel_duration[v];
  int L = route_len(o, v);
  int arrival = o->stops[v][last_stop_idx].arrival_tick;
  int last_loc = (o->t.vessel_start_offset[v] + last_stop_idx) % L;
  int k = 0;
  for (int loc = last_loc + 1; loc < last_loc + n + 1; loc++, k++) {
    ports[k] = route_port_at(o, v, loc % L);
    double dist = route_dist_at(o, v, (loc - 1) % L);
    arrival += (int)(duration + ceil(dist / speed));
    ticks[k] = arrival;
  }
}

/* vessel.py:90-111 set_stop_list + vessel_past_stops_wrapper.py:23-38 */
static void set_past_stops(cim_oracle* o, int v, int last_loc_idx, int loc_idx) {
  int n = o->past_n;
  int last_stop_idx = loc_idx + (last_loc_idx == loc_idx ? 0 : -1);
  int start = last_stop_idx - n + 1; if (start < 0) start = 0;
  int cnt = loc_idx - start; if (cnt < 0) cnt = 0;
  if (loc_idx > o->n_stops[v]) cnt = o->n_stops[v] - start;
  if (n == 0) return; /* `if past_stop_list:` is falsy for an empty list */
  int pad = n - cnt;
  int lo = ves_list_off(o, VA_PAST_STOP_LIST), to = ves_list_off(o, VA_PAST_STOP_TICK_LIST);
  for (int i = 0; i < n; i++) {
    if (i < pad) { VES(o, v, lo + i) = -1; VES(o, v, to + i) = -1; }
    else { const stop_t* s = &o->stops[v][start + i - pad]; VES(o, v, lo + i) = s->port_idx; VES(o, v, to + i) = s->arrival_tick; }
  }
}
/* vessel_future_stops_prediction.py:31-37 */
static void set_future_stops(cim_oracle* o, int v, int last_loc_idx, int loc_idx) {
  int n = o->future_n; if (n == 0) return;
  int last_stop_idx = loc_idx + (last_loc_idx == loc_idx ? 0 : -1);
  int ports[64], ticks[64];
  predict_future_stops(o, v, last_stop_idx, n, ports, ticks);
  int lo = ves_list_off(o, VA_FUTURE_STOP_LIST), to = ves_list_off(o, VA_FUTURE_STOP_TICK_LIST);
  for (int i = 0; i < n; i++) { VES(o, v, lo + i) = ports[i]; VES(o, v, to + i) = ticks[i]; }
}
/* vessel_sailing_plan_wrapper.py:24-28 applied at business_engine.py:393-398, 627-632 */
static void apply_planned_stops(cim_oracle* o, int v, int next_loc_idx) {
  int L = route_len(o, v); int ports[64], ticks[64];
  predict_future_stops(o, v, next_loc_idx, L, ports, ticks);
  for (int i = 0; i < L; i++) PLAN(o, v, ports[i]) = ticks[i];
}

/* ------------------------------------------------------------------ event buffer */
static event* ev_new(int tick, int type, int a, int b, int c, int d) {
  event* e = (event*)calloc(1, sizeof(event));
  e->tick = tick; e->type = type; e->state = ST_PENDING; e->a = a; e->b = b; e->c = c; e->d = d;
  return e;
}
static void ev_free_chain(event* e) {
  while (e) { event* n = e->next; ev_free_chain(e->imm_head); free(e->actions); free(e); e = n; }
}
/* event_buffer.py:180-188 insert_event; events for ticks that are never executed are dropped */
static void eb_insert(cim_oracle* o, event* e) {
  if (e->tick < 0 || e->tick >= o->n_lists) { ev_free_chain(e); return; }
  ev_list* l = &o->lists[e->tick];
  e->next = NULL;
  if (l->tail) l->tail->next = e; else l->head = e;
  l->tail = e;
}
/* event.py:110-144 add_immediate_event */
static void ev_add_immediate(event* parent, event* e, int is_head) {
  e->next = NULL;
  if (!parent->imm_head) { parent->imm_head = parent->imm_tail = e; }
  else if (is_head) { e->next = parent->imm_head; parent->imm_head = e; }
  else { parent->imm_tail->next = e; parent->imm_tail = e; }
  parent->imm_count++;
}
static void eb_reset(cim_oracle* o) {
  for (int i = 0; i < o->n_lists; i++) { ev_free_chain(o->lists[i].head); o->lists[i].head = o->lists[i].tail = NULL; }
}

/* ------------------------------------------------------------------ snapshot list (np_backend.pyx:481-518) */
static void take_snapshot(cim_oracle* o, int fi) {
  int target = 0;
  for (int s = 1; s <= o->S; s++) if (o->slot_fi[s] == fi) target = s;
  if (!target) {
    o->cur_index += 1;
    if (o->cur_index >= o->S + 1) o->cur_index = 1;
    target = o->cur_index;
  }
  /* drop the mapping of whatever the slot held, re-append fi at the end of the insertion order */
  if (o->slot_fi[target] >= 0) {
    int old = o->slot_fi[target], w = 0;
    for (int i = 0; i < o->n_fi_order; i++) if (o->fi_order[i] != old) o->fi_order[w++] = o->fi_order[i];
    o->n_fi_order = w;
  }
  memcpy(o->snaps + (size_t)target * o->fw, o->frame, sizeof(int32_t) * (size_t)o->fw);
  o->slot_fi[target] = fi;
  o->fi_order[o->n_fi_order++] = fi;
}
static void snapshots_reset(cim_oracle* o) { /* np_backend.pyx:569-586 */
  o->cur_index = 0; o->n_fi_order = 0;
  for (int s = 0; s <= o->S; s++) o->slot_fi[s] = -1;
  memset(o->snaps, 0, sizeof(int32_t) * (size_t)(o->S + 1) * o->fw);
}
static int frame_index(const cim_oracle* o, int tick) { return (int)floor((double)(tick - o->start_tick) / (double)o->resolution); } /* utils/common.py:81-93 */

/* ------------------------------------------------------------------ business engine */
/* business_engine.py:371-379 */
static void load_departure_events(cim_oracle* o) {
  for (int v = 0; v < o->V; v++)
    for (int k = 0; k < o->n_stops[v]; k++)
      eb_insert(o, ev_new(o->stops[v][k].leave_tick, EV_DEPARTURE, o->stops[v][k].port_idx, v, 0, 0));
}
/* business_engine.py:381-398 */
static void init_vessel_plans(cim_oracle* o) {
  for (int v = 0; v < o->V; v++) {
    int last = VES(o, v, VA_LAST_LOC_IDX), next = VES(o, v, VA_NEXT_LOC_IDX);
    VES(o, v, VA_IS_PARKING) = (last == next) ? 1 : 0;
    VES(o, v, VA_LOC_PORT_IDX) = o->stops[v][last].port_idx;
    set_past_stops(o, v, last, next);
    set_future_stops(o, v, last, next);
    apply_planned_stops(o, v, last);
  }
}
/* business_engine.py:321-356 (_init_nodes / _reset_nodes after frame.reset zeroes everything) */
static void reset_nodes(cim_oracle* o) {
  memset(o->frame, 0, sizeof(int32_t) * (size_t)o->fw); /* np_backend.pyx:379-389 */
  for (int p = 0; p < o->P; p++) { PORT(o, p, PA_CAPACITY) = o->t.port_capacity[p]; PORT(o, p, PA_EMPTY) = o->t.port_init_empty[p]; }
  for (int v = 0; v < o->V; v++) {
    VES(o, v, VA_CAPACITY) = o->t.vessel_capacity[v];
    VES(o, v, VA_ROUTE_IDX) = o->t.vessel_route[v];
    VES(o, v, VA_EMPTY) = o->t.vessel_init_empty[v];
    vessel_update_remaining(o, v);
  }
  for (int i = 0; i < o->V * o->P; i++) o->frame[o->off_plans + i] = -1;
}

/* cim_data_container.py:309-398 _gen_orders; emits ORDER events (business_engine.py:138-143) */
static void gen_orders(cim_oracle* o, int tick, int64_t total_empty) {
  const mrx_cim_topology* t = &o->t;
  if (o->is_need_reset_seed) { /* cim_data_container.py:292-296, 304-307; the real-data container only has the base :250-252 */
    simrand_reset_seed(&o->rnd, K_BUFFER_TICK);
    if (t->data_mode != 2) simrand_reset_seed(&o->rnd, K_ORDER_NUM);
    o->is_need_reset_seed = 0;
  }
  if (t->data_mode == 2) { /* CimRealDataContainer.get_orders :422-443: the tick's orders as listed in the file */
    if (tick >= t->data_max_tick) return;
    for (int p = 0; p < o->P; p++)
      for (int j = t->target_offset[p]; j < t->target_offset[p + 1]; j++) {
        const int q = t->fixed_orders[(size_t)tick * t->n_targets + j];
        if (q > 0) eb_insert(o, ev_new(tick, EV_ORDER, p, t->target_port[j], q, 0));
      }
    return;
  }
  if (tick >= (t->data_mode ? t->data_max_tick : o->max_tick)) return;
  int64_t orders_to_gen = (int64_t)o->order_proportion[tick];
  if (t->order_mode == 1) { /* UNFIXED :327-333 */
    int64_t delta = (int64_t)t->total_containers - total_empty;
    if (orders_to_gen <= delta) return;
    orders_to_gen -= delta;
  }
  int64_t remaining = orders_to_gen;
  int P = o->P;
  double* src = (double*)malloc(sizeof(double) * (size_t)P);
  mt_state* rng = simrand_get(&o->rnd, K_ORDER_NUM);
  for (int p = 0; p < P; p++) src[p] = apply_noise(t->source_base[p], t->source_noise[p], rng);
  double tot = 0; for (int p = 0; p < P; p++) tot += src[p]; /* list_sum_normalize utils.py:44-56 */
  if (tot != 0) for (int p = 0; p < P; p++) src[p] = src[p] / tot;
  double tg[256];
  for (int p = 0; p < P; p++) {
    if (remaining == 0) break;
    int off = t->target_offset[p], n = t->target_offset[p + 1] - off;
    double ts = 0;
    for (int j = 0; j < n; j++) tg[j] = apply_noise(t->target_base[off + j], t->target_noise[off + j], rng);
    for (int j = 0; j < n; j++) ts += tg[j];
    if (ts != 0) for (int j = 0; j < n; j++) tg[j] = tg[j] / ts;
    int64_t cur_port = (int64_t)ceil((double)orders_to_gen * src[p]);
    if (cur_port > remaining) cur_port = remaining;
    remaining -= cur_port;
    if (cur_port > 0) {
      int64_t trem = cur_port;
      for (int j = 0; j < n; j++) {
        int64_t cur = (int64_t)ceil((double)cur_port * tg[j]);
        if (cur > trem) cur = trem;
        trem -= cur;
        if (cur > 0) eb_insert(o, ev_new(tick, EV_ORDER, p, t->target_port[off + j], (int)cur, 0));
      }
    }
  }
  free(src);
}

/* business_engine.py:122-199 */
static void be_step(cim_oracle* o, int tick) {
  int64_t total_empty = 0;
  for (int p = 0; p < o->P; p++) total_empty += PORT(o, p, PA_EMPTY);
  for (int v = 0; v < o->V; v++) total_empty += VES(o, v, VA_EMPTY);
  gen_orders(o, tick, total_empty);
  event* decisions[256]; int nd = 0;
  for (int v = 0; v < o->V; v++) {
    int loc = VES(o, v, VA_NEXT_LOC_IDX);
    const stop_t* s = &o->stops[v][loc];
    if (loc > 0 && s->arrival_tick == tick) {
      eb_insert(o, ev_new(tick, EV_ARRIVAL, s->port_idx, v, 0, 0));
      eb_insert(o, ev_new(tick, EV_LOAD_FULL, s->port_idx, v, 0, 0));
      decisions[nd++] = ev_new(tick, EV_PENDING_DECISION, s->port_idx, v, 0, 0);
      PLAN(o, v, s->port_idx) = s->arrival_tick; /* :194-195 */
    }
  }
  for (int i = 0; i < nd; i++) eb_insert(o, decisions[i]); /* :197-199 */
}

/* port_buffer_tick_wrapper.py:29-35 */
static int buffer_ticks(cim_oracle* o, double base, double noise) {
  return (int)ceil(apply_noise(base, noise, simrand_get(&o->rnd, K_BUFFER_TICK)));
}

static void on_order_generated(cim_oracle* o, event* e) { /* :448-497 */
  int src = e->a, dst = e->b, qty = e->c;
  int execute_qty = qty, src_empty = PORT(o, src, PA_EMPTY);
  PORT(o, src, PA_BOOKING) += execute_qty; port_update_fulfillment(o, src);
  PORT(o, src, PA_ACC_BOOKING) += execute_qty;
  if (src_empty < qty) {
    int shortage = qty - src_empty;
    PORT(o, src, PA_SHORTAGE) += shortage; port_update_fulfillment(o, src);
    PORT(o, src, PA_ACC_SHORTAGE) += shortage;
    execute_qty = src_empty;
  }
  PORT(o, src, PA_EMPTY) -= execute_qty;
  PORT(o, src, PA_ON_SHIPPER) += execute_qty;
  int bt = buffer_ticks(o, o->t.full_return_base[src], o->t.full_return_noise[src]);
  event* r = ev_new(e->tick + bt, EV_RETURN_FULL, src, dst, execute_qty, 0);
  if (bt == 0) ev_add_immediate(e, r, 0); else eb_insert(o, r);
}
static void on_full_return(cim_oracle* o, event* e) { /* :499-522 */
  PORT(o, e->a, PA_ON_SHIPPER) -= e->c;
  PORT(o, e->a, PA_FULL) += e->c;
  FOP(o, e->a, e->b) += e->c;
}
static void on_full_load(cim_oracle* o, event* e) { /* :524-598 */
  int port = e->a, v = e->b;
  int vol = o->t.container_volume, cap = VES(o, v, VA_CAPACITY);
  VES(o, v, VA_LAST_LOC_IDX) = VES(o, v, VA_NEXT_LOC_IDX);
  int remaining_space = cap - VES(o, v, VA_FULL) * vol;
  int acceptable = (int)floor((double)remaining_space / (double)vol);
  /* vessel_reachable_stops_wrapper.py:23-29 — python slicing truncates at the end of the stop list */
  int next = VES(o, v, VA_NEXT_LOC_IDX), L = route_len(o, v);
  for (int k = next + 1; k < next + 1 + L && k < o->n_stops[v]; k++) {
    int np_ = o->stops[v][k].port_idx, arr = o->stops[v][k].arrival_tick;
    int pending = FOP(o, port, np_);
    if (acceptable > 0 && pending > 0) {
      int loaded = pending < acceptable ? pending : acceptable;
      FOP(o, port, np_) = pending - loaded;
      PORT(o, port, PA_FULL) -= loaded;
      VES(o, v, VA_FULL) += loaded; vessel_update_remaining(o, v);
      FOV(o, v, np_) += loaded;
      acceptable -= loaded;
      eb_insert(o, ev_new(arr, EV_DISCHARGE_FULL, v, port, np_, loaded));
    }
  }
  int total = VES(o, v, VA_FULL) + VES(o, v, VA_EMPTY);
  VES(o, v, VA_EARLY_DISCHARGE) = 0;
  if ((int64_t)total * vol > VES(o, v, VA_CAPACITY)) {
    int early = total - (int)ceil((double)VES(o, v, VA_CAPACITY) / (double)vol);
    VES(o, v, VA_EMPTY) -= early; vessel_update_remaining(o, v);
    PORT(o, port, PA_EMPTY) += early;
    VES(o, v, VA_EARLY_DISCHARGE) = early;
  }
}
static void on_arrival(cim_oracle* o, event* e) { /* :600-632 */
  int v = e->b;
  VES(o, v, VA_LAST_LOC_IDX) = VES(o, v, VA_NEXT_LOC_IDX);
  VES(o, v, VA_IS_PARKING) = 1;
  int next = VES(o, v, VA_NEXT_LOC_IDX);
  VES(o, v, VA_LOC_PORT_IDX) = o->stops[v][next].port_idx;
  set_future_stops(o, v, next, next);
  apply_planned_stops(o, v, next);
}
static void on_departure(cim_oracle* o, event* e) { /* :634-656 */
  int v = e->b;
  VES(o, v, VA_NEXT_LOC_IDX) += 1;
  VES(o, v, VA_IS_PARKING) = 0;
  VES(o, v, VA_LOC_PORT_IDX) = -1;
  set_past_stops(o, v, VES(o, v, VA_LAST_LOC_IDX), VES(o, v, VA_NEXT_LOC_IDX));
}
static void on_discharge(cim_oracle* o, event* e) { /* :658-693 */
  int v = e->a, port = e->c, qty = e->d;
  VES(o, v, VA_FULL) -= qty; vessel_update_remaining(o, v);
  PORT(o, port, PA_ON_CONSIGNEE) += qty;
  FOV(o, v, port) -= qty;
  int bt = buffer_ticks(o, o->t.empty_return_base[port], o->t.empty_return_noise[port]);
  event* r = ev_new(e->tick + bt, EV_RETURN_EMPTY, port, qty, 0, 0);
  if (bt == 0) ev_add_immediate(e, r, 0); else eb_insert(o, r);
}
static void on_empty_return(cim_oracle* o, event* e) { /* :695-706 */
  PORT(o, e->a, PA_ON_CONSIGNEE) -= e->b;
  PORT(o, e->a, PA_EMPTY) += e->b;
}
static float i2f(int32_t x) { float f; memcpy(&f, &x, 4); return f; }
static int32_t f2i(float f) { int32_t x; memcpy(&x, &f, 4); return x; }
static void on_action_received(cim_oracle* o, event* e) { /* :708-748 */
  for (int i = 0; i < e->n_actions; i++) {
    const int32_t* a = e->actions + 4 * i;
    int v = a[0], port = a[1], move = a[2], type = a[3];
    if (v < 0 || v >= o->V || port < 0 || port >= o->P || move < 0) { o->error |= 1; continue; }
    int port_empty = PORT(o, port, PA_EMPTY), vessel_empty = VES(o, v, VA_EMPTY);
    if (type == MRX_ACTION_DISCHARGE) {
      if (!(move <= vessel_empty)) { o->error |= 1; continue; } /* assert :731 */
      PORT(o, port, PA_EMPTY) = port_empty + move;
      VES(o, v, VA_EMPTY) = vessel_empty - move; vessel_update_remaining(o, v);
    } else {
      int rs = VES(o, v, VA_REMAINING_SPACE);
      if (!(move <= (port_empty < rs ? port_empty : rs))) { o->error |= 1; continue; } /* assert :736 */
      PORT(o, port, PA_EMPTY) = port_empty - move;
      VES(o, v, VA_EMPTY) = vessel_empty + move; vessel_update_remaining(o, v);
    }
    o->total_operate_num += move;
    /* float32 attribute: python float (double) add, stored back as f32 */
    PORT(o, port, PA_TRANSFER_COST) = f2i((float)((double)i2f(PORT(o, port, PA_TRANSFER_COST)) + (double)move));
    PLAN(o, v, port) += o->vessel_period[v];
  }
}

static void dispatch(cim_oracle* o, event* e) {
  switch (e->type) {
    case EV_DEPARTURE: on_departure(o, e); break;
    case EV_RETURN_FULL: on_full_return(o, e); break;
    case EV_RETURN_EMPTY: on_empty_return(o, e); break;
    case EV_ORDER: on_order_generated(o, e); break;
    case EV_ARRIVAL: on_arrival(o, e); break;
    case EV_LOAD_FULL: on_full_load(o, e); break;
    case EV_DISCHARGE_FULL: on_discharge(o, e); break;
    case EV_TAKE_ACTION: on_action_received(o, e); break;
    default: break; /* PENDING_DECISION has no handler */
  }
}

/* event_buffer.py:190-247 execute + event_linked_list.py:86-137.  Returns the first pending
 * decision event (Sequential mode only ever consumes decision_payloads[0], core.py:350-353). */
static event* eb_execute(cim_oracle* o, int tick) {
  if (tick < 0 || tick >= o->n_lists) return NULL;
  ev_list* l = &o->lists[tick];
  for (;;) {
    /* _clear_finished_events */
    while (l->head && l->head->state == ST_FINISHED) {
      event* e = l->head;
      l->head = e->next; if (!l->head) l->tail = NULL;
      if (e->imm_count) { /* _extract_sub_events: splice immediate list to the head */
        e->imm_tail->next = l->head;
        if (!l->head) l->tail = e->imm_tail;
        l->head = e->imm_head;
        e->imm_head = e->imm_tail = NULL; e->imm_count = 0;
      }
      e->next = NULL; ev_free_chain(e);
    }
    event* f = l->head;
    if (!f) return NULL;
    if (f->type == EV_PENDING_DECISION && f->state != ST_EXECUTING) return f;
    f->state = ST_EXECUTING;
    dispatch(o, f);
    f->state = ST_FINISHED;
  }
}

/* business_engine.py:201-224 */
static int post_step(cim_oracle* o, int tick) {
  if ((tick + 1) % o->resolution == 0) {
    for (int p = 0; p < o->P; p++) PORT(o, p, PA_ACC_FULFILLMENT) = PORT(o, p, PA_ACC_BOOKING) - PORT(o, p, PA_ACC_SHORTAGE);
    take_snapshot(o, frame_index(o, tick));
    for (int p = 0; p < o->P; p++) {
      PORT(o, p, PA_SHORTAGE) = 0; PORT(o, p, PA_BOOKING) = 0; PORT(o, p, PA_FULFILLMENT) = 0;
      PORT(o, p, PA_TRANSFER_COST) = f2i(0.0f);
    }
  }
  return tick + 1 == o->max_tick;
}

static void get_metrics(const cim_oracle* o, int64_t m[3]) { /* :270-282 */
  int64_t b = 0, s = 0;
  for (int p = 0; p < o->P; p++) { s += PORT(o, p, PA_ACC_SHORTAGE); b += PORT(o, p, PA_ACC_BOOKING); }
  m[0] = b; m[1] = s; m[2] = o->total_operate_num;
}

/* ------------------------------------------------------------------ public API */
cim_oracle* cim_oracle_create(const mrx_cim_topology* topo, int start_tick, int durations, int resolution, int max_snapshots) {
  cim_oracle* o = (cim_oracle*)calloc(1, sizeof(cim_oracle));
  topo_copy(o, topo);
  o->P = topo->n_ports; o->V = topo->n_vessels; o->past_n = topo->past_stop_number; o->future_n = topo->future_stop_number;
  o->vw = 10 + 2 * o->past_n + 2 * o->future_n;
  o->start_tick = start_tick; o->durations = durations; o->max_tick = start_tick + durations; o->resolution = resolution;
  o->off_ports = 0; o->off_vessels = o->P * PA_COUNT; o->off_fop = o->off_vessels + o->V * o->vw;
  o->off_fov = o->off_fop + o->P * o->P; o->off_plans = o->off_fov + o->V * o->P;
  o->fw = o->off_plans + o->V * o->P;
  o->S = max_snapshots > 0 ? max_snapshots : (int)ceil((double)(o->max_tick - start_tick) / (double)resolution); /* abs_business_engine.py:115-129 */
  o->snaps = (int32_t*)calloc((size_t)(o->S + 1) * o->fw, sizeof(int32_t));
  o->frame = o->snaps; /* row 0 is the live frame, like np_backend */
  o->slot_fi = (int*)malloc(sizeof(int) * (size_t)(o->S + 1));
  o->fi_order = (int*)malloc(sizeof(int) * (size_t)(o->S + 2));
  o->stops = (stop_t**)calloc((size_t)o->V, sizeof(stop_t*)); o->n_stops = (int*)calloc((size_t)o->V, sizeof(int)); o->stops_cap = (int*)calloc((size_t)o->V, sizeof(int));
  o->vessel_period = (int*)calloc((size_t)o->V, sizeof(int));
  o->order_proportion = (int32_t*)calloc((size_t)o->max_tick + 1, sizeof(int32_t));
  o->n_lists = o->max_tick; o->lists = (ev_list*)calloc((size_t)o->n_lists + 1, sizeof(ev_list));
  simrand_clear(&o->rnd); /* fresh process */
  gen_cim_data(o, topo->seed); /* CimDataContainerWrapper.__init__ -> _init_data_container(None) */
  snapshots_reset(o);
  reset_nodes(o);          /* _init_frame/_init_nodes */
  load_departure_events(o);
  init_vessel_plans(o);
  o->tick = start_tick;
  return o;
}
void cim_oracle_destroy(cim_oracle* o) {
  if (!o) return;
  eb_reset(o);
  for (int v = 0; v < o->V; v++) free(o->stops[v]);
  free(o->stops); free(o->n_stops); free(o->stops_cap); free(o->vessel_period); free(o->order_proportion);
  free(o->lists); free(o->snaps); free(o->slot_fi); free(o->fi_order);
  free(o);
}
/* core.py:219-229 -> cim_data_container_helpers.py:68-70 */
void cim_oracle_set_seed(cim_oracle* o, int64_t seed) { o->wrapper_random_seed = seed; o->wrapper_has_seed = 1; o->re_init_flag = 1; }

/* core.py:143-170 + business_engine.py:226-242 + cim_data_container_helpers.py:56-66 */
void cim_oracle_reset(cim_oracle* o, int keep_seed) {
  o->error &= ~1; /* the invalid-action flag describes ONE episode (the reference raises on the spot; the engine's per-env status word is
                     rewritten by reset): without this a two-episode comparison differs on the flag alone — found by GPU fuzz seed 9099 */
  o->tick = o->start_tick; o->waiting_action = 0; o->finished = 0; o->stop_iteration = 0; o->pending0 = NULL;
  eb_reset(o);
  snapshots_reset(o);
  reset_nodes(o);
  if (!keep_seed) {
    o->wrapper_random_seed = (int64_t)mt_randbelow(simrand_get(&o->rnd, K_ROUTE_INIT), 4096); /* randint(0, 4095) */
    o->wrapper_has_seed = 1; o->re_init_flag = 1;
  }
  if (o->re_init_flag) { gen_cim_data(o, o->wrapper_has_seed ? o->wrapper_random_seed : o->t.seed); o->re_init_flag = 0; }
  else o->is_need_reset_seed = 1; /* cim_data_container.py:246-248 */
  load_departure_events(o);
  init_vessel_plans(o);
  o->total_operate_num = 0;
}

/* Env.step, Sequential mode (core.py:92-133 driving the generator :317-381).
 * decision[8] = (tick, port, vessel, scope.load, scope.discharge, early_discharge, frame_index, valid)
 * returns is_done (1/0); after the final (metrics,None,True) further calls return done with valid=-1
 * (reference: (None, None, True), core.py:128-133). */
int cim_oracle_step(cim_oracle* o, const int32_t* actions, int n_actions, int32_t decision[8], int64_t metrics[3]) {
  memset(decision, 0, sizeof(int32_t) * 8);
  if (o->finished) { decision[7] = -1; metrics[0] = metrics[1] = metrics[2] = 0; return 1; }
  if (o->waiting_action) {
    /* _assign_action core.py:301-315 */
    event* de = o->pending0;
    de->state = ST_EXECUTING;
    event* ae = ev_new(o->tick, EV_TAKE_ACTION, 0, 0, 0, 0);
    ae->n_actions = n_actions;
    ae->actions = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)(n_actions > 0 ? n_actions : 1));
    if (n_actions > 0) memcpy(ae->actions, actions, sizeof(int32_t) * 4 * (size_t)n_actions);
    ev_add_immediate(de, ae, 1);
    o->waiting_action = 0;
  } else {
    be_step(o, o->tick);
  }
  for (;;) {
    event* pend = eb_execute(o, o->tick);
    if (pend) {
      int fi = frame_index(o, o->tick);
      take_snapshot(o, fi); /* core.py:345 */
      int port = pend->a, v = pend->b;
      int pe = PORT(o, port, PA_EMPTY), rs = VES(o, v, VA_REMAINING_SPACE);
      decision[0] = o->tick; decision[1] = port; decision[2] = v;
      decision[3] = pe < rs ? pe : rs; decision[4] = VES(o, v, VA_EMPTY); /* action_scope :247-260 */
      decision[5] = VES(o, v, VA_EARLY_DISCHARGE); decision[6] = fi; decision[7] = 1;
      get_metrics(o, metrics);
      o->pending0 = pend; o->waiting_action = 1;
      return 0;
    }
    if (post_step(o, o->tick)) break;
    o->tick += 1;
    be_step(o, o->tick);
  }
  if ((o->tick + 1) % o->resolution != 0) take_snapshot(o, frame_index(o, o->tick)); /* core.py:376-378 */
  get_metrics(o, metrics);
  decision[0] = o->tick; decision[6] = frame_index(o, o->tick); decision[7] = 0;
  o->finished = 1;
  return 1;
}

/* Env.step in DecisionMode.Joint (mode 1) / JointWithSequentialAction (mode 2), core.py:354-366: every pending
 * decision event of the tick is reported (EventLinkedList._collect_pending_decision_events, event_linked_list.py:108-115);
 * the first `n_answered` get their actions, the others are FINISHED (mode 1) or stay pending (mode 2).  The flat
 * `actions` list is attached to the first answered event: pending-decision events have no handlers, so running all
 * actions when the first event is popped is the order the reference executes them in.
 * decisions: [V][8] rows (valid flag in column 7).  Returns 1 when the episode is over. */
int cim_oracle_step_joint(cim_oracle* o, int mode, const int32_t* actions, int n_actions, int n_answered, int32_t* decisions,
                          int64_t metrics[3]) {
  memset(decisions, 0, sizeof(int32_t) * 8 * (size_t)o->V);
  if (o->finished) { decisions[7] = -1; metrics[0] = metrics[1] = metrics[2] = 0; return 1; }
  if (o->waiting_action) {
    int i = 0;
    for (event* de = o->pending0; de && de->type == EV_PENDING_DECISION; de = de->next, i++) {
      if (i < n_answered) {
        de->state = ST_EXECUTING; /* _assign_action core.py:301-315 */
        event* ae = ev_new(o->tick, EV_TAKE_ACTION, 0, 0, 0, 0);
        const int n = i == 0 ? n_actions : 0;
        ae->n_actions = n;
        ae->actions = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)(n > 0 ? n : 1));
        if (n > 0) memcpy(ae->actions, actions, sizeof(int32_t) * 4 * (size_t)n);
        ev_add_immediate(de, ae, 1);
      } else if (mode == 1) {
        de->state = ST_FINISHED; /* core.py:364-366 */
      }
    }
    o->waiting_action = 0;
  } else {
    be_step(o, o->tick);
  }
  for (;;) {
    event* pend = eb_execute(o, o->tick);
    if (pend) {
      int fi = frame_index(o, o->tick);
      take_snapshot(o, fi); /* core.py:345 */
      int r = 0;
      for (event* de = pend; de && de->type == EV_PENDING_DECISION; de = de->next, r++) {
        int port = de->a, v = de->b;
        int pe = PORT(o, port, PA_EMPTY), rs = VES(o, v, VA_REMAINING_SPACE);
        int32_t* d = decisions + 8 * r;
        d[0] = o->tick; d[1] = port; d[2] = v; d[3] = pe < rs ? pe : rs; d[4] = VES(o, v, VA_EMPTY);
        d[5] = VES(o, v, VA_EARLY_DISCHARGE); d[6] = fi; d[7] = 1;
      }
      get_metrics(o, metrics);
      o->pending0 = pend; o->waiting_action = 1;
      return 0;
    }
    if (post_step(o, o->tick)) break;
    o->tick += 1;
    be_step(o, o->tick);
  }
  if ((o->tick + 1) % o->resolution != 0) take_snapshot(o, frame_index(o, o->tick));
  get_metrics(o, metrics);
  decisions[0] = o->tick; decisions[6] = frame_index(o, o->tick); decisions[7] = 0;
  o->finished = 1;
  return 1;
}

static int attr_slots(const cim_oracle* o, int node_type, int attr) {
  if (node_type == 0) return attr >= 0 && attr < PA_COUNT ? 1 : -1;
  if (node_type == 1) {
    if (attr < 0 || attr >= VA_COUNT) return -1;
    if (attr == VA_PAST_STOP_LIST || attr == VA_PAST_STOP_TICK_LIST) return o->past_n;
    if (attr == VA_FUTURE_STOP_LIST || attr == VA_FUTURE_STOP_TICK_LIST) return o->future_n;
    return 1;
  }
  if (attr == MA_FULL_ON_PORTS) return o->P * o->P;
  if (attr == MA_FULL_ON_VESSELS || attr == MA_VESSEL_PLANS) return o->V * o->P;
  return -1;
}

/* np_backend.pyx:520-549.  nt==0 -> all stored frames (insertion order); nn==0 -> all nodes.
 * Returns the number of doubles written (call with out==NULL to size). */
int64_t cim_oracle_query(cim_oracle* o, int node_type, const int32_t* ticks, int nt, const int32_t* nodes, int nn,
                         const int32_t* attrs, int na, double* out) {
  int n_nodes = node_type == 0 ? o->P : node_type == 1 ? o->V : 1;
  int nticks = nt ? nt : o->n_fi_order;
  int nnodes = nn ? nn : n_nodes;
  int64_t w = 0;
  for (int ti = 0; ti < nticks; ti++) {
    int fi = nt ? ticks[ti] : o->fi_order[ti];
    int slot = 0;
    for (int s = 1; s <= o->S; s++) if (o->slot_fi[s] == fi) slot = s;
    const int32_t* fr = o->snaps + (size_t)slot * o->fw;
    for (int ni = 0; ni < nnodes; ni++) {
      int node = nn ? nodes[ni] : ni;
      for (int ai = 0; ai < na; ai++) {
        int a = attrs[ai], ns = attr_slots(o, node_type, a);
        if (ns < 0) return -1;
        for (int s = 0; s < ns; s++, w++) {
          if (!out) continue;
          if (!slot) { out[w] = 0.0; continue; }
          int32_t raw;
          if (node_type == 0) raw = fr[o->off_ports + node * PA_COUNT + a];
          else if (node_type == 1) raw = fr[o->off_vessels + node * o->vw + ves_list_off(o, a) + s];
          else raw = fr[(a == MA_FULL_ON_PORTS ? o->off_fop : a == MA_FULL_ON_VESSELS ? o->off_fov : o->off_plans) + s];
          out[w] = (node_type == 0 && a == PA_TRANSFER_COST) ? (double)i2f(raw) : (double)raw;
        }
      }
    }
  }
  return w;
}

/* live-frame query (frame row 0) with the same flattening; used by tests for the pre-decision state */
int64_t cim_oracle_query_live(cim_oracle* o, int node_type, const int32_t* attrs, int na, double* out) {
  int saved = o->slot_fi[0]; int64_t w;
  /* temporarily expose row 0 as frame index INT32_MIN+1 */
  o->slot_fi[0] = -7; (void)saved;
  int n_nodes = node_type == 0 ? o->P : node_type == 1 ? o->V : 1; w = 0;
  for (int node = 0; node < n_nodes; node++)
    for (int ai = 0; ai < na; ai++) {
      int a = attrs[ai], ns = attr_slots(o, node_type, a);
      for (int s = 0; s < ns; s++, w++) {
        int32_t raw;
        if (node_type == 0) raw = o->frame[o->off_ports + node * PA_COUNT + a];
        else if (node_type == 1) raw = o->frame[o->off_vessels + node * o->vw + ves_list_off(o, a) + s];
        else raw = o->frame[(a == MA_FULL_ON_PORTS ? o->off_fop : a == MA_FULL_ON_VESSELS ? o->off_fov : o->off_plans) + s];
        out[w] = (node_type == 0 && a == PA_TRANSFER_COST) ? (double)i2f(raw) : (double)raw;
      }
    }
  o->slot_fi[0] = -1;
  return w;
}

/* ---- introspection used by the parity tests of the device-side data generator ---- */
int cim_oracle_tick(const cim_oracle* o) { return o->tick; }
int cim_oracle_error(const cim_oracle* o) { return o->error; }
int64_t cim_oracle_data_seed(const cim_oracle* o) { return o->data_seed; }
int cim_oracle_num_frames(const cim_oracle* o) { return o->n_fi_order; }
int cim_oracle_frame_indices(const cim_oracle* o, int32_t* out, int cap) { int n = o->n_fi_order < cap ? o->n_fi_order : cap; for (int i = 0; i < n; i++) out[i] = o->fi_order[i]; return o->n_fi_order; }
int cim_oracle_num_stops(const cim_oracle* o, int v) { return o->n_stops[v]; }
int cim_oracle_get_stops(const cim_oracle* o, int v, int32_t* arrival, int32_t* leave, int32_t* port, int cap) {
  int n = o->n_stops[v] < cap ? o->n_stops[v] : cap;
  for (int k = 0; k < n; k++) { arrival[k] = o->stops[v][k].arrival_tick; leave[k] = o->stops[v][k].leave_tick; port[k] = o->stops[v][k].port_idx; }
  return o->n_stops[v];
}
void cim_oracle_get_order_proportion(const cim_oracle* o, int32_t* out) { memcpy(out, o->order_proportion, sizeof(int32_t) * (size_t)o->max_tick); }
void cim_oracle_get_vessel_period(const cim_oracle* o, int32_t* out) { for (int v = 0; v < o->V; v++) out[v] = o->vessel_period[v]; }
void cim_oracle_stream_seeds(const cim_oracle* o, int64_t out[4]) { for (int k = 0; k < K_COUNT; k++) out[k] = o->rnd.created[k] >= 0 ? o->rnd.seed_of[k] : -1; }

/* Whole-rollout driver used by bench.py's cpu_baseline leg (keeps the timed loop in C): steps the env
 * with the counter-based random agent of oracle/cim_oracle.py::hash_policy_action until done or
 * max_steps decisions; returns the number of decisions answered, *ticks_out = ticks advanced. */
static uint64_t mix64(uint64_t seed, uint64_t step) {
  uint64_t x = seed * 0x9E3779B97F4A7C15ull + step * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}
int64_t cim_oracle_rollout(cim_oracle* o, int64_t env_seed, int64_t max_steps, int64_t* ticks_out, int64_t metrics[3]) {
  int32_t dec[8], act[4]; int64_t n = 0;
  int t0 = o->tick;
  int done = cim_oracle_step(o, NULL, 0, dec, metrics);
  while (!done && (max_steps < 0 || n < max_steps)) {
    uint64_t x = mix64((uint64_t)env_seed, (uint64_t)n), r = x >> 1;
    act[0] = dec[2]; act[1] = dec[1];
    if ((x & 1ull) == 0 && dec[3] > 0) { act[2] = (int32_t)(r % (uint64_t)(dec[3] + 1)); act[3] = MRX_ACTION_LOAD; }
    else { act[2] = (int32_t)(r % (uint64_t)(dec[4] + 1)); act[3] = MRX_ACTION_DISCHARGE; }
    done = cim_oracle_step(o, act, 1, dec, metrics);
    n++;
  }
  if (ticks_out) *ticks_out = o->tick - t0 + (done ? 1 : 0);
  return n;
}

/* bench.py's cpu_baseline leg: whole episodes (set_seed + reset + rollout with the counter-based agent) until
 * budget_s seconds have passed, entirely in C so that one thread per host core runs without touching the interpreter. */
int64_t cim_oracle_bench(cim_oracle* o, int64_t first_seed, double budget_s, int64_t* ticks_out, int64_t* episodes_out) {
  struct timespec t0, t;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  int64_t steps = 0, ticks = 0, episodes = 0, metrics[3];
  for (;;) {
    clock_gettime(CLOCK_MONOTONIC, &t);
    if ((double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec) >= budget_s) break;
    int64_t tk = 0;
    cim_oracle_set_seed(o, first_seed + episodes);
    cim_oracle_reset(o, 1);
    steps += cim_oracle_rollout(o, first_seed + episodes, -1, &tk, metrics);
    ticks += tk;
    episodes++;
  }
  if (ticks_out) *ticks_out = ticks;
  if (episodes_out) *episodes_out = episodes;
  return steps;
}

/* CPython random pinning hooks (tests/test_oracle_mt.py) */
void cim_oracle_mt_selftest(int64_t seed, int n, double* out_random, uint32_t* out_randbelow4096) {
  mt_state s; mt_seed_int(&s, seed);
  for (int i = 0; i < n; i++) out_random[i] = mt_random(&s);
  for (int i = 0; i < n; i++) out_randbelow4096[i] = mt_randbelow(&s, 4096);
}
