// cb_device.h — device code of the batched citi_bike rollout engine (one env per LANE).
//
// citi_bike's per-tick work is a short, strictly sequential chain of events over a tiny state (a few words per
// station), and every env of a batch replays the SAME trip table.  So, unlike the CIM engine (one env per wave),
// one lane owns one env: state is struct-of-arrays `X[word][env]`, the trip/return tables are wave-uniform loads,
// and whenever lanes agree on the station index (all trip-driven events) state accesses are fully coalesced.
// There are no cross-lane operations, so the same source compiles for the host (tests/emu/cb_emu.cpp).
//
// What it replaces in the reference (per env):
//   EventBuffer.execute + Env._simulate        maro/event_buffer/event_buffer.py:190-247, simulator/core.py:317-381
//   CitibikeBusinessEngine.step/post_step      simulator/scenarios/citi_bike/business_engine.py:101-147
//   _on_required_bike/_on_bike_returned/_on_rebalance_bikes/_on_bike_deliver/_on_action_received   …:398-559
//   BikeDecisionStrategy + neighbour filters   simulator/scenarios/citi_bike/decision_strategy.py:11-343
//   Station callbacks                          simulator/scenarios/citi_bike/station.py:58-75
// The event queue is gone.  A tick is a fixed sequence of phases:
//   1 events queued by earlier ticks (ReturnBike from the shared return schedule filtered by this env's
//     "trip got a bike" bits, merged by scheduling tick with this env's DeliverBike pool)
//   2 RequireBike for the tick's trips   3 RebalanceBike -> decision set; zero-duration returns
//   4 decisions, one env-step each       5 DeliverBike events with transfer time 0   6 post_step / snapshot.
// Everything in 1-3 and 6 that does not depend on the env is laid out ONCE on the host as a single stream of event records
// (CbParams::ev_rec, cb_layout.h); an env's step replays the stream from its cursor until a decision comes up.  The step is
// written for SIMT execution: lanes of a wave stand at different places of the stream, so the replay is ONE flat loop that
// handles one record per iteration whatever its kind (a wave then runs max-over-lanes(records per step) iterations; nested
// per-tick / per-phase loops ran the SUM over ticks of the per-tick maxima, ~10x more), and the expensive, rare pieces (the
// station sweep of a rebalance check, the end of a tick with its snapshot, the action scope of the decision) sit outside
// that loop at points where the wave has reconverged, so each runs once per wave, not once per divergent subset of lanes.
#pragma once
#include <math.h>
#include <stdint.h>

#include "cb_params.h"

// CD(f) / CDA(f, i): an integer dimension of the plan (CbParams::f, CbParams::f[i]).  The generic kernels read it from the
// kernel arguments; a build specialised for one plan (cb_spec.hip, compiled at engine creation) turns every one of them into
// a compile-time constant MRXC_f / MRXC_f(i): station loops unroll, the SoA strides fold, SGPR spills drop from 100 to 4.
#ifdef MRX_SPECIALIZED
#define CD(f) (MRXC_##f)
#define CDA(f, i) (MRXC_##f(i))
#else
#define CD(f) (K.f)
#define CDA(f, i) (K.f[i])
#endif

#ifdef __HIPCC__
#define MRX_DEVM_EARLY __device__ __forceinline__
#else
#define MRX_DEVM_EARLY inline
#endif
namespace cb {

// tools only (-DMRX_CB_PROFILE, e.g. MARO_AMD_SPEC_FLAGS): shader-clock cycles of a lane's step attributed to phases, summed
// per env in CbParams::prof (tools/cb_phase_profile.py)
#if defined(MRX_CB_PROFILE) && defined(__HIPCC__)
struct Prof {
  long long last;
  int acc[16];
  __device__ __forceinline__ Prof() { for (int i = 0; i < 16; i++) acc[i] = 0; last = clock64(); }
  __device__ __forceinline__ void mark(int i) { const long long c = clock64(); acc[i] += (int)(c - last); last = c; }
  __device__ __forceinline__ void flush(const CbParams& K, int e) {
    for (int i = 0; i < 16; i++) if (acc[i]) K.prof[CB_IX(CD(aos), CD(stride), 16, i, e)] += acc[i];
  }
};
// ... and of a WAVE's step (the wave kernels: one env per wave), lane 0 adding each stretch straight into the env's counters
// (tools/cb_wave_profile.py: phases 0-7 the replay kernel, 8-11 the in-tick kernel, 12 / 13 their call counts)
struct WProf {
  long long last;
  __device__ __forceinline__ WProf() { last = clock64(); }
  __device__ __forceinline__ void mark(const CbParams& K, int e, int i) {
    const long long c = clock64();
    if (threadIdx.x == 0) atomicAdd(&K.prof[CB_IX(CD(aos), CD(stride), 16, i, e)], (int)(c - last));
    last = c;
  }
  __device__ __forceinline__ void count(const CbParams& K, int e, int i) {
    if (threadIdx.x == 0) atomicAdd(&K.prof[CB_IX(CD(aos), CD(stride), 16, i, e)], 1);
  }
};
#else
struct Prof {
  MRX_DEVM_EARLY void mark(int) {}
  MRX_DEVM_EARLY void flush(const CbParams&, int) {}
};
struct WProf {
  MRX_DEVM_EARLY void mark(const CbParams&, int, int) {}
  MRX_DEVM_EARLY void count(const CbParams&, int, int) {}
};
#endif

#define GHDR(w) K.hdr[CB_IX(CD(aos), CD(stride), CH_WORDS, (w), e)] /* header word in HBM */
// the live frame in HBM (reset, query, and the generic step)
#define GST(a, s) K.live[CB_IX(CD(aos), CD(stride), CD(FW), ((size_t)(a) * CD(S) + (size_t)(s)), e)]
#ifdef __HIPCC__
#define MRX_DEVM __device__ __forceinline__
#else
#define MRX_DEVM inline /* host harness (tests/emu): MRX_DEV is `static inline`, not valid on members */
#endif

// Where an env's working state lives during a step.
//   generic build:      in HBM (struct-of-arrays, any size), every access an L2 round trip
//   specialised build:  in LDS, one column per lane — the live frame, the stations' capacities, the fulfilled-trip bit ring,
//                       the pending-decision masks, the action-scope work arrays and a block of event records: loaded once at
//                       the start of the step, stored back at its end.  A lane indexes its state with RUNTIME station numbers;
//                       LDS takes a per-lane address in one instruction (the first specialised engine kept the frame in
//                       registers behind select chains: S dependent v_cndmask per access, and a wave alone on its SIMD —
//                       the usual batch gives the chip one wave per CU — pays every dependent instruction in full).
//                       Layout: the lanes' event blocks first (lane-major), then word w of lane l at
//                       lds[((32 + w) << K.lsh) + l], K.lsh = log2(envs per wave); the engine
//                       lowers envs-per-wave until lds_words x 4 B x envs-per-wave fits (mrx_cb_set_lanes_per_wave).
#if defined(MRX_SPECIALIZED) && (MRXC_lds_words * 4 <= MRX_CB_LDS_BYTES)
#define MRX_CB_LDSFRAME 1
#ifdef __HIPCC__
// Read-modify-writes of the replay loop as fire-and-forget LDS operations (ds_add / ds_min / ds_or / ds_and without return): a
// `+=` on an LDS word is a read, a wait for it, and a write — and since the station index is a runtime value the compiler keeps
// every such triple in program order, i.e. one full LDS round trip per counter with the wave alone on its SIMD.  The LDS executes
// one wave's operations in order, so later reads of the same word see them.
#define CB_ADD(lv, v) ((void)__hip_atomic_fetch_add(&(lv), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#define CB_MIN(lv, v) ((void)__hip_atomic_fetch_min(&(lv), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#define CB_OR(lv, v) ((void)__hip_atomic_fetch_or(&(lv), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#define CB_AND(lv, v) ((void)__hip_atomic_fetch_and(&(lv), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#define CB_LAND(x) asm volatile("" : "+v"(x)) /* the value is needed HERE: keeps a group of independent reads in one round trip (the compiler would sink each into the branch that uses it) */
extern __shared__ int32_t mrx_cb_lds[];
#define LEV() (mrx_cb_lds + threadIdx.x * (CB_EV_BLOCK * 4)) /* the lane's event block: lane-major, so a record is one 16-byte read */
// envs per wave as a shift: a plan constant where the plan has one (CbParams::lsh_plan: the column addresses then fold into the
// LDS instructions' offset fields — the per-access shift + add was a sixth of a replay iteration's instructions), else the launch's
#if MRXC_lsh_plan >= 0
#define CB_LSH MRXC_lsh_plan
#else
#define CB_LSH K.lsh
#endif
#define LF(w) mrx_cb_lds[((CB_EV_BLOCK * 4 + (w)) << CB_LSH) + threadIdx.x]
#define LF0(w) mrx_cb_lds[CB_EV_BLOCK * 4 + (w)] /* word w of the wave's ONE column, from any lane (the wave replay kernel: K.lsh = 0) */
#else
static int32_t mrx_cb_lds_host[MRXC_lds_words]; /* host harness: one env at a time */
#define LEV() mrx_cb_lds_host
#define LF(w) mrx_cb_lds_host[CB_EV_BLOCK * 4 + (w)]
#define LF0(w) LF(w)
#endif
#define LDS_CAP (MRXC_FW)
#define LDS_FUL (LDS_CAP + MRXC_S)
#define LDS_DMK (LDS_FUL + MRXC_w_words)
#define LDS_SCR (LDS_DMK + 2 * MRXC_mask_words)
// The action scope's three work arrays: in the LDS column for the lane kernel's plans; env-major plans (the wave-cooperative
// kernels, which rank the candidates in registers) keep them in HBM — only the scalar fallback of a station with more candidates
// than the wave path takes uses them there — so that a wave's column is 3 S words (9.6 KB at 800 stations) smaller and more envs
// are resident per CU.
#if MRXC_aos
#define LDS_SCR_WORDS 0
#else
#define LDS_SCR_WORDS (3 * MRXC_S)
#endif
#define LDS_HDR (LDS_SCR + LDS_SCR_WORDS)
#define LDS_TWC (LDS_HDR + CH_WORDS)
#if MRXC_ring_slots <= CB_TWC_LDS
#define MRX_CB_TWC_LDS 1 /* the trip-window filter's (frame, tick) tag of every ring slot: read at every decision */
#define LDS_TWC_WORDS (2 * MRXC_ring_slots)
#define TWCF(slot) LF(LDS_TWC + (slot))
#define TWCT(slot) LF(LDS_TWC + MRXC_ring_slots + (slot))
#else
#define LDS_TWC_WORDS 0
#endif
// The delivery pool's hot end (env-major plans: the wave replay kernel, K.pool_stage): the landing-tick bucket table, and a copy of
// the CB_POOL_STAGE pool entries from the ring's head on — a decision tick at the reference's size leaves ~200 deliveries in
// flight, and walking them in HBM cost four dependent round trips per delivery (two thirds of a replay call's cycles, measured).
#if MRXC_aos
#define MRX_CB_POOL_LDS 1
#define LDS_BKT (LDS_TWC + LDS_TWC_WORDS)
#define LDS_PSA (LDS_BKT + CB_BKT_WORDS)   /* the window's anchor: pool index of ring position 0 of the copy */
#define LDS_PST (LDS_PSA + 1)
#define LDS_EVM (LDS_PST + CB_POOL_STAGE * CB_POOL_WORDS)   /* the event window's bounds: first stream index held, first not held */
#define LDS_EVW ((LDS_EVM + 2 + 3) & ~3)                    /* its records (16-byte aligned: the column starts on a 16-byte boundary) */
#define LDS_POOL_WORDS CB_POOL_STAGE_WORDS
#else
#define LDS_POOL_WORDS 0
#endif
static_assert(LDS_TWC + LDS_TWC_WORDS + LDS_POOL_WORDS + CB_EV_BLOCK * 4 == MRXC_lds_words, "cb_layout.h and cb_device.h disagree on the LDS column");
// the env header too: as a register array that rare branches (the delivery pool) modify, every join of the replay loop copied
// all 16 words back and forth
#define HDR(w) LF(LDS_HDR + (w))
#define LIVE(w) LF(w)
#define ST(a, s) LF((a) * MRXC_S + (s))
#define CAP(s) LF(LDS_CAP + (s))
#define FUL(i) (*(uint32_t*)&LF(LDS_FUL + (i)))
#define DMK(i) (*(uint32_t*)&LF(LDS_DMK + (i)))
#if MRXC_aos
#define SCR(i) K.scratch[CB_IX(CD(aos), CD(stride), (3 * CD(S)), (i), e)]
#else
#define SCR(i) LF(LDS_SCR + (i))
#endif
#else
#define HDR(w) hd[(w)] /* header word of the env being stepped: a register copy (step_env) */
#define LIVE(w) K.live[CB_IX(CD(aos), CD(stride), CD(FW), (w), e)]
#define ST(a, s) GST(a, s)
#define CAP(s) K.capacity[s]
#define FUL(i) GFUL(i)
#define DMK(i) GDMK(i)
#define SCR(i) K.scratch[CB_IX(CD(aos), CD(stride), (3 * CD(S)), (i), e)]
#endif
#ifndef MRX_CB_TWC_LDS
#define TWCF(slot) K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), (slot), e)]
#define TWCT(slot) K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), (slot), e)]
#endif

#ifndef CB_ADD  /* state in HBM, or the host build: plain read-modify-writes */
#define CB_ADD(lv, v) ((lv) += (v))
#define CB_MIN(lv, v) ((lv) = (v) < (lv) ? (v) : (lv))
#define CB_OR(lv, v) ((lv) |= (v))
#define CB_AND(lv, v) ((lv) &= (v))
#define CB_LAND(x) ((void)0)
#endif

// The env's place in the shared event stream (4 words per record).  LDS build: records are consumed out of a block of
// CB_EV_BLOCK records in the lane's LDS block (one 16-byte LDS read per record), refilled with CB_EV_BLOCK independent loads:
// an L2 round trip per block, not per record.  The stream is padded by 2 x CB_EV_BLOCK records so a refill may run past the
// last record.
struct EvWin {
  int pos;  // stream index of the next record
#ifdef MRX_CB_LDSFRAME
  int base;  // stream index of the first record of the LDS block
  struct alignas(16) Rec { int32_t w0, a, b, c; };
  // CB_EV_BLOCK records from stream index idx into the lane's LDS block: the loads are independent (one L2 round trip per
  // block).  (Holding the NEXT block in registers while the current one is consumed was measured first: the registers are
  // loop-carried, so the compiler copies all 32 of them — and waits for the loads — at the head of every iteration.)
  MRX_DEVM void refill(const CbParams& K, int idx) {
    Rec r[CB_EV_BLOCK];
#pragma unroll
    for (int k = 0; k < CB_EV_BLOCK; k++) r[k] = *(const Rec*)(K.ev_rec + (size_t)(idx + k) * 4);
    Rec* blk = (Rec*)LEV();
#pragma unroll
    for (int k = 0; k < CB_EV_BLOCK; k++) blk[k] = r[k];
  }
  MRX_DEVM void open(const CbParams& K, int p) {
    pos = base = p;
    refill(K, p);
  }
  MRX_DEVM Rec rec(const CbParams& K) const { return ((const Rec*)LEV())[pos - base]; }
  MRX_DEVM void advance(const CbParams& K) {
    pos++;
    if (pos - base == CB_EV_BLOCK) {
      base += CB_EV_BLOCK;
      refill(K, base);
    }
  }
  MRX_DEVM void close(const CbParams&) {}
#else
  struct alignas(16) Rec { int32_t w0, a, b, c; };
  MRX_DEVM void open(const CbParams&, int p) { pos = p; }
  MRX_DEVM Rec rec(const CbParams& K) const { return *(const Rec*)(K.ev_rec + (size_t)pos * 4); }
  MRX_DEVM void advance(const CbParams&) { pos++; }
  MRX_DEVM void close(const CbParams&) {}
#endif
};

#ifdef MRX_CB_POOL_LDS
// The wave replay kernel's window (K.pool_stage): CB_EVW_RECS records fetched by the whole wave, one per lane, together with the
// env's state (evw_fetch / evw_put) — a step budget's worth of records without a round trip inside the sequential part.  Lane 0 consumes
// them; past the window it refills CB_EV_BLOCK at a time.  The bounds live in LDS so that the runs of one call share the window.
MRX_DEV EvWin::Rec evw_fetch(const CbParams& K, int pos) { return *(const EvWin::Rec*)(K.ev_rec + (size_t)(pos + wave::lane()) * 4); }
MRX_DEV void evw_put(const CbParams& K, int pos, const EvWin::Rec& r) {  // (fetch early, put late: the load joins the state's round trip)
  const int lane = wave::lane();
  ((EvWin::Rec*)&LF0(LDS_EVW))[lane] = r;
  if (lane == 0) { LF0(LDS_EVM) = pos; LF0(LDS_EVM + 1) = pos + CB_EVW_RECS; }
}
struct EvWinW {
  int pos, base, lim;
  typedef EvWin::Rec Rec;
  MRX_DEVM void refill(const CbParams& K, int idx) {
    Rec r[CB_EV_BLOCK];
#pragma unroll
    for (int k = 0; k < CB_EV_BLOCK; k++) r[k] = *(const Rec*)(K.ev_rec + (size_t)(idx + k) * 4);
    Rec* blk = (Rec*)&LF(LDS_EVW);
#pragma unroll
    for (int k = 0; k < CB_EV_BLOCK; k++) blk[k] = r[k];
    base = idx;
    lim = idx + CB_EV_BLOCK;
  }
  MRX_DEVM void open(const CbParams& K, int p) {
    pos = p;
    base = LF(LDS_EVM);
    lim = LF(LDS_EVM + 1);
    if (p < base || p >= lim) refill(K, p);
  }
  MRX_DEVM Rec rec(const CbParams& K) const { return ((const Rec*)&LF(LDS_EVW))[pos - base]; }
  MRX_DEVM void advance(const CbParams& K) {
    pos++;
    if (pos == lim) refill(K, pos);
  }
  MRX_DEVM void close(const CbParams& K) { LF(LDS_EVM) = base; LF(LDS_EVM + 1) = lim; }
};
#endif

#define RINGW(slot, w) K.ring[CB_IX(CD(aos), CD(stride), ((size_t)CD(ring_slots) * (CD(FW) + 1)), ((size_t)(slot) * (CD(FW) + 1) + (size_t)(w)), e)] /* word w of ring slot `slot` (w = FW: the tick it was taken at) */
#define GFUL(i) K.fulfilled[CB_IX(CD(aos), CD(stride), CD(w_words), (i), e)]
#define GDMK(i) K.decmask[CB_IX(CD(aos), CD(stride), (2 * CD(mask_words)), (i), e)]
#define BKT(i) K.bkt[CB_IX(CD(aos), CD(stride), (2 * CB_LAND_SLOTS + CB_LAND_SLOTS / 32), (i), e)] /* landing-tick buckets of the delivery pool */
#define POOL(i, w) K.pool[CB_IX(CD(aos), CD(stride), (CD(pool_cap) * CB_POOL_WORDS), ((size_t)(i) * CB_POOL_WORDS + (w)), e)]

// Delivery pool / bucket table accessors.  Everywhere but the wave replay kernel: the words in HBM.  There (K.pool_stage): reads
// come out of the LDS copy — every bucket word; a pool entry when it lies within K.pool_stage (<= CB_POOL_STAGE) ring positions of the anchor —
// and writes go to both, so HBM is current at every moment and nothing is written back.
#ifdef MRX_CB_POOL_LDS
MRX_DEV int pool_pos(const CbParams& K, int idx) {  // ring position of pool index idx relative to the anchor
  const int a = LF(LDS_PSA);
  return idx - a + (idx < a ? CD(pool_cap) : 0);
}
MRX_DEV int pool_rd(const CbParams& K, int e, int idx, int w) {
  if (K.pool_stage) {
    const int p = pool_pos(K, idx);
    if (p < K.pool_stage) return LF(LDS_PST + p * CB_POOL_WORDS + w);
  }
  return POOL(idx, w);
}
// Entry idx if it lands at tick t and was scheduled before sched_lt (false: out[] is not complete).  From HBM the two words
// of the test come first — the walk ends on an entry that fails it; from the LDS copy the whole entry is one round trip.
MRX_DEV bool pool_rd_entry(const CbParams& K, int e, int idx, int t, int sched_lt, int* out) {
  if (K.pool_stage) {
    const int p = pool_pos(K, idx);
    if (p < K.pool_stage) {
#pragma unroll
      for (int w = 0; w < CB_POOL_WORDS; w++) out[w] = LF(LDS_PST + p * CB_POOL_WORDS + w);
      return out[0] == t && out[1] < sched_lt;
    }
  }
  out[0] = POOL(idx, 0); out[1] = POOL(idx, 1);
  if (!(out[0] == t && out[1] < sched_lt)) return false;
#pragma unroll
  for (int w = 2; w < CB_POOL_WORDS; w++) out[w] = POOL(idx, w);
  return true;
}
MRX_DEV void pool_wr(const CbParams& K, int e, int idx, int w, int v) {
  POOL(idx, w) = v;
  if (K.pool_stage) {
    const int p = pool_pos(K, idx);
    if (p < K.pool_stage) LF(LDS_PST + p * CB_POOL_WORDS + w) = v;
  }
}
MRX_DEV int bkt_rd(const CbParams& K, int e, int i) { return K.pool_stage ? LF(LDS_BKT + i) : BKT(i); }
MRX_DEV void bkt_wr(const CbParams& K, int e, int i, int v) {
  BKT(i) = v;
  if (K.pool_stage) LF(LDS_BKT + i) = v;
}
// Fills the copy (all lanes of the env's wave; head / tail: the env's CH_POOL_HEAD / CH_POOL_TAIL): the bucket table, and the entries
// between the ring's head and tail that fit.  Entries pushed later land in the copy through pool_wr.  In two halves — fetch issues
// the loads, put writes LDS — so that the caller can have them in flight together with the rest of the env's state.
// Every load in flight before the first LDS write: no load sits behind a branch (positions past the table / past the ring's tail
// read a valid word again and are not used), a loop with a run-time trip count would wait for each trip's loads in turn.
struct PoolStage {
  enum { NB = (CB_BKT_WORDS + 63) / 64, NP = CB_POOL_STAGE * CB_POOL_WORDS / 64 };
  int vb[NB], vp[NP], a;
};
MRX_DEV void pool_stage_fetch(const CbParams& K, int e, int head, int tail, PoolStage& S) {
  const int lane = wave::lane();
  S.a = head % CD(pool_cap);
  int n = tail - head;
  if (n > K.pool_stage) n = K.pool_stage;
#pragma unroll
  for (int k = 0; k < PoolStage::NB; k++) {
    const int i = k * 64 + lane;
    S.vb[k] = BKT(i < CB_BKT_WORDS ? i : 0);
  }
#pragma unroll
  for (int k = 0; k < PoolStage::NP; k++) {
    const int i = k * 64 + lane;
    const int p = i / CB_POOL_WORDS, w = i - p * CB_POOL_WORDS;
    int idx = S.a + (p < n ? p : 0);
    if (idx >= CD(pool_cap)) idx -= CD(pool_cap);
    S.vp[k] = POOL(idx, w);
  }
}
MRX_DEV void pool_stage_put(const CbParams& K, const PoolStage& S) {
  const int lane = wave::lane();
#pragma unroll
  for (int k = 0; k < PoolStage::NB; k++) {
    const int i = k * 64 + lane;
    if (i < CB_BKT_WORDS) LF0(LDS_BKT + i) = S.vb[k];
  }
#pragma unroll
  for (int k = 0; k < PoolStage::NP; k++) LF0(LDS_PST + k * 64 + lane) = S.vp[k];
  if (lane == 0) LF0(LDS_PSA) = S.a;
}
#else
MRX_DEV int pool_rd(const CbParams& K, int e, int idx, int w) { return POOL(idx, w); }
MRX_DEV bool pool_rd_entry(const CbParams& K, int e, int idx, int t, int sched_lt, int* out) {  // (see the LDS-copy form above)
  out[0] = POOL(idx, 0); out[1] = POOL(idx, 1);
  if (!(out[0] == t && out[1] < sched_lt)) return false;
#pragma unroll
  for (int w = 2; w < CB_POOL_WORDS; w++) out[w] = POOL(idx, w);
  return true;
}
MRX_DEV void pool_wr(const CbParams& K, int e, int idx, int w, int v) { POOL(idx, w) = v; }
MRX_DEV int bkt_rd(const CbParams& K, int e, int i) { return BKT(i); }
MRX_DEV void bkt_wr(const CbParams& K, int e, int i, int v) { BKT(i) = v; }
#endif

MRX_DEV void set_bikes(const CbParams& K, int e, int32_t* hd, int s, int v) {  // station.py:71-75
  ST(LV_BIKES, s) = v;
  if (v < ST(LV_MIN_BIKES, s)) ST(LV_MIN_BIKES, s) = v;
}

#if defined(MRX_SPECIALIZED) && MRXC_aos
// ---- env-major plans: these run on ONE lane of the env's wave (the wave replay kernel), where every dependent read is paid in
// full — so each reads everything it may need up front, in one round trip.  (Same arithmetic, same order of writes as the forms
// below, which the one-env-per-lane kernels keep: there 64 lanes share the memory pipes and speculative reads cost throughput —
// measured on toy.3s_4t, 32768 envs: 413 -> 381 M env-steps/s with these forms.)
// decision_strategy.py:295-343 — bikes that do not fit go to the neighbours of `cur`, nearest first
enum { CB_NB_CHUNK = MRXC_nb_stride >= 8 ? 8 : MRXC_nb_stride > 1 ? MRXC_nb_stride : 1 };
MRX_DEV void move_to_neighbor(const CbParams& K, int e, int32_t* hd, int src, int cur, int number) {
  // A chunk of (up to eight) neighbours per trip: their numbers in one round trip (independent loads; the row is nb_stride long,
  // -1 past the count; the first chunk does not wait for the count), then their bikes / capacity / min_bikes in one (distinct
  // stations, and nothing below writes another neighbour's), then the arithmetic.  One load after the other made every
  // overflowing delivery a chain of a dozen dependent round trips — and a decision tick at the reference's size lands ~200
  // deliveries per env.
  const int32_t* row = K.nb + (size_t)cur * CD(nb_stride);
  int nbv[CB_NB_CHUNK];
#pragma unroll
  for (int k = 0; k < CB_NB_CHUNK; k++) nbv[k] = row[k < CD(nb_stride) ? k : 0];
  const int cnt = K.nb_cnt[cur];
  for (int i0 = 0; i0 < cnt && number > 0; i0 += CB_NB_CHUNK) {
    if (i0) {
#pragma unroll
      for (int k = 0; k < CB_NB_CHUNK; k++) nbv[k] = row[i0 + k < CD(nb_stride) ? i0 + k : i0];
    }
    int bv[CB_NB_CHUNK], cv[CB_NB_CHUNK], mv[CB_NB_CHUNK];
#pragma unroll
    for (int k = 0; k < CB_NB_CHUNK; k++) {
      const int nb = i0 + k < cnt ? nbv[k] : nbv[0];
      bv[k] = ST(LV_BIKES, nb); cv[k] = CAP(nb); mv[k] = ST(LV_MIN_BIKES, nb);
    }
#pragma unroll
    for (int k = 0; k < CB_NB_CHUNK; k++) {
      const int i = i0 + k;
      if (i < cnt && number > 0) {
        const int nb = nbv[k];
        int accept = cv[k] - bv[k];
        if (accept > number) accept = number;
        const int v = bv[k] + accept;  // set_bikes (station.py:71-75)
        ST(LV_BIKES, nb) = v;
        if (v < mv[k]) ST(LV_MIN_BIKES, nb) = v;
        const int target = CD(extra_cost_mode) == 0 ? src : CD(extra_cost_mode) == 1 ? cur : nb;
        CB_ADD(ST(LV_EXTRA_COST, target), accept * (i + 1));
        number -= accept;
      }
    }
  }
}

// _on_bike_returned :439-466 (deliver = false) and _on_bike_deliver :494-519 (deliver = true)
MRX_DEV void land_bikes(const CbParams& K, int e, int32_t* hd, bool deliver, int frm, int to, int n) {
  const int b = ST(LV_BIKES, to), cap = CAP(to), mn = ST(LV_MIN_BIKES, to);  // (everything the landing reads, in one round trip)
  int accept = cap - b;
  if (accept > n) accept = n;
  if (accept < n) {
    if (!deliver) CB_ADD(ST(LV_FAILED_RETURN, to), n - accept);
    move_to_neighbor(K, e, hd, frm, to, n - accept);   // (`to` is not among its own neighbours)
  }
  if (deliver && accept > 0) {
    CB_ADD(ST(LV_TRANSFER_COST, to), accept);
    CB_ADD(HDR(CH_OPER), accept);
  }
  ST(LV_BIKES, to) = b + accept;  // set_bikes
  if (b + accept < mn) ST(LV_MIN_BIKES, to) = b + accept;
}

#else
// decision_strategy.py:295-343 — bikes that do not fit go to the neighbours of `cur`, nearest first
MRX_DEV void move_to_neighbor(const CbParams& K, int e, int32_t* hd, int src, int cur, int number) {
  const int cnt = K.nb_cnt[cur];
  for (int i = 0; i < cnt && number > 0; i++) {
    const int nb = K.nb[(size_t)cur * CD(nb_stride) + i];
    const int b = ST(LV_BIKES, nb);
    int accept = CAP(nb) - b;
    if (accept > number) accept = number;
    set_bikes(K, e, hd, nb, b + accept);
    const int target = CD(extra_cost_mode) == 0 ? src : CD(extra_cost_mode) == 1 ? cur : nb;
    ST(LV_EXTRA_COST, target) += accept * (i + 1);
    number -= accept;
  }
}

// _on_bike_returned :439-466 (deliver = false) and _on_bike_deliver :494-519 (deliver = true)
MRX_DEV void land_bikes(const CbParams& K, int e, int32_t* hd, bool deliver, int frm, int to, int n) {
  const int b = ST(LV_BIKES, to);
  int accept = CAP(to) - b;
  if (accept > n) accept = n;
  if (accept < n) {
    if (!deliver) ST(LV_FAILED_RETURN, to) += n - accept;
    move_to_neighbor(K, e, hd, frm, to, n - accept);
  }
  if (deliver && accept > 0) {
    ST(LV_TRANSFER_COST, to) += accept;
    HDR(CH_OPER) += accept;
  }
  set_bikes(K, e, hd, to, b + accept);
}

#endif

// The delivery pool: DeliverBike events in flight, appended in scheduling (= insertion) order, and — since a decision tick at the
// reference's topology size puts ~200 of them in flight per env — threaded by LANDING tick: slot (land mod CB_LAND_SLOTS) of the
// env's bucket table holds the first / last pool entry landing at that tick (entry word 5 = the next one), a 128-bit mask says
// which slots are occupied.  So "the deliveries landing at tick t, in insertion order" is a walk over exactly those entries and
// "the next landing tick" a bit scan, where a flat pool was scanned end to end for every event of a landing tick (measured on
// city.800s: 97 pool entries visited per trip / return event).  A slot only ever holds one landing tick: a delivery is pushed at
// tick now with land - now < CB_LAND_SLOTS (longer transfers are flagged as MRX_CB_ENV_DELIVERY_OVERFLOW and dropped), and every
// slot up to `now` has been flushed by then.
// bucket table words: [0, CB_LAND_SLOTS) first entry of a slot, [CB_LAND_SLOTS, 2 CB_LAND_SLOTS) last entry, then the occupancy mask
#define BKT_HEAD(slot) (slot)
#define BKT_TAIL(slot) (CB_LAND_SLOTS + (slot))
#define BKT_MASK(w) (2 * CB_LAND_SLOTS + (w))

// Executes, in insertion order, the deliveries landing at tick `t` whose scheduling tick is < sched_lt.
MRX_DEV void pool_exec_until(const CbParams& K, int e, int32_t* hd, int t, int sched_lt) {
  const int slot = t & (CB_LAND_SLOTS - 1);
  int idx = bkt_rd(K, e, BKT_HEAD(slot));
  while (idx >= 0) {
    int w[CB_POOL_WORDS];
    if (!pool_rd_entry(K, e, idx, t, sched_lt, w)) break;
    land_bikes(K, e, hd, true, w[2], w[3], w[4]);
    pool_wr(K, e, idx, 4, -1);
    idx = w[5];
  }
  bkt_wr(K, e, BKT_HEAD(slot), idx);
  if (idx < 0) {
    bkt_wr(K, e, BKT_TAIL(slot), -1);
    bkt_wr(K, e, BKT_MASK(slot >> 5), bkt_rd(K, e, BKT_MASK(slot >> 5)) & ~(int32_t)(1u << (slot & 31)));
  }
}

// After an execution at landing tick `from` (the old minimum): give executed entries at the head of the ring back, and find the
// next landing tick — slot `from` itself if it still holds entries, else the next occupied slot within the window.
MRX_DEV void pool_compact(const CbParams& K, int e, int32_t* hd) {
  int head = HDR(CH_POOL_HEAD);
  const int tail = HDR(CH_POOL_TAIL);
  while (head != tail && pool_rd(K, e, head % CD(pool_cap), 4) < 0) head++;
  HDR(CH_POOL_HEAD) = head;
  int m = CB_NO_LAND;
  const int from = HDR(CH_POOL_MINLAND);
  if (from != CB_NO_LAND && head != tail) {
    const int s0 = from & (CB_LAND_SLOTS - 1);
    for (int k = 0; k <= CB_LAND_SLOTS / 32 && m == CB_NO_LAND; k++) {  // mask words from s0's word on, wrapping once
      const int w = ((s0 >> 5) + k) & (CB_LAND_SLOTS / 32 - 1);
      uint32_t bits = (uint32_t)bkt_rd(K, e, BKT_MASK(w));
      if (k == 0) bits &= ~0u << (s0 & 31);
      if (k == CB_LAND_SLOTS / 32) bits &= (s0 & 31) ? ((1u << (s0 & 31)) - 1u) : 0u;
      if (bits) {
        int j = 0;
        while (!(bits >> j & 1u)) j++;
        m = pool_rd(K, e, bkt_rd(K, e, BKT_HEAD(w * 32 + j)), 0);
      }
    }
  }
  HDR(CH_POOL_MINLAND) = m;
}

// `now`: the tick the delivery is scheduled at (sched)
MRX_DEV void pool_push(const CbParams& K, int e, int32_t* hd, int land, int sched, int frm, int to, int n) {
  const int head = HDR(CH_POOL_HEAD), tail = HDR(CH_POOL_TAIL);
  if (tail - head >= CD(pool_cap) || land - sched >= CB_LAND_SLOTS) {
    HDR(CH_STATUS) |= MRX_CB_ENV_DELIVERY_OVERFLOW;
    return;
  }
  const int idx = tail % CD(pool_cap);
  pool_wr(K, e, idx, 0, land); pool_wr(K, e, idx, 1, sched); pool_wr(K, e, idx, 2, frm); pool_wr(K, e, idx, 3, to); pool_wr(K, e, idx, 4, n);
  pool_wr(K, e, idx, 5, -1);
  const int slot = land & (CB_LAND_SLOTS - 1);
  const int last = bkt_rd(K, e, BKT_TAIL(slot));
  if (last >= 0) pool_wr(K, e, last, 5, idx);
  else { bkt_wr(K, e, BKT_HEAD(slot), idx); bkt_wr(K, e, BKT_MASK(slot >> 5), bkt_rd(K, e, BKT_MASK(slot >> 5)) | (int32_t)(1u << (slot & 31))); }
  bkt_wr(K, e, BKT_TAIL(slot), idx);
  HDR(CH_POOL_TAIL) = tail + 1;
  if (land < HDR(CH_POOL_MINLAND)) HDR(CH_POOL_MINLAND) = land;
}

// DeliverBike events (this env's pool) due before the records of tick t: landing ticks that have no record of their own,
// oldest first — nothing else happens in between, so running them late changes nothing
MRX_DEV void pool_flush_before(const CbParams& K, int e, int32_t* hd, int t) {
  while (HDR(CH_POOL_MINLAND) < t) {
    pool_exec_until(K, e, hd, HDR(CH_POOL_MINLAND), CB_NO_LAND);
    pool_compact(K, e, hd);
  }
}
// ... and, when tick t's scheduled returns are over, the rest of the deliveries landing at t itself
MRX_DEV void pool_flush_at(const CbParams& K, int e, int32_t* hd, int t) {
  pool_flush_before(K, e, hd, t);
  if (HDR(CH_POOL_MINLAND) == t) {
    pool_exec_until(K, e, hd, t, CB_NO_LAND);
    pool_compact(K, e, hd);
  }
}

// One LIGHT record of the event stream at tick t: RequireBike :398-437 (a = trip index, b = src station) or ReturnBike
// :439-466 (a = trip index, b = the tick it was scheduled at, c = src | dst << 16).  Both touch one station's dock: everything
// either may need is READ up front (independent LDS reads, one latency), the common outcomes are computed without branches
// and only the writes differ — a lane alone on its SIMD pays every dependent LDS round trip in full.
// `minland`: the caller's register copy of HDR(CH_POOL_MINLAND) — only the delivery pool's own operations change that word, so the
// replay loop reads it from LDS once per run of light records instead of once per record.
MRX_DEV void light_event(const CbParams& K, int e, int32_t* hd, int t, int kind, int a, int b, int c, int& minland) {
  const bool trip = kind == CB_EV_TRIP;
  if (minland <= t) {  // (rare) this env's DeliverBike events that run first
    if (kind == CB_EV_RET) {
      // events queued by earlier ticks run by (scheduling tick, ReturnBike before DeliverBike, insertion order)
      pool_flush_before(K, e, hd, t);
      if (HDR(CH_POOL_MINLAND) == t) pool_exec_until(K, e, hd, t, b);
    } else {
      pool_flush_at(K, e, hd, t);
    }
    minland = HDR(CH_POOL_MINLAND);
  }
  const int st = trip ? b : (int)((uint32_t)c >> 16);  // the station whose dock changes
  const int fw = (a & CD(w_mask)) >> 5;
  const uint32_t bit = 1u << (a & 31);
  uint32_t ful = FUL(fw);
  int bikes = ST(LV_BIKES, st);
  int cap = CAP(st);
  CB_LAND(ful); CB_LAND(bikes); CB_LAND(cap);
  if (trip) {  // the only value a trip needs is the dock's bike count: everything else is an update that nobody waits for
    const int ok = bikes >= 1 ? 1 : 0;
    CB_ADD(ST(LV_TRIP_REQUIREMENT, st), 1);
    CB_ADD(ST(LV_SHORTAGE, st), 1 - ok);
    CB_ADD(ST(LV_FULFILLMENT, st), ok);
    CB_ADD(HDR(CH_TRIPS), 1);
    CB_ADD(HDR(CH_SHORT), 1 - ok);
    const int nb = bikes - ok;  // station.py:71-75 (min_bikes <= bikes always, so the min is a no-op when nothing left)
    ST(LV_BIKES, st) = nb;
    CB_MIN(ST(LV_MIN_BIKES, st), nb);
    CB_OR(FUL(fw), ok ? bit : 0u);
    CB_AND(FUL(fw), ok ? ~0u : ~bit);
  } else if (ful & bit) {  // the trip did get a bike: it comes back now
    if (bikes < cap) ST(LV_BIKES, st) = bikes + 1;
    else land_bikes(K, e, hd, false, c & 0xffff, st, 1);  // full dock: failed return, on to the neighbours
  }
}

// RebalanceBike :468-492 + decision_strategy.py:229-251: which stations ask for a decision (decided BEFORE the tick's
// zero-duration returns run)
MRX_DEV void rebalance_check(const CbParams& K, int e, int32_t* hd, int t) {
  if (HDR(CH_POOL_MINLAND) <= t) pool_flush_at(K, e, hd, t);
  for (int w = 0; w < CD(mask_words); w++) {
    uint32_t sup = 0, dem = 0;
    for (int j = 0; j < 32 && w * 32 + j < CD(S); j++) {
      const int s = w * 32 + j;
      const double ratio = (double)ST(LV_BIKES, s) / (double)CAP(s);
      if (ratio >= K.supply_wm) sup |= 1u << j;
      else if (ratio <= K.demand_wm) dem |= 1u << j;
    }
    DMK(w) = sup;
    DMK(CD(mask_words) + w) = dem;
  }
}

// The tick at which the (complete) frame fi was snapshot: the one tick of its range with (tick + 1) % res == 0 (post_step
// :130-147) — what the ring slot's tick word holds, without the read.
MRX_DEV int snapshot_tick(const CbParams& K, int fi) {
  const int lo = CD(start_tick) + fi * CD(res);
  return lo + (CD(res) - 1 - lo % CD(res));
}

// np_backend.pyx:481-518 — frame `fi` goes to ring slot fi % ring_slots (frames are taken in increasing order)
MRX_DEV void take_snapshot(const CbParams& K, int e, int32_t* hd, int t) {
  const int fi = (t - CD(start_tick)) / CD(res), slot = fi % CD(ring_slots);
  for (int w = 0; w < CD(FW); w++) RINGW(slot, w) = LIVE(w);
  RINGW(slot, CD(FW)) = t;
  K.ring_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), slot, e)] = fi;
}

// the end of tick t (flags of its TICK_END record: 1 = a frame ends here, 2 = last tick); returns true when the episode is over
MRX_DEV bool end_tick(const CbParams& K, int e, int32_t* hd, int t, int ev_flags) {
  if (HDR(CH_LATE) > 0) {  // DeliverBike appended to this very tick (transfer time 0)
    pool_exec_until(K, e, hd, t, CB_NO_LAND);
    pool_compact(K, e, hd);
    HDR(CH_LATE) = 0;
  }
  const bool frame_end = (ev_flags & 1) != 0;  // post_step :130-147
  if (frame_end) {
    take_snapshot(K, e, hd, t);
    for (int s = 0; s < CD(S); s++) {
      ST(LV_SHORTAGE, s) = 0; ST(LV_TRIP_REQUIREMENT, s) = 0; ST(LV_EXTRA_COST, s) = 0; ST(LV_TRANSFER_COST, s) = 0;
      ST(LV_FULFILLMENT, s) = 0; ST(LV_FAILED_RETURN, s) = 0;
      ST(LV_MIN_BIKES, s) = ST(LV_BIKES, s);
    }
  }
  if (ev_flags & 2) {
    if (!frame_end) take_snapshot(K, e, hd, t);  // core.py:371-375: the last, partial frame
    return true;
  }
  return false;
}

MRX_DEV int next_decision(const CbParams& K, int e, int32_t* hd, int* type) {
  for (int w = 0; w < CD(mask_words); w++) {
    const uint32_t sup = DMK(w), dem = DMK(CD(mask_words) + w);
    const uint32_t any = sup | dem;
    if (any) {
      int j = 0;
      while (!(any >> j & 1u)) j++;
      *type = (sup >> j & 1u) ? MRX_CB_SUPPLY : MRX_CB_DEMAND;
      return w * 32 + j;
    }
  }
  return -1;
}

// The best `n_out` of the `n` rows of the scope work arrays (key, val, trips) come first, best first.
// mode 0: (val, key) descending   1: (trips, key) ascending   2: (trips, key) descending.  Keys are distinct, so the order is total.
// (Measured and rejected: ranking all rows in registers, n^2 branch-free compares — 576 for 24 neighbours — is slower than this
// selection sort over LDS; a wave alone on its SIMD pays ~13 cycles per instruction whatever it is.)
MRX_DEV void scope_select(const CbParams& K, int e, int n, int n_out, int mode) {
  const int S = CD(S);
  for (int j = 0; j < n_out; j++) {
    int best = j;
    int bk = SCR(j), bv = mode == 0 ? SCR(S + j) : SCR(2 * S + j);
    for (int i = j + 1; i < n; i++) {
      const int ik = SCR(i), iv = mode == 0 ? SCR(S + i) : SCR(2 * S + i);
      const bool better = mode == 1 ? (iv < bv || (iv == bv && ik < bk)) : (iv > bv || (iv == bv && ik > bk));
      if (better) { best = i; bk = ik; bv = iv; }
    }
    if (best != j) {
      for (int a = 0; a < 3; a++) {
        const int tmp = SCR(a * S + j);
        SCR(a * S + j) = SCR(a * S + best);
        SCR(a * S + best) = tmp;
      }
    }
  }
}

// BikeDecisionStrategy.action_scope (decision_strategy.py:253-293) for station s at tick t.  Evaluated at every
// decision (the reference evaluates it lazily when the agent reads DecisionEvent.action_scope; reading it is
// what feeds the TripsWindowFilter cache, :131-138).  Writes ordered (station, max) pairs; returns their count.
MRX_DEV int action_scope(const CbParams& K, int e, int32_t* hd, int s, int type, int t, int32_t* out, Prof& P) {
  const int S = CD(S);
  int n = K.nb_cnt[s];
  for (int i = 0; i < n; i++) {
    const int nb = K.nb[(size_t)s * CD(nb_stride) + i];
    SCR(i) = nb;
    SCR(S + i) = type == MRX_CB_SUPPLY ? CAP(nb) - ST(LV_BIKES, nb) : (int)floor((double)ST(LV_BIKES, nb) * K.scope_high);
  }
  P.mark(9);
  const int fi_cur = (t - CD(start_tick)) / CD(res);
  for (int f = 0; f < CD(n_filters); f++) {
    const int n_out = CDA(f_num, f) < n ? CDA(f_num, f) : n;
    if (CDA(f_type, f) == MRX_CB_FILTER_DISTANCE) {
      n = n_out;  // still in distance order (cb_plan rejects a distance filter after a reordering one)
    } else if (CDA(f_type, f) == MRX_CB_FILTER_REQUIREMENTS) {
      scope_select(K, e, n, n_out, 0);
      n = n_out;
      P.mark(10);
    } else {
      // TripsWindowFilter :88-163: sum trip_requirement over the latest frames, the current (pre-decision
      // snapshot, aliased to the live frame) included; a frame's value is frozen the first time it is seen,
      // except that the newest frame of the list is always re-read.  `lst[-0:]` is the whole list in Python,
      // so windows == 0 means "every frame, none of them special".
      const int avail = fi_cur + 1 < CD(ring_slots) ? fi_cur + 1 : CD(ring_slots);
      const int aw = CDA(f_win, f) < avail ? CDA(f_win, f) : avail;
      const int first = aw > 0 ? fi_cur - aw + 1 : fi_cur - avail + 1;
      // trip_requirement counts EVERY RequireBike since the last frame reset, fulfilled or not: like trips_adj it is a
      // function of (shared trip table, tick).  So the filter's per-frame cache is one word — the tick the frame was last
      // read at — and a value is two reads of the shared running counts req_cum[tick][station]; a frame never seen as the
      // current one is read at the tick its snapshot was taken.
      if (aw > 0 && aw <= CB_TWC_REG) {
        // the usual case (a window of a few frames): first where each frame's two rows of req_cum are (per-env words, all
        // requested at once), then per candidate ONE batch of independent reads — as a plain frame-by-candidate loop every
        // read was a dependent L2 round trip, a hundred of them per decision
        int hi_o[CB_TWC_REG], lo_o[CB_TWC_REG];
#pragma unroll
        for (int k = 0; k < CB_TWC_REG; k++) {
          hi_o[k] = lo_o[k] = 0;  // (frames beyond the window read row 0 twice: + 0)
          if (k < aw) {
            const int fi = fi_cur - k, slot = fi % CD(ring_slots);
            if (k == 0) { TWCT(slot) = t; TWCF(slot) = fi; }
            const int tb = TWCF(slot) == fi ? TWCT(slot) : snapshot_tick(K, fi);
            int w0 = tb / CD(res) * CD(res);
            if (w0 < CD(start_tick)) w0 = CD(start_tick);
            hi_o[k] = (tb + 1 - CD(start_tick)) * S;
            lo_o[k] = (w0 - CD(start_tick)) * S;
          }
        }
        P.mark(11);
        for (int i = 0; i < n; i++) {
          const int x = SCR(i);
          int sum = 0;
#pragma unroll
          for (int k = 0; k < CB_TWC_REG; k++) sum += K.req_cum[hi_o[k] + x] - K.req_cum[lo_o[k] + x];
          SCR(2 * S + i) = sum;
        }
      } else {
        for (int i = 0; i < n; i++) SCR(2 * S + i) = 0;
        for (int fi = first; fi <= fi_cur; fi++) {
          const int slot = fi % CD(ring_slots);
          if (fi == fi_cur && (aw > 0 || TWCF(slot) != fi)) { TWCT(slot) = t; TWCF(slot) = fi; }
          const int tb = TWCF(slot) == fi ? TWCT(slot) : snapshot_tick(K, fi);
          int w0 = tb / CD(res) * CD(res);  // the frame reset before tick tb happened at the end of tick w0 - 1 (post_step :130-147)
          if (w0 < CD(start_tick)) w0 = CD(start_tick);
          const int32_t* hi = K.req_cum + (size_t)(tb + 1 - CD(start_tick)) * S;
          const int32_t* lo = K.req_cum + (size_t)(w0 - CD(start_tick)) * S;
          for (int i = 0; i < n; i++) { const int x = SCR(i); SCR(2 * S + i) += hi[x] - lo[x]; }
        }
      }
      P.mark(12);
      scope_select(K, e, n, n_out, type == MRX_CB_DEMAND ? 2 : 1);
      n = n_out;
      P.mark(13);
    }
  }
  for (int i = 0; i < n; i++) { out[2 * i] = SCR(i); out[2 * i + 1] = SCR(S + i); }
  out[2 * n] = s;
  out[2 * n + 1] = type == MRX_CB_SUPPLY ? (int)floor((double)ST(LV_BIKES, s) * K.scope_low_keep) : CAP(s) - ST(LV_BIKES, s);
  for (int i = n + 1; i < CD(scope_cap); i++) { out[2 * i] = -1; out[2 * i + 1] = -1; }
  return n + 1;
}

// _on_action_received :521-559 for the pending decision of station `s` at tick t
// The answer a deferred env was given (mrx_cb_set_replay_period: cb::defer_env_wave put it there): copies it out, returns its length
MRX_DEV int stash_take(const CbParams& K, int e, int32_t* hd, int32_t* sa) {
#pragma unroll
  for (int i = 0; i < CB_STASH_MAX * 3; i++) sa[i] = K.stash[CB_IX(CD(aos), CD(stride), (CB_STASH_MAX * 3), i, e)];
  const int n = HDR(CH_RES1);
  return n < 0 ? 0 : n > CB_STASH_MAX ? CB_STASH_MAX : n;
}

MRX_DEV void apply_actions(const CbParams& K, int e, int32_t* hd, int t, int s, const int32_t* actions, int n_actions) {
  // pop the decision from the tick's list
  DMK(s >> 5) &= ~(1u << (s & 31));
  DMK(CD(mask_words) + (s >> 5)) &= ~(1u << (s & 31));
  // Reference behaviour (event_linked_list.py:86-108): when the answered decision was the LAST element of the
  // tick's list, popping it leaves `_tail` on the removed node, so a DeliverBike appended to this same tick
  // (transfer time 0) is linked behind it and never runs.  It does run when later decisions, or an earlier
  // zero-time delivery, still follow it in the list.
  int ty;
  const bool tail_stale = next_decision(K, e, hd, &ty) < 0 && HDR(CH_LATE) == 0;
  for (int a = 0; a < n_actions; a++) {
    const int frm = actions[3 * a], to = actions[3 * a + 1], number = actions[3 * a + 2];
    if (frm < 0 || to < 0) continue;
    if (frm >= CD(S) || to >= CD(S)) { HDR(CH_STATUS) |= MRX_CB_ENV_INVALID_ACTION; continue; }
    const int b = ST(LV_BIKES, frm);
    const int ex = b < number ? b : number;
    if (ex <= 0) continue;
    set_bikes(K, e, hd, frm, b - ex);
    const int pos = HDR(CH_TT_POS);
    int tt = 1;
    if (pos < CD(tt_cap)) tt = K.tt[CB_IX(CD(aos), CD(stride), CD(tt_cap), pos, e)];
    else HDR(CH_STATUS) |= MRX_CB_ENV_TRANSFER_TIMES_OUT;
    HDR(CH_TT_POS) = pos + 1;
    if (tt < 0 || t + tt >= CD(max_tick)) continue;  // lands in the past / after the episode: never executed
    if (tt == 0) {
      if (tail_stale) continue;
      HDR(CH_LATE) += 1;
    }
    pool_push(K, e, hd, t + tt, t, frm, to, ex);
  }
}

// Env.step for one env.  Sequential mode: dec[8], scope[scope_cap][2], actions[A][3], n_actions = their count (n_act_ev unused).
// Joint modes: dec[S][8], scope[S][scope_cap][2], actions[S][A][3], n_act_ev[S] = the action count of each reported event,
// n_actions = how many of the reported events were answered.  met[3].
MRX_DEV void step_env(const CbParams& K, int e, const int32_t* actions, int n_actions, const int32_t* n_act_ev, int32_t* dec, int32_t* scope,
                      int64_t* met, uint8_t* done) {
#ifdef MRX_CB_LDSFRAME
  int32_t* hd = nullptr;  // (the header lives in the LDS column, HDR())
#ifndef __HIPCC__
  for (int w = 0; w < CH_WORDS; w++) HDR(w) = GHDR(w);
#endif
#else
  int32_t hd[CH_WORDS];  // the env's header lives in registers for the whole step
#pragma unroll
  for (int w = 0; w < CH_WORDS; w++) hd[w] = GHDR(w);
#endif
  int flags = HDR(CH_FLAGS);
  int t = HDR(CH_TICK);
  bool finished = (flags & CFL_FINISHED) != 0;
  if (!finished) {
    Prof P;
    EvWin W;
    W.open(K, HDR(CH_EV_POS));
#if defined(MRX_CB_LDSFRAME) && !defined(__HIPCC__) /* (the HIP kernel moves the state with all 64 lanes: cb_step_kernels.h) */
    for (int w = 0; w < MRXC_FW; w++) LF(w) = K.live[CB_IX(CD(aos), CD(stride), CD(FW), w, e)];
    for (int w = 0; w < MRXC_S; w++) LF(LDS_CAP + w) = K.capacity[w];
    for (int w = 0; w < MRXC_w_words; w++) LF(LDS_FUL + w) = (int32_t)GFUL(w);
    for (int w = 0; w < 2 * MRXC_mask_words; w++) LF(LDS_DMK + w) = (int32_t)GDMK(w);
#ifdef MRX_CB_TWC_LDS
    for (int w = 0; w < MRXC_ring_slots; w++) { TWCF(w) = K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, e)]; TWCT(w) = K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, e)]; }
#endif
#endif
    P.mark(0);
    // a paused env stands AT the TICK_END record of its decision tick, with that tick's deliveries already done
    bool resumed = (flags & CFL_PENDING) != 0;
    if (resumed) {
      if (CD(decision_mode) == 0) {
        if (flags & CFL_STASH) {  // the env was deferred: its answer came with an earlier call
          int32_t sa[CB_STASH_MAX * 3];
          const int ns = stash_take(K, e, hd, sa);
          apply_actions(K, e, hd, t, HDR(CH_CUR_STATION), sa, ns);
        } else {
          apply_actions(K, e, hd, t, HDR(CH_CUR_STATION), actions, n_actions);
        }
      } else {
        // core.py:354-366: the agent's i-th action list goes to the i-th reported event; each answered decision event runs, is
        // popped, and its action event runs right behind it, in event (= station) order.  apply_actions' "was it the last
        // element of the tick's list" test sees the unanswered events still in the list, as in the reference.
        for (int i = 0; i < n_actions; i++) {
          int ty;
          const int s = next_decision(K, e, hd, &ty);
          if (s < 0) break;
          int na = n_act_ev ? n_act_ev[i] : 0;
          if (na > CD(max_actions)) na = CD(max_actions);
          apply_actions(K, e, hd, t, s, actions + (size_t)i * CD(max_actions) * 3, na < 0 ? 0 : na);
        }
        if (CD(decision_mode) == 1)  // Joint: the unanswered events are finished without effect
          for (int w = 0; w < 2 * CD(mask_words); w++) DMK(w) = 0;
      }
      flags &= ~(CFL_PENDING | CFL_STASH);
      P.mark(1);
    }
    flags &= ~CFL_FRESH;
    int dec_s = -1, dec_type = 0;
    // mrx_cb_set_step_budget: at most this many records per call; an env that has not reached its next decision by then
    // reports "no decision yet" and goes on from the same place in the next call (any record boundary is a consistent state)
    int left = K.step_budget > 0 ? K.step_budget : 0x7fffffff;
    for (;;) {
      // ---- light records, one per iteration whatever their kind
      EvWin::Rec r = W.rec(K);
      int minland = HDR(CH_POOL_MINLAND);  // (apply_actions / the heavy records below may have changed it)
      while ((r.w0 & 7) != CB_EV_REBAL && (r.w0 & 7) != CB_EV_TICK_END && left > 0) {
        light_event(K, e, hd, CD(start_tick) + (r.w0 >> 3), r.w0 & 7, r.a, r.b, r.c, minland);
        W.advance(K);
        r = W.rec(K);
        left--;
      }
      const int w0 = r.w0;
      P.mark(2);
      // ---- (the wave reconverges here) the rare, heavy records
      t = CD(start_tick) + (w0 >> 3);
      if ((w0 & 7) != CB_EV_REBAL && (w0 & 7) != CB_EV_TICK_END) break;  // budget spent between two light records
      left -= 4;
      if ((w0 & 7) == CB_EV_REBAL) {
        rebalance_check(K, e, hd, t);
        W.advance(K);
        P.mark(3);
        continue;
      }
      if (!resumed && HDR(CH_POOL_MINLAND) <= t) pool_flush_at(K, e, hd, t);
      resumed = false;
      dec_s = next_decision(K, e, hd, &dec_type);
      if (dec_s >= 0) break;
      const bool over = end_tick(K, e, hd, t, r.a);
      P.mark(4);
      if (over) {
        flags |= CFL_FINISHED;
        finished = true;
        break;
      }
      W.advance(K);
    }
    P.mark(5);
    if (dec_s >= 0) {  // (reconverged again: every lane of the wave that found a decision computes its scope together)
      // core.py:345 takes a snapshot of the current frame here; queries alias it to the live frame instead — the frame's own
      // post_step snapshot comes later and overwrites it anyway.  Except when start_tick is not a multiple of the snapshot
      // resolution: then a frame's post_step snapshot falls on an EARLIER tick of the frame and this one is what stays.
      if (t > snapshot_tick(K, (t - CD(start_tick)) / CD(res))) take_snapshot(K, e, hd, t);
      flags |= CFL_PENDING;
      HDR(CH_CUR_STATION) = dec_s;
      HDR(CH_CUR_TYPE) = dec_type;
      HDR(CH_NDEC) += 1;
      if (CD(decision_mode) == 0) {
        dec[0] = t; dec[1] = dec_s; dec[2] = dec_type; dec[3] = (t - CD(start_tick)) / CD(res);
        dec[4] = action_scope(K, e, hd, dec_s, dec_type, t, scope, P);
        dec[5] = 1; dec[6] = 0; dec[7] = 0;
      } else {
        // every pending event of the tick, in station order (the order _on_rebalance_bikes inserted them)
        int k = 0;
        for (int w = 0; w < CD(mask_words); w++) {
          const uint32_t sup = DMK(w), any = sup | DMK(CD(mask_words) + w);
          for (int j = 0; j < 32; j++) {
            if (!(any >> j & 1u)) continue;
            const int s = w * 32 + j, ty = (sup >> j & 1u) ? MRX_CB_SUPPLY : MRX_CB_DEMAND;
            int32_t* d = dec + (size_t)k * 8;
            d[0] = t; d[1] = s; d[2] = ty; d[3] = (t - CD(start_tick)) / CD(res);
            d[4] = action_scope(K, e, hd, s, ty, t, scope + (size_t)k * CD(scope_cap) * 2, P);
            d[5] = 1; d[7] = k;
            k++;
          }
        }
        for (int i = 0; i < k; i++) dec[(size_t)i * 8 + 6] = k;
        for (int i = k; i < CD(S); i++) dec[(size_t)i * 8 + 5] = 0;  // every row beyond the events: valid = 0 (the buffer is the caller's and may hold an earlier report)
        HDR(CH_NDEC) += k - 1;
      }
      P.mark(6);
    }
    HDR(CH_EV_POS) = W.pos;
    P.flush(K, e);
    HDR(CH_TICK) = t;
    HDR(CH_FLAGS) = flags;
#if !(defined(MRX_CB_LDSFRAME) && defined(__HIPCC__))
    for (int w = 0; w < CH_WORDS; w++) GHDR(w) = HDR(w);
#endif
#if defined(MRX_CB_LDSFRAME) && !defined(__HIPCC__)
    for (int w = 0; w < MRXC_FW; w++) K.live[CB_IX(CD(aos), CD(stride), CD(FW), w, e)] = LF(w);
    for (int w = 0; w < MRXC_w_words; w++) GFUL(w) = (uint32_t)LF(LDS_FUL + w);
    for (int w = 0; w < 2 * MRXC_mask_words; w++) GDMK(w) = (uint32_t)LF(LDS_DMK + w);
#ifdef MRX_CB_TWC_LDS
    for (int w = 0; w < MRXC_ring_slots; w++) { K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, e)] = TWCF(w); K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, e)] = TWCT(w); }
#endif
#endif
  }
  if (finished || !(flags & CFL_PENDING)) {  // episode over, or (step budget) no decision reached yet
    dec[0] = t; dec[1] = -1; dec[2] = -1; dec[3] = (t - CD(start_tick)) / CD(res); dec[4] = 0; dec[5] = 0; dec[6] = 0; dec[7] = 0;
    for (int i = 0; i < CD(scope_cap); i++) { scope[2 * i] = -1; scope[2 * i + 1] = -1; }
    if (CD(decision_mode) != 0)
      for (int i = 1; i < CD(S); i++) dec[(size_t)i * 8 + 5] = 0;  // Joint modes: no row of an earlier report stays valid
  }
  met[0] = HDR(CH_TRIPS); met[1] = HDR(CH_SHORT); met[2] = HDR(CH_OPER);
  *done = finished ? 1 : 0;
}

// ==========================================================================================
// WAVE-COOPERATIVE pieces (one env per wave64; written against wave::{lane, sync, ballot, bcast} — the includer brings wave.h or,
// in tests/emu, the 64-fiber emulation).  See cb_wave.h for why: at the reference's own topology size a decision tick raises
// hundreds of decision events per env, and the action scope's per-lane selection sort / the station sweeps dominate everything.
enum { CBW_MAX = 128 };  // candidates ranked in registers (two per lane)

// rank[a] = how many of the n candidates come before candidate (a, lane) — mode 0 / 2: (v, key) descending, 1: ascending
MRX_DEV void cbw_rank(const CbParams& K, int n, int mode, const int* v, const int* key, int* rank) {
  rank[0] = rank[1] = 0;
  // The usual case — values of at most 20 bits (bike counts, trip sums), keys = distinct station numbers below 2048 — packs
  // (v, key) into ONE word that orders the same way: a broadcast, a compare and an add-with-carry per candidate and side, eight
  // candidates per loop trip.  The wave kernels run four waves per SIMD at 4096 envs and this loop is most of their instructions.
  // Positions >= n hold a word that comes before nobody, so the trip count rounds up freely.
  if (CD(S) <= 2048 && !wave::ballot(((((uint32_t)v[0] | (uint32_t)v[1]) >> 20) != 0u))) {
    const int lane = wave::lane();
    const uint32_t flip = mode == 1 ? 0x7fffffffu : 0u;  // ascending = descending on the complemented word (31 bits: 0 stays the smallest)
    uint32_t c[2];
#pragma unroll
    for (int a = 0; a < 2; a++) c[a] = a * 64 + lane < n ? (((((uint32_t)v[a] << 11) | (uint32_t)key[a]) ^ flip) + 1u) : 0u;
    for (int j0 = 0; j0 < n; j0 += 8) {  // wave-uniform; a trip stays on one side (64 % 8 == 0)
      const uint32_t src = j0 < 64 ? c[0] : c[1];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t cj = (uint32_t)wave::bcast((int)src, (j0 + k) & 63);
#pragma unroll
        for (int a = 0; a < 2; a++) rank[a] += cj > c[a] ? 1 : 0;
      }
    }
    return;
  }
  for (int j = 0; j < n; j++) {  // wave-uniform
    const int vj = wave::bcast(j < 64 ? v[0] : v[1], j & 63), kj = wave::bcast(j < 64 ? key[0] : key[1], j & 63);
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const bool before = mode == 1 ? (vj < v[a] || (vj == v[a] && kj < key[a])) : (vj > v[a] || (vj == v[a] && kj > key[a]));
      rank[a] += before ? 1 : 0;
    }
  }
}

// keep the n_out best of n candidates, best first: survivors move to position = rank through the LDS scratch (2 x CBW_MAX words)
MRX_DEV void cbw_select(const CbParams& K, int32_t* scr, int n, int n_out, int mode, const int* v, int* key, int* val) {
  const int lane = wave::lane();
  int rank[2];
  cbw_rank(K, n, mode, v, key, rank);
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

// candidates of station s after the leading distance filters (a prefix of the distance-sorted list); *nf0 = first other filter
MRX_DEV int cbw_candidates(const CbParams& K, int s, int* nf0) {
  int n = K.nb_cnt[s], f = 0;
  while (f < CD(n_filters) && CDA(f_type, f) == MRX_CB_FILTER_DISTANCE) {
    n = CDA(f_num, f) < n ? CDA(f_num, f) : n;
    f++;
  }
  *nf0 = f;
  return n;
}

// can scope_wave evaluate (s, t)?  At most CBW_MAX candidates, at most 64 frames in a trip window.
// The wave kernels' form of the fused observation (env-major plans): row i of an env = the station of row i of the decision's ACTION
// SCOPE (the deciding station and its filtered neighbours — what an agent can act on; -1 padding and envs without a decision: zeros),
// i.e. what mrx_cb_query("stations", decision frame, scope[:, :, 0], attrs) returns.  live(lv, station) reads the env's current state.
template <class LV>
MRX_DEV void write_scope_observation_row(const CbParams& K, int e, int i, int st, int t, LV live) {
  double* o = K.obs + ((size_t)e * CD(scope_cap) + (size_t)i) * K.obs_n;
  for (int a = 0; a < K.obs_n; a++) {
    const int attr = K.obs_attr[a];
    double v = 0.0;
    if (st >= 0 && st < CD(S)) {
      switch (attr) {
        case SA_BIKES: v = (double)live(LV_BIKES, st); break;
        case SA_SHORTAGE: v = (double)live(LV_SHORTAGE, st); break;
        case SA_TRIP_REQUIREMENT: v = (double)live(LV_TRIP_REQUIREMENT, st); break;
        case SA_FULFILLMENT: v = (double)live(LV_FULFILLMENT, st); break;
        case SA_EXTRA_COST: v = (double)live(LV_EXTRA_COST, st); break;
        case SA_TRANSFER_COST: v = (double)live(LV_TRANSFER_COST, st); break;
        case SA_FAILED_RETURN: v = (double)live(LV_FAILED_RETURN, st); break;
        case SA_MIN_BIKES: v = (double)live(LV_MIN_BIKES, st); break;
        case SA_CAPACITY: v = (double)K.capacity[st]; break;
        case SA_ID: v = (double)K.station_id[st]; break;
        default: v = (double)K.cal[(size_t)K.tick_day[t - CD(start_tick)] * 4 + (attr - SA_WEEKDAY)]; break;
      }
    }
    o[a] = v;
  }
}

MRX_DEV bool scope_wave_ok(const CbParams& K, int s, int t) {
  int nf0;
  if (cbw_candidates(K, s, &nf0) > CBW_MAX) return false;
  const int fi_cur = (t - CD(start_tick)) / CD(res);
  for (int f = nf0; f < CD(n_filters); f++)
    if (CDA(f_type, f) == MRX_CB_FILTER_TRIP_WINDOW) {
      const int avail = fi_cur + 1 < CD(ring_slots) ? fi_cur + 1 : CD(ring_slots);
      const int aw = CDA(f_win, f) < avail ? CDA(f_win, f) : avail;
      if ((aw > 0 ? aw : avail) > 64) return false;
    }
  return true;
}

// BikeDecisionStrategy.action_scope (decision_strategy.py:253-293) of station s at tick t, across the lanes of a wave: the same
// values, order and cache side effects as cb::action_scope above.  bikes(station) = the env's current bikes; tag_get(slot, &fi,
// &tick) / tag_set(slot, fi, tick) = the trip-window filter's per-slot cache words.  Writes `out`, returns the number of rows.
// row(i, station) is called by the lane that writes scope row i (the fused observation of the wave kernels hooks in there).
template <class BK, class TG, class TS, class RW>
MRX_DEV int scope_wave(const CbParams& K, int s, int type, int t, int32_t* scr, int32_t* out, BK bikes, TG tag_get, TS tag_set, RW row) {
  const int lane = wave::lane();
  const int S = CD(S);
  int nf0;
  int n = cbw_candidates(K, s, &nf0);
  const int fi_cur = (t - CD(start_tick)) / CD(res);
  int key[2], val[2];
#pragma unroll
  for (int a = 0; a < 2; a++) {  // candidate i = a * 64 + lane
    const int i = a * 64 + lane;
    const int nb = K.nb[(size_t)s * CD(nb_stride) + (i < n ? i : 0)];
    const int bk = bikes(nb);
    key[a] = i < n ? nb : -1;
    val[a] = type == MRX_CB_SUPPLY ? K.capacity[nb] - bk : (int)floor((double)bk * K.scope_high);
  }
  for (int f = nf0; f < CD(n_filters); f++) {  // wave-uniform
    const int n_out = CDA(f_num, f) < n ? CDA(f_num, f) : n;
    if (CDA(f_type, f) == MRX_CB_FILTER_DISTANCE) {
      // (cb_plan rejects a distance filter behind a reordering one; a later prefix cut of a still distance-ordered list)
    } else if (CDA(f_type, f) == MRX_CB_FILTER_REQUIREMENTS) {
      const int v0[2] = {val[0], val[1]};
      cbw_select(K, scr, n, n_out, 0, v0, key, val);
    } else {
      // TripsWindowFilter :88-163 — see cb::action_scope: a frame's value is frozen at the tick it was last read as the
      // current frame; the newest frame is re-read now (with windows == 0, Python's lst[-0:], only when it is new)
      const int avail = fi_cur + 1 < CD(ring_slots) ? fi_cur + 1 : CD(ring_slots);
      const int aw = CDA(f_win, f) < avail ? CDA(f_win, f) : avail;
      const int cnt = aw > 0 ? aw : avail;
      int hi_o = 0, lo_o = 0;
      if (lane < cnt) {  // lane k: frame fi_cur - k
        const int fi = fi_cur - lane, slot = fi % CD(ring_slots);
        int tag_fi, tag_t;
        tag_get(slot, &tag_fi, &tag_t);
        if (lane == 0 && (aw > 0 || tag_fi != fi)) { tag_fi = fi; tag_t = t; tag_set(slot, fi, t); }
        const int tb = tag_fi == fi ? tag_t : snapshot_tick(K, fi);
        int w0 = tb / CD(res) * CD(res);
        if (w0 < CD(start_tick)) w0 = CD(start_tick);
        hi_o = (tb + 1 - CD(start_tick)) * S;
        lo_o = (w0 - CD(start_tick)) * S;
      }
      int trips[2] = {0, 0};
      if (cnt <= CB_TWC_REG) {
        // the usual window: every read of every frame in flight before the first sum (lanes >= cnt hold row 0 twice: + 0) — frame
        // after frame each trip of the loop below waits for its own four reads, ten dependent round trips per decision
        int vh[CB_TWC_REG][2], vl[CB_TWC_REG][2];
#pragma unroll
        for (int k = 0; k < CB_TWC_REG; k++) {
          const int hi = wave::bcast(hi_o, k), lo = wave::bcast(lo_o, k);
#pragma unroll
          for (int a = 0; a < 2; a++) {
            const int x = key[a] >= 0 ? key[a] : 0;
            vh[k][a] = K.req_cum[hi + x];
            vl[k][a] = K.req_cum[lo + x];
          }
        }
#pragma unroll
        for (int k = 0; k < CB_TWC_REG; k++)
#pragma unroll
          for (int a = 0; a < 2; a++) trips[a] += vh[k][a] - vl[k][a];
      } else
      for (int k = 0; k < cnt; k++) {  // wave-uniform
        const int hi = wave::bcast(hi_o, k), lo = wave::bcast(lo_o, k);
#pragma unroll
        for (int a = 0; a < 2; a++) {
          const int x = key[a] >= 0 ? key[a] : 0;
          trips[a] += K.req_cum[hi + x] - K.req_cum[lo + x];
        }
      }
      cbw_select(K, scr, n, n_out, type == MRX_CB_DEMAND ? 2 : 1, trips, key, val);
    }
    n = n_out;
  }
#pragma unroll
  for (int a = 0; a < 2; a++) {
    const int i = a * 64 + lane;
    if (i < n) { out[2 * i] = key[a]; out[2 * i + 1] = val[a]; row(i, key[a]); }
  }
  const int bs = bikes(s);
  if (lane == 0) {
    out[2 * n] = s;
    out[2 * n + 1] = type == MRX_CB_SUPPLY ? (int)floor((double)bs * K.scope_low_keep) : K.capacity[s] - bs;
    row(n, s);
  }
  for (int i = n + 1 + lane; i < CD(scope_cap); i += 64) { out[2 * i] = -1; out[2 * i + 1] = -1; row(i, -1); }
  return n + 1;
}

#ifdef MRX_CB_LDSFRAME
// ---- the GENERAL step of one env on one wave, state in the wave's LDS column (plan-specialised LDS-frame build launched with one
// env per workgroup): the event replay, the delivery pool and the action stay scalar (lane 0 — they are strictly sequential),
// the station sweeps (rebalance check, snapshot, frame reset), the scan for the next decision and the action scope run across
// the 64 lanes.  Sequential decision mode, aligned frames (the host only launches it then).  Mirrors step_env above line by line.
#ifdef __HIPCC__
#define LW(w) mrx_cb_lds[CB_EV_BLOCK * 4 + (w)] /* lane-independent view of the wave's one column (LF() of lane 0, K.lsh = 0) */
#else
#define LW(w) mrx_cb_lds_host[CB_EV_BLOCK * 4 + (w)]
#endif
MRX_DEV void step_env_wave(const CbParams& K, int e, const int32_t* actions, int n_actions, int32_t* dec, int32_t* scope, int64_t* met, uint8_t* done,
                           int32_t* scr) {
  const int lane = wave::lane();
  const int S = CD(S), MW = CD(mask_words);
  int32_t* hd = nullptr;
  int32_t* ctl = scr + 2 * CBW_MAX;  // lane 0 -> wave: [0] w0 of the record it stopped at, [1] its a, [2] stream position, [3] budget left
#ifndef __HIPCC__
  if (lane == 0) {  // host harness: one env at a time through the static LDS stand-in
    for (int w = 0; w < CH_WORDS; w++) HDR(w) = GHDR(w);
    for (int w = 0; w < MRXC_FW; w++) LF(w) = K.live[CB_IX(CD(aos), CD(stride), CD(FW), w, e)];
    for (int w = 0; w < MRXC_S; w++) LF(LDS_CAP + w) = K.capacity[w];
    for (int w = 0; w < MRXC_w_words; w++) LF(LDS_FUL + w) = (int32_t)GFUL(w);
    for (int w = 0; w < 2 * MRXC_mask_words; w++) LF(LDS_DMK + w) = (int32_t)GDMK(w);
#ifdef MRX_CB_TWC_LDS
    for (int w = 0; w < MRXC_ring_slots; w++) { TWCF(w) = K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, e)]; TWCT(w) = K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, e)]; }
#endif
  }
  wave::sync();
#ifdef MRX_CB_POOL_LDS
  if (K.pool_stage) {
    PoolStage PS;
    pool_stage_fetch(K, e, GHDR(CH_POOL_HEAD), GHDR(CH_POOL_TAIL), PS);
    pool_stage_put(K, PS);
    evw_put(K, GHDR(CH_EV_POS), evw_fetch(K, GHDR(CH_EV_POS)));
  }
  wave::sync();
#endif
#endif
  int flags = LW(LDS_HDR + CH_FLAGS), t = LW(LDS_HDR + CH_TICK);
  bool finished = (flags & CFL_FINISHED) != 0;
  WProf WP;
  if (!finished) {
    int pos = LW(LDS_HDR + CH_EV_POS);
    bool resumed = (flags & CFL_PENDING) != 0;
    if (resumed) {
      if (lane == 0) {
        if (flags & CFL_STASH) {  // the env was deferred: its answer came with an earlier call
          int32_t sa[CB_STASH_MAX * 3];
          const int ns = stash_take(K, e, hd, sa);
          apply_actions(K, e, hd, t, HDR(CH_CUR_STATION), sa, ns);
        } else {
          apply_actions(K, e, hd, t, HDR(CH_CUR_STATION), actions, n_actions);
        }
      }
      flags &= ~(CFL_PENDING | CFL_STASH);
      wave::sync();
    }
    WP.mark(K, e, 1);
    flags &= ~CFL_FRESH;
    int dec_s = -1, dec_type = 0;
    int left = K.step_budget > 0 ? K.step_budget : 0x7fffffff;
    for (;;) {
      if (lane == 0) {  // ---- light records: strictly sequential
        auto light_run = [&](auto& W) {
          W.open(K, pos);
          EvWin::Rec r = W.rec(K);
          int minland = HDR(CH_POOL_MINLAND);
          while ((r.w0 & 7) != CB_EV_REBAL && (r.w0 & 7) != CB_EV_TICK_END && left > 0) {
            light_event(K, e, hd, CD(start_tick) + (r.w0 >> 3), r.w0 & 7, r.a, r.b, r.c, minland);
            W.advance(K);
            r = W.rec(K);
            left--;
          }
          W.close(K);
          ctl[0] = r.w0; ctl[1] = r.a; ctl[2] = W.pos; ctl[3] = left;
        };
#ifdef MRX_CB_POOL_LDS
        if (K.pool_stage) {
          EvWinW W;
          light_run(W);
        } else
#endif
        {
          EvWin W;
          light_run(W);
        }
      }
      wave::sync();
      const int w0 = ctl[0], ra = ctl[1];
      pos = ctl[2]; left = ctl[3];
      wave::sync();
      WP.mark(K, e, 2);
      t = CD(start_tick) + (w0 >> 3);
      if ((w0 & 7) != CB_EV_REBAL && (w0 & 7) != CB_EV_TICK_END) break;  // budget spent between two light records
      left -= 4;
      if ((w0 & 7) == CB_EV_REBAL) {
        // rebalance_check :468-492 + decision_strategy.py:229-251, lane = station
        if (lane == 0 && HDR(CH_POOL_MINLAND) <= t) pool_flush_at(K, e, hd, t);
        wave::sync();
        for (int c0 = 0; c0 < S; c0 += 64) {
          const int s = c0 + lane;
          bool sup = false, dem = false;
          if (s < S) {
            const double ratio = (double)LW(LV_BIKES * S + s) / (double)LW(LDS_CAP + s);
            sup = ratio >= K.supply_wm;
            dem = !sup && ratio <= K.demand_wm;
          }
          const uint64_t ms = wave::ballot(sup), md = wave::ballot(dem);
          if (lane == 0) {
            const int w = c0 >> 5;
            LW(LDS_DMK + w) = (int32_t)(uint32_t)(ms & 0xffffffffull); LW(LDS_DMK + MW + w) = (int32_t)(uint32_t)(md & 0xffffffffull);
            if (w + 1 < MW) { LW(LDS_DMK + w + 1) = (int32_t)(uint32_t)(ms >> 32); LW(LDS_DMK + MW + w + 1) = (int32_t)(uint32_t)(md >> 32); }
          }
        }
        wave::sync();
        pos++;
        WP.mark(K, e, 3);
        continue;
      }
      if (lane == 0 && !resumed && HDR(CH_POOL_MINLAND) <= t) pool_flush_at(K, e, hd, t);
      resumed = false;
      wave::sync();
      {  // next_decision: lowest pending station, Supply wins; one mask word per lane (two rounds above 2048 stations)
        dec_s = -1;
        for (int w0m = 0; w0m < MW && dec_s < 0; w0m += 64) {
          const int w = w0m + lane;
          const uint32_t sup = w < MW ? (uint32_t)LW(LDS_DMK + w) : 0u, any = sup | (w < MW ? (uint32_t)LW(LDS_DMK + MW + w) : 0u);
          const uint64_t pend = wave::ballot(any != 0u);
          if (pend) {
            const int l1 = __builtin_ctzll(pend);
            const uint32_t wa = (uint32_t)wave::bcast((int)any, l1), ws = (uint32_t)wave::bcast((int)sup, l1);
            const int j1 = __builtin_ctz(wa);
            dec_s = (w0m + l1) * 32 + j1;
            dec_type = (ws >> j1 & 1u) ? MRX_CB_SUPPLY : MRX_CB_DEMAND;
          }
        }
      }
      if (dec_s >= 0) { WP.mark(K, e, 4); break; }
      // ---- end_tick
      if (lane == 0 && HDR(CH_LATE) > 0) {  // DeliverBike appended to this very tick (transfer time 0)
        pool_exec_until(K, e, hd, t, CB_NO_LAND);
        pool_compact(K, e, hd);
        HDR(CH_LATE) = 0;
      }
      wave::sync();
      const bool frame_end = (ra & 1) != 0;
      if (frame_end || (ra & 2)) {  // take_snapshot: np_backend.pyx:481-518 (frame_end: post_step :130-147; last tick: core.py:371-375)
        const int fi = (t - CD(start_tick)) / CD(res), slot = fi % CD(ring_slots);
        for (int w = lane; w < CD(FW); w += 64) RINGW(slot, w) = LW(w);
        if (lane == 0) { RINGW(slot, CD(FW)) = t; K.ring_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), slot, e)] = fi; }
        wave::sync();  // every lane has read its words of the frame before the reset below rewrites them
      }
      if (frame_end) {
        for (int s = lane; s < S; s += 64) {
          LW(LV_SHORTAGE * S + s) = 0; LW(LV_TRIP_REQUIREMENT * S + s) = 0; LW(LV_EXTRA_COST * S + s) = 0; LW(LV_TRANSFER_COST * S + s) = 0;
          LW(LV_FULFILLMENT * S + s) = 0; LW(LV_FAILED_RETURN * S + s) = 0;
          LW(LV_MIN_BIKES * S + s) = LW(LV_BIKES * S + s);
        }
        wave::sync();
      }
      WP.mark(K, e, 4);
      if (ra & 2) {
        flags |= CFL_FINISHED;
        finished = true;
        break;
      }
      pos++;
    }
    if (dec_s >= 0) {
      flags |= CFL_PENDING;
      if (lane == 0) { HDR(CH_CUR_STATION) = dec_s; HDR(CH_CUR_TYPE) = dec_type; HDR(CH_NDEC) += 1; }
      int cnt;
      if (scope_wave_ok(K, dec_s, t)) {
        cnt = scope_wave(K, dec_s, dec_type, t, scr, scope,
                         [&](int st) { return LW(LV_BIKES * S + st); },
#ifdef MRX_CB_TWC_LDS
                         [&](int slot, int* fi, int* tk) { *fi = LW(LDS_TWC + slot); *tk = LW(LDS_TWC + MRXC_ring_slots + slot); },
                         [&](int slot, int fi, int tk) { LW(LDS_TWC + slot) = fi; LW(LDS_TWC + MRXC_ring_slots + slot) = tk; },
#else
                         [&](int slot, int* fi, int* tk) { *fi = K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), slot, e)]; *tk = K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), slot, e)]; },
                         [&](int slot, int fi, int tk) { K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), slot, e)] = fi; K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), slot, e)] = tk; },
#endif
                         [&](int i, int st) { if (K.obs) write_scope_observation_row(K, e, i, st, t, [&](int lv, int s2) { return LW(lv * S + s2); }); });
      } else {
        Prof P;
        if (lane == 0) {
          ctl[0] = action_scope(K, e, hd, dec_s, dec_type, t, scope, P);
          if (K.obs)   // (scalar fallback: the lane that wrote the scope rows reads them back)
            for (int i = 0; i < CD(scope_cap); i++) write_scope_observation_row(K, e, i, scope[2 * i], t, [&](int lv, int s2) { return LW(lv * S + s2); });
        }
        wave::sync();
        cnt = ctl[0];
      }
      if (lane == 0) {
        dec[0] = t; dec[1] = dec_s; dec[2] = dec_type; dec[3] = (t - CD(start_tick)) / CD(res); dec[4] = cnt; dec[5] = 1; dec[6] = 0; dec[7] = 0;
      }
      WP.mark(K, e, 5);
    }
    if (lane == 0) { HDR(CH_EV_POS) = pos; HDR(CH_TICK) = t; HDR(CH_FLAGS) = flags; }
    wave::sync();
#ifndef __HIPCC__
    if (lane == 0) {
      for (int w = 0; w < CH_WORDS; w++) GHDR(w) = HDR(w);
      for (int w = 0; w < MRXC_FW; w++) K.live[CB_IX(CD(aos), CD(stride), CD(FW), w, e)] = LF(w);
      for (int w = 0; w < MRXC_w_words; w++) GFUL(w) = (uint32_t)LF(LDS_FUL + w);
      for (int w = 0; w < 2 * MRXC_mask_words; w++) GDMK(w) = (uint32_t)LF(LDS_DMK + w);
#ifdef MRX_CB_TWC_LDS
      for (int w = 0; w < MRXC_ring_slots; w++) { K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, e)] = TWCF(w); K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), w, e)] = TWCT(w); }
#endif
    }
    wave::sync();
#endif
  }
  if (finished || !(flags & CFL_PENDING)) {  // episode over, or (step budget) no decision reached yet
    if (lane == 0) { dec[0] = t; dec[1] = -1; dec[2] = -1; dec[3] = (t - CD(start_tick)) / CD(res); dec[4] = 0; dec[5] = 0; dec[6] = 0; dec[7] = 0; }
    for (int i = lane; i < CD(scope_cap); i += 64) {
      scope[2 * i] = -1; scope[2 * i + 1] = -1;
      if (K.obs) write_scope_observation_row(K, e, i, -1, t, [&](int, int) { return 0; });
    }
  }
  if (lane == 0) {
    met[0] = LW(LDS_HDR + CH_TRIPS); met[1] = LW(LDS_HDR + CH_SHORT); met[2] = LW(LDS_HDR + CH_OPER);
    *done = finished ? 1 : 0;
  }
  WP.mark(K, e, 6);
}
#undef LW
#endif  // MRX_CB_LDSFRAME

// Env.reset: citi_bike/business_engine.py:164-190, station.py:63-69
MRX_DEV void reset_env(const CbParams& K, int e) {
  for (int w = 0; w < CH_WORDS; w++) GHDR(w) = 0;
  GHDR(CH_TICK) = CD(start_tick);
  GHDR(CH_FLAGS) = CFL_FRESH;
  GHDR(CH_POOL_MINLAND) = CB_NO_LAND;
  for (int w = 0; w < CD(FW); w++) K.live[CB_IX(CD(aos), CD(stride), CD(FW), w, e)] = 0;
  for (int s = 0; s < CD(S); s++) { GST(LV_BIKES, s) = K.init_bikes[s]; GST(LV_MIN_BIKES, s) = K.init_bikes[s]; }
  for (int i = 0; i < CD(ring_slots); i++) { K.ring_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), i, e)] = -1; K.twc_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), i, e)] = -1; K.twc_tick[CB_IX(CD(aos), CD(stride), CD(ring_slots), i, e)] = 0; }
  for (int w = 0; w < 2 * CD(mask_words); w++) GDMK(w) = 0;
  for (int w = 0; w < CD(w_words); w++) GFUL(w) = 0;
  for (int i = 0; i < 2 * CB_LAND_SLOTS; i++) BKT(i) = -1;
  for (int w = 0; w < CB_LAND_SLOTS / 32; w++) BKT(BKT_MASK(w)) = 0;
}

// Observation fused into the step (mrx_cb_set_observation): what `mrx_cb_query("stations", decision frame, every station, attrs)`
// returns for the env's NEW decision — the decision's frame is the live frame (the pre-decision snapshot of core.py:345), so the
// values come straight out of the state the step holds (its LDS column in the LDS-frame build) and the agent's per-step snapshot
// slice costs no launch of its own.  Rows of an env without a valid decision (finished, or out of step budget) are zeros, like the
// query's padding for frame index -1.
MRX_DEV void write_observation(const CbParams& K, int e, const int32_t* dec) {
  double* o = K.obs + (size_t)e * CD(S) * K.obs_n;
  const bool valid = dec[5] != 0;
  const int t = dec[0];
  for (int a = 0; a < K.obs_n; a++) {
    const int attr = K.obs_attr[a];
    int lv = -1;
    switch (attr) {
      case SA_BIKES: lv = LV_BIKES; break;
      case SA_SHORTAGE: lv = LV_SHORTAGE; break;
      case SA_TRIP_REQUIREMENT: lv = LV_TRIP_REQUIREMENT; break;
      case SA_FULFILLMENT: lv = LV_FULFILLMENT; break;
      case SA_EXTRA_COST: lv = LV_EXTRA_COST; break;
      case SA_TRANSFER_COST: lv = LV_TRANSFER_COST; break;
      case SA_FAILED_RETURN: lv = LV_FAILED_RETURN; break;
      case SA_MIN_BIKES: lv = LV_MIN_BIKES; break;
      default: break;
    }
    for (int s = 0; s < CD(S); s++) {
      double v = 0.0;
      if (valid) {
        if (lv >= 0) v = (double)LIVE((size_t)lv * CD(S) + s);
        else if (attr == SA_CAPACITY) v = (double)K.capacity[s];
        else if (attr == SA_ID) v = (double)K.station_id[s];
        else v = (double)K.cal[(size_t)K.tick_day[t - CD(start_tick)] * 4 + (attr - SA_WEEKDAY)];
      }
      o[(size_t)s * K.obs_n + a] = v;
    }
  }
}

MRX_DEV int attr_slots(const CbParams& K, int node_type, int attr) {
  if (node_type == 0) return attr >= 0 && attr < SA_COUNT ? 1 : 0;
  return attr == CB_MA_TRIPS_ADJ ? CD(S) * CD(S) : 0;
}

// one element of snapshot_list[...]: row = (env, tick, node), col = flat (attr, slot)
MRX_DEV double query_elem(const CbParams& K, int node_type, const int32_t* ticks, int nt, int ticks_per_env, const int32_t* nodes, int nn,
                          int nodes_per_env, const int32_t* attrs, int na, long long row, int col) {
  const int ni = (int)(row % nn);
  const int ti = (int)((row / nn) % nt);
  const int e = (int)(row / ((long long)nn * nt));
  const int fi = ticks[(size_t)e * ticks_per_env + ti];
  if (fi < 0) return 0.0;
  const int slot = fi % CD(ring_slots);
  // while an env is paused at a decision its current frame is the live frame (core.py:345), which also evicts
  // whatever the slot held
  const int flags = GHDR(CH_FLAGS);
  const bool paused = (flags & CFL_PENDING) != 0;
  const int t_cur = GHDR(CH_TICK);
  const int cur_fi = (t_cur - CD(start_tick)) / CD(res);
  int from_ring;  // which frame holds fi: the live one (0) or ring slot `slot` (1)
  int t_frame;
  if (paused && fi == cur_fi) { from_ring = 0; t_frame = t_cur; }
  else if (paused && slot == cur_fi % CD(ring_slots)) return 0.0;
  else if (K.ring_fi[CB_IX(CD(aos), CD(stride), CD(ring_slots), slot, e)] == fi) {
    from_ring = 1;
    t_frame = RINGW(slot, CD(FW));
  } else return 0.0;  // padding for missing frames, np_backend.pyx:541-545
  int a = 0, sl = col;
  for (int i = 0; i < na; i++) {
    const int ns = attr_slots(K, node_type, attrs[i]);
    if (sl < ns) { a = attrs[i]; break; }
    sl -= ns;
  }
  const int node = nodes[(size_t)e * nodes_per_env + ni];
  if (node_type == 1) {
    // trips_adj of the frame taken at tick t_frame: the (src, dst) pair's trips up to and including that tick
    if (node != 0) return 0.0;
    const int bound = K.trip_off[t_frame - CD(start_tick) + 1];
    const int lo0 = K.adj_off[sl];
    int lo = lo0, hi = K.adj_off[sl + 1];
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (K.adj_idx[mid] < bound) lo = mid + 1; else hi = mid;
    }
    return (double)(lo - lo0);
  }
  if (node < 0 || node >= CD(S)) return 0.0;
  int lv = -1;
  switch (a) {
    case SA_BIKES: lv = LV_BIKES; break;
    case SA_SHORTAGE: lv = LV_SHORTAGE; break;
    case SA_TRIP_REQUIREMENT: lv = LV_TRIP_REQUIREMENT; break;
    case SA_FULFILLMENT: lv = LV_FULFILLMENT; break;
    case SA_EXTRA_COST: lv = LV_EXTRA_COST; break;
    case SA_TRANSFER_COST: lv = LV_TRANSFER_COST; break;
    case SA_FAILED_RETURN: lv = LV_FAILED_RETURN; break;
    case SA_MIN_BIKES: lv = LV_MIN_BIKES; break;
    case SA_CAPACITY: return (double)K.capacity[node];
    case SA_ID: return (double)K.station_id[node];
    default: return (double)K.cal[(size_t)K.tick_day[t_frame - CD(start_tick)] * 4 + (a - SA_WEEKDAY)];
  }
  const size_t fw = (size_t)lv * CD(S) + node;
  return (double)(from_ring ? RINGW(slot, fw) : LIVE(fw));
}

MRX_DEV uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// mrx_cb_random_policy for one env; returns 1 when the env had a valid decision
MRX_DEV int random_policy_env(const CbParams& K, int e, const int32_t* dec, const int32_t* scope, int64_t step, int32_t* action, int32_t* n_action) {
  *n_action = 0;
  if (!dec[5]) return 0;
  const int n = dec[4];
  if (n >= 2) {
    const int self = scope[2 * (n - 1)], self_max = scope[2 * (n - 1) + 1];
    const int other = scope[0], other_max = scope[1];
    int m = self_max < other_max ? self_max : other_max;
    if (m < 0) m = 0;
    const int number = (int)(mix64(mix64((uint64_t)step) ^ (uint64_t)e) % (uint64_t)(m + 1));
    action[0] = dec[2] == MRX_CB_SUPPLY ? self : other;
    action[1] = dec[2] == MRX_CB_SUPPLY ? other : self;
    action[2] = number;
    *n_action = 1;
  }
  return 1;
}

#undef HDR
#undef GHDR
#undef RINGW
#undef ST
#undef POOL
#undef BKT
#undef BKT_HEAD
#undef BKT_TAIL
#undef BKT_MASK
#undef SCR

}  // namespace cb
