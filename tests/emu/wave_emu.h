// wave_emu.h — TEST INFRASTRUCTURE ONLY (never part of libmaro_amd.so).
//
// A host-side emulation of maro_amd/csrc/wave.h: the 64 lanes of one wavefront run as 64
// ucontext fibers; every collective (sync / ballot / shfl / reduce) is a rendezvous.  It lets the
// *same* device source (maro_amd/csrc/cim_device.h) be compiled with g++ and checked against the
// CPU oracle in the `-m "not gpu"` suite, so kernel-logic bugs are caught before a GPU box is
// requested.  Lanes are run in forward or reverse order (EmuWave::reverse) — a missing
// wave::sync() between an LDS write and another lane's read shows up as a forward/reverse
// mismatch.  This file is only ever included by tests/emu/cim_emu.cpp.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>

#define MRX_DEV static inline
#define MRX_WAVE 64

namespace wave {

struct EmuWave {
  ucontext_t sched;
  ucontext_t ctx[64];
  char* stacks[64];
  bool done[64];
  int cur = 0;
  bool reverse = false;
  long long in[64];
  long long snap[64];
  int op[64];
  std::function<void()> body;
  long rounds = 0;
};

inline EmuWave*& cur_wave() {
  static thread_local EmuWave* w = nullptr;
  return w;
}

inline void trampoline() {
  EmuWave* w = cur_wave();
  w->body();
  w->done[w->cur] = true;
  swapcontext(&w->ctx[w->cur], &w->sched);
}

// Run `body` once per lane, SIMT style.
inline void run_wave(EmuWave& w, std::function<void()> body) {
  const size_t STK = 256 * 1024;
  w.body = body;
  cur_wave() = &w;
  for (int l = 0; l < 64; l++) {
    w.done[l] = false;
    w.in[l] = 0;
    w.op[l] = 0;
    if (!w.stacks[l]) w.stacks[l] = (char*)malloc(STK);
    getcontext(&w.ctx[l]);
    w.ctx[l].uc_stack.ss_sp = w.stacks[l];
    w.ctx[l].uc_stack.ss_size = STK;
    w.ctx[l].uc_link = &w.sched;
    makecontext(&w.ctx[l], (void (*)())trampoline, 0);
  }
  for (;;) {
    int alive = 0;
    for (int i = 0; i < 64; i++) {
      int l = w.reverse ? 63 - i : i;
      if (w.done[l]) continue;
      w.cur = l;
      w.op[l] = -1;
      swapcontext(&w.sched, &w.ctx[l]);
      if (!w.done[l]) alive++;
    }
    if (!alive) break;
    // rendezvous complete: every live lane must be parked in the same collective
    int opc = -2;
    for (int l = 0; l < 64; l++) {
      if (w.done[l]) { w.in[l] = 0; continue; }
      if (opc == -2) opc = w.op[l];
      if (w.op[l] != opc) {
        fprintf(stderr, "wave_emu: divergent collectives (lane %d op %d vs %d)\n", l, w.op[l], opc);
        abort();
      }
    }
    memcpy(w.snap, w.in, sizeof(w.in));
    w.rounds++;
  }
}

inline void free_wave(EmuWave& w) {
  for (int l = 0; l < 64; l++) { free(w.stacks[l]); w.stacks[l] = nullptr; }
}

inline int lane() { return cur_wave()->cur; }

inline void rendezvous(int opcode, long long v) {
  EmuWave* w = cur_wave();
  w->in[w->cur] = v;
  w->op[w->cur] = opcode;
  swapcontext(&w->ctx[w->cur], &w->sched);
}

inline int first_live_lane() { EmuWave* w = cur_wave(); for (int l = 0; l < 64; l++) if (!w->done[l]) return l; return 0; }

inline void sync() { rendezvous(1, 0); }

inline uint64_t ballot(bool pred) {
  rendezvous(2, pred ? 1 : 0);
  EmuWave* w = cur_wave();
  uint64_t m = 0;
  for (int l = 0; l < 64; l++) if (w->snap[l]) m |= (1ull << l);
  return m;
}

inline int shfl(int v, int src) { rendezvous(3, v); return (int)cur_wave()->snap[src & 63]; }
inline long long shfl(long long v, int src) { rendezvous(4, v); return cur_wave()->snap[src & 63]; }
inline int readlane(int v, int src) { return shfl(v, src); }
inline int bcast(int v, int src) { return shfl(v, src); }
inline void global_or(int32_t* p, int v) { *p |= v; }
inline int global_add(int32_t* p, int v) { const int o = *p; *p += v; return o; }
inline void global_add_nr(int32_t* p, int v) { *p += v; }

inline long long reduce_add(long long v) {
  rendezvous(5, v);
  EmuWave* w = cur_wave();
  long long s = 0;
  for (int l = 0; l < 64; l++) s += w->snap[l];
  return s;
}

inline void lds_add(int32_t* p, int v) { *p += v; }
inline void touch(int&) {}
inline void touch(uint32_t&) {}
inline void touch(double&) {}

inline int scan_incl_add(int v) {
  rendezvous(7, v);
  EmuWave* w = cur_wave();
  long long s = 0;
  for (int l = 0; l <= w->cur; l++) s += w->snap[l];
  return (int)s;
}

// readfirstlane: modelled as a rendezvous so that wave-uniform read-modify-write code (every lane reads,
// then every lane writes the same value) behaves as it does in lockstep on the GPU
inline int uniform(int v) { rendezvous(6, v); return (int)cur_wave()->snap[first_live_lane()]; }

struct v4i { int x, y, z, w; };
inline v4i ld16(const int32_t* p) { v4i v; memcpy(&v, p, 16); return v; }
inline void st16_nt(int32_t* p, v4i v) { memcpy(p, &v, 16); }
inline void st16_nt_addr(uintptr_t a, v4i v) { memcpy((void*)a, &v, 16); }
inline v4i lds_ld16(const int32_t* p) { return ld16(p); }
inline void lds_st16(int32_t* p, v4i v) { memcpy(p, &v, 16); }
inline int ld_uniform_v(const int32_t* p) { return *p; }
template <int N> inline void wait_vm() {}
inline void opaque(int&) {}

inline void lds_dma_16(int32_t* lds_chunk, const int32_t* gsrc_lane) { memcpy(lds_chunk + 4 * lane(), gsrc_lane, 16); }
inline void lds_dma_wait() { sync(); }

}  // namespace wave
