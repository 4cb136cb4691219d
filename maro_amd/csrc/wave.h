// wave.h — wavefront-level collectives for gfx950 (CDNA4, wave64).
//
// The engine runs ONE environment per 64-lane wavefront (one wave per workgroup), so every
// cross-lane exchange is a wave collective: no s_barrier, no inter-workgroup traffic.
// All functions must be called from wave-uniform control flow.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MRX_DEV __device__ __forceinline__
#define MRX_WAVE 64

namespace wave {

MRX_DEV int lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// LDS visibility point inside one wave: orders this lane's LDS writes before other lanes'
// subsequent reads (s_waitcnt lgkmcnt(0)) and stops the compiler from moving LDS accesses
// across it.  A one-wave workgroup needs no s_barrier.
MRX_DEV void sync() {
  // LDS only: wait for this wave's outstanding LDS operations and stop the compiler from moving memory accesses
  // across this point.  Deliberately NOT a workgroup-scope fence: that would also drain vmcnt, i.e. stall on every
  // in-flight global load/store (prefetches, fire-and-forget stores) at each of the many syncs per tick.
#ifdef MRX_SYNC_NOWAIT
  // experiment: the LDS executes one wave's DS instructions in issue order, so a later ds_read of ANY lane sees an earlier
  // ds_write / ds_add of any other lane of the same wave without waiting for its acknowledgement; values that travel through
  // registers get their own s_waitcnt from the compiler.  Only the compiler barrier remains.
  asm volatile("" ::: "memory");
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  __builtin_amdgcn_wave_barrier();
}

MRX_DEV uint64_t ballot(bool pred) { return __ballot(pred); }

MRX_DEV int shfl(int v, int src) { return __shfl(v, src, 64); }
// value of lane `src`, src wave-uniform: v_readlane_b32 (a register read; no trip through the LDS crossbar)
MRX_DEV int readlane(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
// value of lane `src` for a WAVE-UNIFORM src (any lane for a lane-varying src is undefined behaviour): the cheap form of shfl
MRX_DEV int bcast(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src) & 63); }
MRX_DEV long long shfl(long long v, int src) {
  int lo = __shfl((int)(v & 0xffffffffll), src, 64), hi = __shfl((int)(v >> 32), src, 64);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// all-reduce (sum) over the 64 lanes; every lane gets the total.  Three 22-bit limbs, each summed by the DPP scan (64 x 2^22
// fits an int) and read from lane 63: no ds_bpermute.  |v| < 2^62.
MRX_DEV int scan_incl_add(int v);
MRX_DEV long long reduce_add(long long v) {
  const unsigned long long u = (unsigned long long)v;
  const int l0 = (int)(u & 0x3fffffu), l1 = (int)((u >> 22) & 0x3fffffu), l2 = (int)((long long)v >> 44);  // (top limb keeps the sign)
  const long long s0 = __builtin_amdgcn_readlane(scan_incl_add(l0), 63), s1 = __builtin_amdgcn_readlane(scan_incl_add(l1), 63),
                  s2 = __builtin_amdgcn_readlane(scan_incl_add(l2), 63);
  return s0 + (s1 << 22) + (s2 << 44);
}

// Direct HBM -> LDS copy (LDS-DMA, `global_load_lds_dwordx4`): every active lane moves 16 bytes from its
// own global address to lds_chunk + 16*lane, with no VGPR round trip; many can be in flight per wave.
MRX_DEV void lds_dma_16(int32_t* lds_chunk, const int32_t* gsrc_lane) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_chunk, 16, 0, 0);
}
MRX_DEV void lds_dma_wait() {
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
  sync();
}

// fire-and-forget LDS add (ds_add_u32): concurrent lanes may target the same word
MRX_DEV void lds_add(int32_t* p, int v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// global counter: returns the previous value (device scope)
MRX_DEV int global_add(int32_t* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// fire-and-forget OR into global memory (no return value, hence no s_waitcnt)
MRX_DEV void global_add_nr(int32_t* p, int v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // (result unused: no return path)
MRX_DEV void global_or(int32_t* p, int v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// inclusive prefix sum over the lanes (lane i gets v_0 + ... + v_i): six DPP adds — shifts inside the rows of 16 lanes, then
// the gfx9 row broadcasts (row_bcast:15 into rows 1 / 3, row_bcast:31 into rows 2 / 3).  No trip through the LDS crossbar
// (a __shfl_up scan is six dependent ds_bpermute round trips, ~10x the latency).  All 64 lanes must be active.
MRX_DEV int scan_incl_add(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

// "Use" a prefetched value here: forces the compiler's s_waitcnt for its load to this (straight-line) point, so
// that no conservative vmcnt(0) — which would also wait for every store issued since — is needed at later uses
// inside loops.
MRX_DEV void touch(int& x) { asm volatile("" : "+v"(x)); }
MRX_DEV void touch(uint32_t& x) { asm volatile("" : "+v"(x)); }
MRX_DEV void touch(double& x) { asm volatile("" : "+v"(x)); }

// make a wave-uniform value provably uniform (scalar register)
MRX_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// 16-byte register transfers of the persistent step kernel (state of the NEXT env prefetched into VGPRs while the
// current one is computed out of LDS): plain vector loads (vmcnt), non-temporal stores, ds_read/write_b128.
typedef int v4i __attribute__((ext_vector_type(4)));
MRX_DEV v4i ld16(const int32_t* p) { return *(const v4i*)p; }
MRX_DEV void st16_nt(int32_t* p, v4i v) { __builtin_nontemporal_store(v, (v4i*)p); }
// ... to a computed address: the cast keeps it a GLOBAL store (an address that went through integer arithmetic would
// otherwise become a flat_store, which also counts in lgkmcnt — the next wave::sync() would wait for it)
MRX_DEV void st16_nt_addr(uintptr_t a, v4i v) { __builtin_nontemporal_store(v, (__attribute__((address_space(1))) v4i*)a); }
MRX_DEV v4i lds_ld16(const int32_t* p) { return *(const v4i*)p; }
MRX_DEV void lds_st16(int32_t* p, v4i v) { *(v4i*)p = v; }
// hide a value's provenance (e.g. its wave-uniformity) from the compiler
MRX_DEV void opaque(int& x) { asm volatile("" : "+v"(x)); }

// wait until at most N vector-memory operations (loads AND stores: gfx950 has one in-order counter) are outstanding
template <int N>
MRX_DEV void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));  // vmcnt(N); expcnt / lgkmcnt: no wait
}

// a wave-uniform word through the VECTOR memory path (a plain global_load: vmcnt only).  A scalar load — and a
// flat_load — count in lgkmcnt, so the next wave::sync() would wait for them: they could not stay in flight across LDS
// phases.  The empty asm makes the (global-address-space) pointer opaque, so the compiler cannot scalarise the load.
MRX_DEV int ld_uniform_v(const int32_t* p) {
  typedef const __attribute__((address_space(1))) int32_t* gptr_t;
  gptr_t g = (gptr_t)p;
  asm volatile("" : "+v"(g));
  return *g;
}

}  // namespace wave
