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
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

MRX_DEV uint64_t ballot(bool pred) { return __ballot(pred); }

MRX_DEV int shfl(int v, int src) { return __shfl(v, src, 64); }
// value of lane `src`, src wave-uniform: v_readlane_b32 (a register read; no trip through the LDS crossbar)
MRX_DEV int readlane(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
MRX_DEV long long shfl(long long v, int src) {
  int lo = __shfl((int)(v & 0xffffffffll), src, 64), hi = __shfl((int)(v >> 32), src, 64);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// butterfly all-reduce (sum) over the 64 lanes; every lane gets the total
MRX_DEV long long reduce_add(long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    int lo = __shfl_xor((int)(v & 0xffffffffll), m, 64), hi = __shfl_xor((int)(v >> 32), m, 64);
    v += ((long long)hi << 32) | (unsigned int)lo;
  }
  return v;
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

// fire-and-forget OR into global memory (no return value, hence no s_waitcnt)
MRX_DEV void global_or(int32_t* p, int v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// inclusive prefix sum over the lanes (lane i gets v_0 + ... + v_i)
MRX_DEV int scan_incl_add(int v) {
  const int l = lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int u = __shfl_up(v, d, 64);
    if (l >= d) v += u;
  }
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
