// cim_prof.h — tools only (-DMRX_PROFILE_PHASES): wave cycles (s_memtime) attributed to the phases of the CIM step kernel,
// accumulated in a device global of the translation unit that includes it (cim_engine.hip: libmaro_amd_prof.so;
// cim_spec.hip: a specialised code object built with MARO_AMD_SPEC_FLAGS=-DMRX_PROFILE_PHASES, read back with
// mrx_cim_read_kernel_global).  Without the macro this header is empty and cim_device.h supplies a no-op Prof.
#pragma once
#ifdef MRX_PROFILE_PHASES
__device__ unsigned long long g_mrx_prof[16];
namespace cim {
struct Prof {
  long long last, acc[16];
  __device__ __forceinline__ Prof() { for (int i = 0; i < 16; i++) acc[i] = 0; last = clock64(); }
  __device__ __forceinline__ void mark(int i) { long long c = clock64(); acc[i] += c - last; last = c; }
  __device__ __forceinline__ void mark(int i, long long add) { acc[i] += add; }
  __device__ __forceinline__ void flush() {
    // one workgroup in 32 reports (scaled back by the reader): the atomics of every wave perturbed the very thing measured
    if (wave::lane() == 0 && (blockIdx.x & 31) == 0) for (int i = 0; i < 16; i++) if (acc[i]) atomicAdd(&g_mrx_prof[i], (unsigned long long)acc[i] * 32ull);
  }
};
}  // namespace cim
#endif
