// tools/mfma_probe.hip — what one wave per SIMD can issue: exact-f32 MFMA forms in straight-line code (a measurement tool, not
// part of the product path).  hipcc -O3 --offload-arch=gfx950 -o tools/mfma_probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// FORM 0: 16x16x4, NA accumulators round-robin, the SAME operand registers for every MFMA
// FORM 1: 16x16x4, NA accumulators, 32 distinct A registers and 4 distinct B registers (as the forward's pass)
// FORM 2: 32x32x2, NA accumulators (16 registers each), distinct operands
// FORM 3: 16x16x4 as form 1 with one ds_read_b128 per 4 MFMAs feeding the B operands
template <int FORM, int NA>
__global__ void __launch_bounds__(256) k(int iters, long long* out, float* sink, const float* src) {
  __shared__ f4 lds[256];
  lds[threadIdx.x] = f4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  float av[32];
#pragma unroll
  for (int i = 0; i < 32; i++) av[i] = src[threadIdx.x + 64 * i];
  f4 bv = *(const f4*)(src + 4096 + threadIdx.x * 4);
  long long c0, c1, r0, r1;
  float s = 0.f;
  if (FORM == 2) {
    f16v acc[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    c0 = (long long)__builtin_amdgcn_s_memtime(); r0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int j = 0; j < 32; j++) acc[j % NA] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j & 3], acc[j % NA], 0, 0, 0);
    }
    c1 = (long long)__builtin_amdgcn_s_memtime(); r1 = (long long)__builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < NA; i++) s += acc[i][0] + acc[i][15];
  } else {
    f4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    c0 = (long long)__builtin_amdgcn_s_memtime(); r0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
      if (FORM == 3) {
#pragma unroll
        for (int kb = 0; kb < 16; kb++) {
          const f4 b = lds[(threadIdx.x + kb * 16 + it) & 255];
#pragma unroll
          for (int j = 0; j < 4; j++) acc[(kb * 4 + j) % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(kb * 4 + j) & 31], b[j], acc[(kb * 4 + j) % NA], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 64; j++)
          acc[j % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(FORM == 0 ? av[0] : av[j & 31], FORM == 0 ? bv[0] : bv[j & 3], acc[j % NA], 0, 0, 0);
      }
    }
    c1 = (long long)__builtin_amdgcn_s_memtime(); r1 = (long long)__builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < NA; i++) s += acc[i][0] + acc[i][3];
  }
  if (s == 12345.678f) sink[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

// straight-line code executed for the first time: N MFMAs unrolled (8 bytes each), the same sequence run REPS times by every wave;
// out[rep] = cycles of pass rep (rep 0 fetches the instructions from memory, the later ones find them in the instruction cache)
template <int N>
__global__ void __launch_bounds__(256) k_cold(long long* out, float* sink, const float* src) {
  float av[32];
#pragma unroll
  for (int i = 0; i < 32; i++) av[i] = src[threadIdx.x + 64 * i];
  f4 bv = *(const f4*)(src + 4096 + threadIdx.x * 4);
  f4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  long long t[5];
  t[0] = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int rep = 0; rep < 4; rep++) {
#pragma unroll
    for (int j = 0; j < N; j++) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j & 31], bv[(j >> 2) & 3], acc[j & 3], 0, 0, 0);
    t[rep + 1] = (long long)__builtin_amdgcn_s_memtime();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) for (int r = 0; r < 4; r++) atomicAdd((unsigned long long*)&out[r], (unsigned long long)(t[r + 1] - t[r]));
}

template <int N>
int run_cold(int wgs, long long* out, float* sink, const float* src, hipStream_t st) {
  for (int launch = 0; launch < 2; launch++) {
    CHECK(hipMemsetAsync(out, 0, 64, st));
    k_cold<N><<<wgs, 256, 0, st>>>(out, sink, src);
    CHECK(hipStreamSynchronize(st));
    long long h[4]; CHECK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"mfma_cold_code\", \"mfmas\": %d, \"code_kb\": %.1f, \"wgs\": %d, \"launch\": %d, \"cycles_per_mfma_by_pass\": [%.1f, %.1f, %.1f, %.1f]}\n", N, N * 8 / 1024.0, wgs, launch,
           (double)h[0] / wgs / N, (double)h[1] / wgs / N, (double)h[2] / wgs / N, (double)h[3] / wgs / N);
  }
  return 0;
}

template <int FORM, int NA>
int run(int wgs, long long* out, float* sink, const float* src, hipStream_t st) {
  const int iters = 2000, per = FORM == 2 ? 32 : 64;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; rep++) {
    CHECK(hipEventRecord(e0, st));
    k<FORM, NA><<<wgs, 256, 0, st>>>(iters, out, sink, src);
    CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
  }
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  long long h[2]; CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
  const double n = (double)iters * per, fl = FORM == 2 ? 4096.0 : 2048.0;
  printf("{\"probe\": \"mfma_issue\", \"form\": %d, \"accumulators\": %d, \"wgs\": %d, \"cycles_per_mfma_per_wave\": %.2f, \"cycles_per_2048_flop\": %.2f, \"tflops_wall\": %.1f}\n", FORM, NA, wgs,
         h[0] / n, h[0] / n * 2048.0 / fl, n * fl * wgs * 4 / (ms * 1e9));
  return 0;
}

int main() {
  hipStream_t st; CHECK(hipStreamCreate(&st));
  long long* out; float *sink, *src;
  CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&src, 65536)); CHECK(hipMemset(src, 0, 65536));
  run_cold<64>(256, out, sink, src, st); run_cold<256>(256, out, sink, src, st); run_cold<1024>(256, out, sink, src, st); run_cold<1024>(1, out, sink, src, st);
  for (int wgs : {256}) {
    run<0, 1>(wgs, out, sink, src, st); run<0, 2>(wgs, out, sink, src, st); run<0, 4>(wgs, out, sink, src, st); run<0, 8>(wgs, out, sink, src, st);
    run<1, 1>(wgs, out, sink, src, st); run<1, 2>(wgs, out, sink, src, st); run<1, 4>(wgs, out, sink, src, st); run<1, 8>(wgs, out, sink, src, st);
    run<2, 1>(wgs, out, sink, src, st); run<2, 2>(wgs, out, sink, src, st); run<2, 4>(wgs, out, sink, src, st);
    run<3, 1>(wgs, out, sink, src, st); run<3, 4>(wgs, out, sink, src, st);
  }
  return 0;
}
