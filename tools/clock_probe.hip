// clock_probe.hip — what the chip actually delivers under this repo's launch patterns (round 6): the shader clock during
// short back-to-back launches vs one long launch (s_memtime = shader cycles, s_memrealtime = 100 MHz), exact-f32 MFMA issue
// rate, dependent-load latency (L2 / HBM) and the weight-streaming rate of the DQN forward's access pattern (every workgroup
// of a port reads the same 418 KB with 16-byte loads).  A measurement tool, not part of the product path.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/clock_probe tools/clock_probe.hip ; output: JSON lines.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <vector>

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// out[launch * 4 + {0,1,2,3}] = shader cycles, 100 MHz ticks, start realtime (low 32), unused — written by block 0 / thread 0
__global__ void __launch_bounds__(256) k_spin(int iters, long long* out, int slot, float* sink) {
  const long long c0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
  float a = (float)threadIdx.x, b = 1.0001f;
  for (int i = 0; i < iters; i++) a = __builtin_fmaf(a, b, 0.5f);
  const long long c1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
  if (a == 12345.678f) sink[0] = a;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[slot * 4 + 0] = c1 - c0;
    out[slot * 4 + 1] = r1 - r0;
    out[slot * 4 + 2] = r0;
  }
}

// exact-f32 MFMA issue rate: NA independent accumulators per wave, one wave per SIMD when launched with 256-thread blocks
template <int NA>
__global__ void __launch_bounds__(256) k_mfma(int iters, long long* out, float* sink) {
  f4 acc[NA];
#pragma unroll
  for (int i = 0; i < NA; i++) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  const float a = (float)(threadIdx.x & 7), b = 1.0f / (float)(1 + (threadIdx.x & 3));
  const long long c0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < NA; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  const long long c1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NA; i++) s += acc[i][0] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

__global__ void k_chase(const int* __restrict__ next, int hops, long long* out, int* sink) {
  int p = 0;
  for (int i = 0; i < 64; i++) p = next[p];   // warm the TLB path a little
  const long long c0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < hops; i++) p = next[p];
  const long long c1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
  sink[0] = p;
  out[0] = c1 - c0; out[1] = r1 - r0;
}

// the DQN forward's weight stream: workgroup b reads region (b / per_region) of `floats` floats with DEPTH 16-byte loads
// in flight per lane, each wave its own quarter of every 4 KB (as the packed layout deals column tiles to waves)
template <int DEPTH, int BS = 256>
__global__ void __launch_bounds__(BS) k_stream(const float* __restrict__ w, int floats, int per_region, int regions, long long* out, float* sink) {
  const int region = (blockIdx.x / per_region) % regions;
  const f4* base = (const f4*)(w + (size_t)region * floats);
  const int n16 = floats >> 2, t = threadIdx.x;
  f4 s = f4{0.f, 0.f, 0.f, 0.f};
  const long long c0 = (long long)__builtin_amdgcn_s_memtime();
  for (int i = t; i + (DEPTH - 1) * BS < n16; i += DEPTH * BS) {
    f4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) v[d] = base[i + d * BS];
    __builtin_amdgcn_sched_barrier(0);   // every load of the round is issued before the first use (hipcc otherwise waits after two or three)
#pragma unroll
    for (int d = 0; d < DEPTH; d++) s += v[d];
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long c1 = (long long)__builtin_amdgcn_s_memtime();
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[0] = s[0];
  if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
}

// the same stream through LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction lands in LDS, no registers), DEPTH pieces
// per wave in flight, one wait per round
template <int DEPTH>
__global__ void __launch_bounds__(256) k_stream_lds(const float* __restrict__ w, int floats, int per_region, int regions, long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) float buf[DEPTH * 1024];   // DEPTH x 4 KB
  const int region = (blockIdx.x / per_region) % regions;
  const float* base = w + (size_t)region * floats;
  const int n16 = floats >> 2, t = threadIdx.x, wv = t >> 6;
  const long long c0 = (long long)__builtin_amdgcn_s_memtime();
  float acc = 0.f;
  for (int i = 0; i + DEPTH * 256 <= n16; i += DEPTH * 256) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)(i + d * 256 + t) * 4),
                                       (__attribute__((address_space(3))) void*)(buf + (d * 256 + wv * 64) * 4), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    acc += buf[(t * 17) & (DEPTH * 1024 - 1)];
  }
  const long long c1 = (long long)__builtin_amdgcn_s_memtime();
  if (acc == 12345.678f) sink[0] = acc;
  if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  long long* out; float* sink;
  CHECK(hipMalloc(&out, 1 << 20)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(out, 0, 1 << 20));
  std::vector<long long> h(131072);
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));

  // ---- 1. shader clock: one long launch, then 2000 short launches back to back, then short launches with a host sync between
  {
    k_spin<<<1024, 256, 0, st>>>(1000, out, 0, sink); CHECK(hipStreamSynchronize(st));
    k_spin<<<1024, 256, 0, st>>>(4000000, out, 0, sink); CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(h.data(), out, 32, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"clock_long\", \"cycles\": %lld, \"us\": %.1f, \"ghz\": %.3f}\n", h[0], h[1] / 100.0, h[0] / (h[1] * 10.0));
    for (int form = 0; form < 3; form++) {   // 0: back to back ~50 us, 1: host sync after each, 2: back to back ~10 us
      const int n = 2000, iters = form == 2 ? 2500 : 12000, wgs = 1024;
      const double t0 = now_ms();
      for (int i = 0; i < n; i++) {
        k_spin<<<wgs, 256, 0, st>>>(iters, out, i, sink);
        if (form == 1) CHECK(hipStreamSynchronize(st));
      }
      CHECK(hipStreamSynchronize(st));
      const double t1 = now_ms();
      CHECK(hipMemcpy(h.data(), out, n * 32, hipMemcpyDeviceToHost));
      double g[4] = {0, 0, 0, 0};
      for (int q = 0; q < 4; q++) {
        long long c = 0, r = 0;
        for (int i = q * n / 4; i < (q + 1) * n / 4; i++) { c += h[i * 4]; r += h[i * 4 + 1]; }
        g[q] = c / (r * 10.0);
      }
      printf("{\"probe\": \"clock_short\", \"form\": %d, \"launches\": %d, \"wall_us_per_launch\": %.2f, \"kernel_us\": %.2f, \"ghz_quarters\": [%.3f, %.3f, %.3f, %.3f]}\n",
             form, n, (t1 - t0) * 1000.0 / n, h[(n - 1) * 4 + 1] / 100.0, g[0], g[1], g[2], g[3]);
    }
  }
  // ---- 1b. the same at lower occupancies (the latency-bound kernels of this repo keep one to three waves per SIMD busy a third of
  //          the time): does the power management clock a lightly loaded chip lower?  2000 back-to-back launches of ~40 us each.
  {
    const int cfgs[][2] = {{1, 64}, {256, 64}, {1024, 64}, {4096, 64}, {256, 256}, {16384, 64}};
    for (auto& c : cfgs) {
      const int n = 2000;
      const double t0 = now_ms();
      for (int i = 0; i < n; i++) k_spin<<<c[0], c[1], 0, st>>>(12000, out, i, sink);
      CHECK(hipStreamSynchronize(st));
      const double t1 = now_ms();
      CHECK(hipMemcpy(h.data(), out, n * 32, hipMemcpyDeviceToHost));
      long long cs = 0, rs = 0;
      for (int i = n / 2; i < n; i++) { cs += h[i * 4]; rs += h[i * 4 + 1]; }
      printf("{\"probe\": \"clock_occupancy\", \"grid\": %d, \"block\": %d, \"wall_us_per_launch\": %.2f, \"kernel_us\": %.2f, \"ghz_second_half\": %.3f}\n",
             c[0], c[1], (t1 - t0) * 1000.0 / n, h[(n - 1) * 4 + 1] / 100.0, cs / (rs * 10.0));
    }
  }
  // ---- 2. exact-f32 MFMA rate: one wave per SIMD (256 WGs x 256), 4 and 8 independent accumulators; and 2 waves per SIMD
  for (int cfg = 0; cfg < 3; cfg++) {
    const int iters = 20000, wgs = cfg == 2 ? 512 : 256, na = cfg == 1 ? 8 : 4;
    for (int rep = 0; rep < 2; rep++) {
      CHECK(hipEventRecord(e0, st));
      if (na == 8) k_mfma<8><<<wgs, 256, 0, st>>>(iters, out, sink); else k_mfma<4><<<wgs, 256, 0, st>>>(iters, out, sink);
      CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(h.data(), out, 32, hipMemcpyDeviceToHost));
    const double n_mfma = (double)iters * na, flops = n_mfma * 2048.0 * wgs * 4;
    printf("{\"probe\": \"mfma_f32_16x16x4\", \"wgs\": %d, \"accumulators\": %d, \"cycles_per_mfma_per_wave\": %.2f, \"ghz\": %.3f, \"tflops_wall\": %.1f}\n",
           wgs, na, h[0] / n_mfma, h[0] / (h[1] * 10.0), flops / (ms * 1e9));
  }
  // ---- 3. dependent-load latency: one lane, 64-byte-strided random cycle through 2 MB (L2) and 1 GB (HBM)
  for (int big = 0; big < 2; big++) {
    const size_t bytes = big ? (size_t)1 << 30 : (size_t)2 << 20, n = bytes / 64;
    std::vector<int> perm(n), nxt(bytes / 4, 0);
    for (size_t i = 0; i < n; i++) perm[i] = (int)i;
    uint64_t x = 88172645463325252ull;
    for (size_t i = n - 1; i > 0; i--) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::swap(perm[i], perm[x % (i + 1)]); }
    for (size_t i = 0; i < n; i++) nxt[(size_t)perm[i] * 16] = perm[(i + 1) % n] * 16;
    int* d; CHECK(hipMalloc(&d, bytes)); CHECK(hipMemcpy(d, nxt.data(), bytes, hipMemcpyHostToDevice));
    const int hops = big ? 20000 : 30000;
    for (int rep = 0; rep < (big ? 1 : 3); rep++) { k_chase<<<1, 1, 0, st>>>(d, hops, out, (int*)sink); CHECK(hipStreamSynchronize(st)); }
    CHECK(hipMemcpy(h.data(), out, 32, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"chase\", \"bytes\": %zu, \"cycles_per_hop\": %.1f, \"ns_per_hop\": %.1f}\n", bytes, (double)h[0] / hops, h[1] * 10.0 / hops);
    CHECK(hipFree(d));
  }
  // ---- 4. the forward's weight stream: 22 regions of 104448 floats, 288 / 576 workgroups, 4 / 16 loads in flight per lane
  {
    const int floats = 104448, regions = 22;
    float* w; CHECK(hipMalloc(&w, (size_t)floats * regions * 4)); CHECK(hipMemset(w, 0, (size_t)floats * regions * 4));
    float* wbig; CHECK(hipMalloc(&wbig, (size_t)floats * 576 * 4)); CHECK(hipMemset(wbig, 0, (size_t)floats * 576 * 4));
    // cfg: bit0 576 instead of 288 workgroups; form 0: 22 shared regions, depth 4; 1: depth 16; 2: ONE region shared by all; 3: a region per
    // workgroup (no sharing, 120 / 240 MB footprint); 4: 512-thread workgroups, depth 8; 5: 1024-thread, depth 4; 6: LDS-DMA depth 8; 7: LDS-DMA depth 16
    for (int cfg = 0; cfg < 16; cfg++) {
      const int wgs = cfg & 1 ? 576 : 288, form = cfg >> 1, depth = form == 0 ? 4 : form == 4 ? 8 : form == 5 ? 4 : form == 6 ? 8 : 16;
      const int regs = form == 2 ? 1 : form == 3 ? wgs : regions, per_region = form == 3 ? 1 : wgs / regs + (wgs % regs ? 1 : 0);
      const float* src = form == 3 ? wbig : w;
      float ms = 0;
      for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, st));
        if (form == 0) k_stream<4><<<wgs, 256, 0, st>>>(src, floats, per_region, regs, out, sink);
        else if (form == 4) k_stream<8, 512><<<wgs, 512, 0, st>>>(src, floats, per_region, regs, out, sink);
        else if (form == 5) k_stream<4, 1024><<<wgs, 1024, 0, st>>>(src, floats, per_region, regs, out, sink);
        else if (form == 6) k_stream_lds<8><<<wgs, 256, 0, st>>>(src, floats, per_region, regs, out, sink);
        else if (form == 7) k_stream_lds<16><<<wgs, 256, 0, st>>>(src, floats, per_region, regs, out, sink);
        else k_stream<16><<<wgs, 256, 0, st>>>(src, floats, per_region, regs, out, sink);
        CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
      }
      CHECK(hipMemcpy(h.data(), out, wgs * 8, hipMemcpyDeviceToHost));
      double mean = 0; long long mx = 0;
      for (int i = 0; i < wgs; i++) { mean += h[i]; if (h[i] > mx) mx = h[i]; }
      mean /= wgs;
      printf("{\"probe\": \"weight_stream\", \"form\": %d, \"wgs\": %d, \"loads_in_flight\": %d, \"kernel_us\": %.1f, \"cycles_mean\": %.0f, \"cycles_max\": %lld, \"bytes_per_cycle_per_wg\": %.1f, \"aggregate_TBps\": %.2f}\n",
             form, wgs, depth, ms * 1000.0, mean, mx, floats * 4.0 / mean, (double)wgs * floats * 4.0 / (ms * 1e9));
    }
    CHECK(hipFree(w)); CHECK(hipFree(wbig));
  }
  return 0;
}
