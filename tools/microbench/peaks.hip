// peaks.hip -- what THIS box reaches on the two rooflines bench.py quotes ("quote measured peaks
// beside the spec figures"): a STREAM-style float4 copy for HBM, and a dependency-free v_mfma_f32_16x16x4_f32 loop for
// the FP32 matrix pipes.  Built as tools/microbench/libpeaks.so; bench.py calls the two functions through ctypes.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One float4 per thread, non-temporal load and store (the copy has no reuse), as many workgroups as it takes: 6.2 TB/s
// with plain accesses, 6.6 TB/s non-temporal on MI355X; grid-stride loops (a few thousand workgroups walking the buffer)
// reach 4.8-5.8 TB/s (round 2's kernel; measured again in round 3).
typedef float f4v __attribute__((ext_vector_type(4)));
static __global__ __launch_bounds__(256) void copy_kernel(const f4v* __restrict__ src, f4v* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// TWO independent accumulators per wavefront, four wavefronts per SIMD: the dependent latency (40 cycles) is covered, and
// -- tools/microbench/mfma_mix, profiles/r03_notes.md -- more accumulators per wavefront are SLOWER on this chip (four: 80 %,
// eight: 89 % of what two reach)
constexpr int kAcc = 2;
static __global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
  f32x4 acc[kAcc];
#pragma unroll
  for (int i = 0; i < kAcc; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < kAcc; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kAcc; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;  // keep the loop alive
}

extern "C" {
// GB/s of a device-to-device float4 copy (read + write counted), best of `reps`
double peaks_hbm_copy_gbs(size_t bytes, int reps) {
  f4v *a = nullptr, *b = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&a), bytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&b), bytes) != hipSuccess) return -1.0;
  (void)hipMemset(a, 1, bytes);
  (void)hipMemset(b, 0, bytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t n = bytes / sizeof(f4v);
  double best = 0.0;
  for (int r = 0; r < reps + 1; ++r) {
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, a, b, n);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double gbs = 2.0 * bytes / (ms * 1e-3) / 1e9;
    if (r > 0 && gbs > best) best = gbs;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  return best;
}
// TFLOP/s of independent v_mfma_f32_16x16x4_f32 (2 * 16 * 16 * 4 flop each) on every SIMD of the chip
double peaks_mfma_f32_tflops(int iters, int reps) {
  float* out = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&out), 64) != hipSuccess) return -1.0;
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 4;  // 4 workgroups x 4 wavefronts per CU: four wavefronts per SIMD
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double best = 0.0;
  for (int r = 0; r < reps + 1; ++r) {
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(256), 0, nullptr, out, iters);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 /* waves */ * iters * 4.0 * kAcc * (2.0 * 16 * 16 * 4);
    const double tf = flops / (ms * 1e-3) / 1e12;
    if (r > 0 && tf > best) best = tf;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(out);
  return best;
}
}
