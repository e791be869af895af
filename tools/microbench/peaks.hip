// peaks.hip -- what THIS box reaches on the two rooflines bench.py quotes ("quote measured peaks
// beside the spec figures"): a STREAM-style float4 copy for HBM, and a dependency-free v_mfma_f32_16x16x4_f32 loop for
// the FP32 matrix pipes.  Built as tools/microbench/libpeaks.so; bench.py calls the two functions through ctypes.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

static __global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// eight independent accumulators per wavefront: the pipe never waits for a dependent result
static __global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;  // keep the loop alive
}

extern "C" {
// GB/s of a device-to-device float4 copy (read + write counted), best of `reps`
double peaks_hbm_copy_gbs(size_t bytes, int reps) {
  float4 *a = nullptr, *b = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&a), bytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&b), bytes) != hipSuccess) return -1.0;
  (void)hipMemset(a, 1, bytes);
  (void)hipMemset(b, 0, bytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t n = bytes / sizeof(float4);
  double best = 0.0;
  for (int r = 0; r < reps + 1; ++r) {
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, nullptr, a, b, n);
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
  const int blocks = p.multiProcessorCount * 8;  // 8 workgroups x 4 wavefronts per CU
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
    const double flops = (double)blocks * 4 /* waves */ * iters * 8.0 * (2.0 * 16 * 16 * 4);
    const double tf = flops / (ms * 1e-3) / 1e12;
    if (r > 0 && tf > best) best = tf;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(out);
  return best;
}
}
