// mfma_mix.hip -- calibration of the FP32 matrix pipe of one SIMD under the conditions of the tick launch's bodies:
// W wavefronts per SIMD, A independent accumulators per wavefront, V dependent-free VALU instructions and L LDS reads per
// MFMA interleaved (is VALU / LDS issue free beside a saturated MFMA pipe?).  Prints TFLOP/s of v_mfma_f32_16x16x4_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int A, int V, int L, int G = 0, int GS = 0>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* __restrict__ gsrc = nullptr, float* __restrict__ gdst = nullptr) {
  __shared__ float lds[4096];
  f32x4 acc[A];
#pragma unroll
  for (int i = 0; i < A; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  lds[threadIdx.x] = a;
  __syncthreads();
  float l = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < A; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < V; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
#pragma unroll
      for (int j = 0; j < L; ++j) l += lds[(threadIdx.x + 64 * (it & 7) + j) & 4095];
#pragma unroll
      for (int j = 0; j < G; ++j) l += gsrc[((blockIdx.x & 63) * 4096 + threadIdx.x + 256 * ((it + j) & 7))];      // 4-byte loads, L1 / L2 hits
#pragma unroll
      for (int j = 0; j < GS; ++j) gdst[(size_t)blockIdx.x * 4096 + threadIdx.x + 256 * ((it + j) & 7)] = l;      // 4-byte stores
    }
  }
  float s = l;
#pragma unroll
  for (int i = 0; i < A; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j];
  if (s == 12345.678f) out[0] = s;
}
template <int A, int V, int L, int G = 0, int GS = 0>
void run(int waves_per_simd, float* out, const float* gsrc = nullptr, float* gdst = nullptr) {
  const int iters = 4000 / A * 4;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * waves_per_simd;  // 256 threads = 4 wavefronts = one per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 0;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<A, V, L, G, GS>), dim3(blocks), dim3(256), 0, 0, out, iters, gsrc, gdst);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)blocks * 4 * iters * A * 2048.0 / (ms * 1e-3) / 1e12;
    if (r > 0 && tf > best) best = tf;
  }
  printf("waves/SIMD %d  accumulators %d  VALU/MFMA %d  LDS reads/MFMA %d  global 4B loads/MFMA %d stores/MFMA %d : %6.1f TFLOP/s (%.0f %% of 157.3; %.1f cycles per MFMA at 2.4 GHz)\n", waves_per_simd, A, V, L, G, GS, best, best / 1.573, 32.0 * 157.3 / best);
}
int main() {
  float* out; hipMalloc(&out, 64);
  for (int w : {1, 2, 4}) { run<1, 0, 0>(w, out); run<2, 0, 0>(w, out); run<4, 0, 0>(w, out); }
  float *gsrc, *gdst; hipMalloc(&gsrc, 64 * 4096 * 4); hipMemset(gsrc, 0, 64 * 4096 * 4); hipMalloc(&gdst, (size_t)8192 * 4096 * 4);
  for (int w : {4}) { run<2, 0, 0, 1, 0>(w, out, gsrc, gdst); run<2, 0, 0, 0, 1>(w, out, gsrc, gdst); run<2, 0, 0, 1, 1>(w, out, gsrc, gdst); run<2, 1, 0>(w, out); run<2, 0, 2>(w, out); }
  for (int w : {2, 4}) { run<2, 2, 0>(w, out); run<2, 4, 0>(w, out); run<2, 8, 0>(w, out); run<2, 0, 1>(w, out); run<2, 4, 1>(w, out); run<1, 4, 1>(w, out); }
  return 0;
}
