// pk_mix.hip -- what does a PACKED float32 VALU instruction (v_pk_fma_f32: two results per lane) cost beside a saturated FP32
// matrix pipe, against one and two scalar v_fma_f32?  (Same frame as mfma_mix.hip: 4 wavefronts per SIMD, two accumulators,
// V instructions of the kind after every MFMA.)  Also: v_med3_f32, v_rcp_f32 (a quarter-rate transcendental), v_cndmask.
// build: hipcc -O3 --offload-arch=gfx950 -o pk_mix pk_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  f32x2 p[8];
  for (int j = 0; j < 8; ++j) p[j] = f32x2{1.0f + j, 2.0f + j};
  const f32x2 c1 = {1.0001f, 1.0002f}, c2 = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        if (KIND == 0) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
        if (KIND == 1) p[j & 7] = __builtin_elementwise_fma(p[j & 7], c1, c2);
        if (KIND == 2) v[j & 7] = __builtin_amdgcn_fmed3f(v[j & 7], -86.0f, 88.0f) + 1.0f;   // (2 instructions)
        if (KIND == 3) v[j & 7] = __builtin_amdgcn_rcpf(v[j & 7]);
        if (KIND == 4) asm volatile("v_mov_b32 %0, %0" : "+v"(v[j & 7]));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int j = 0; j < 8; ++j) s += v[j] + p[j].x + p[j].y;
  if (s == 12345.678f) out[0] = s;
}
template <int KIND, int V>
void run(const char* what, float* out) {
  const int iters = 8000;
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int blocks = pr.multiProcessorCount * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 0;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, V>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)blocks * 4 * iters * 2 * 2048.0 / (ms * 1e-3) / 1e12;
    if (r > 0 && tf > best) best = tf;
  }
  printf("%-28s x %d per MFMA: %6.1f TFLOP/s, %.1f cycles per MFMA\n", what, V, best, 32.0 * 157.3 / best);
}
int main() {
  float* out; hipMalloc(&out, 64);
  run<0, 0>("nothing", out);
  run<0, 1>("v_fma_f32", out); run<0, 2>("v_fma_f32", out); run<0, 4>("v_fma_f32", out);
  run<1, 1>("v_pk_fma_f32", out); run<1, 2>("v_pk_fma_f32", out); run<1, 4>("v_pk_fma_f32", out);
  run<2, 1>("v_med3_f32 + v_add_f32", out); run<2, 2>("v_med3_f32 + v_add_f32", out);
  run<3, 1>("v_rcp_f32", out); run<3, 2>("v_rcp_f32", out);
  run<4, 2>("v_mov_b32", out); run<4, 4>("v_mov_b32", out);
  return 0;
}
