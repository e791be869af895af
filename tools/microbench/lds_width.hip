// lds_width.hip -- issue cost of the A-operand LDS reads beside v_mfma_f32_16x16x4_f32: four ds_read_b32, two ds_read2_b32
// (what the GEMM bodies issue per k-block today) or one ds_read_b128 (a lane's four k values stored contiguously) per 4 MFMAs.
// build: hipcc -O3 --offload-arch=gfx950 -o lds_width lds_width.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int CG>   // MODE 0: no LDS; 1: 4 x b32 (stride 4 floats); 2: 1 x b128; CG column tiles share the A values
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 260 * 2];
  for (int i = threadIdx.x; i < 16 * 260 * 2; i += 256) lds[i] = 1.0f + i * 1e-6f;
  __syncthreads();
  f32x4 acc[CG];
  for (int c = 0; c < CG; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  const float b = 1.0f - lane * 1e-6f;
  const float* a1 = lds + (lane & 15) * 258 + (lane >> 4);          // row l & 15, k = 16 kb + 4 i + (l >> 4)
  const float* a4 = lds + (lane & 15) * 260 + (lane >> 4) * 4;      // row l & 15, the lane's four k values contiguous
  for (int it = 0; it < iters; ++it) {
    const int kb = it & 15;
    float av[4] = {1.f, 1.f, 1.f, 1.f};
    if (MODE == 1) { const float* p = a1 + kb * 16; av[0] = p[0]; av[1] = p[4]; av[2] = p[8]; av[3] = p[12]; }
    if (MODE == 2) { const float4 v = *reinterpret_cast<const float4*>(a4 + kb * 16); av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w; }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CG; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.678f) out[0] = s;
}
template <int MODE, int CG>
double rate(float* out) {
  const int iters = 8000;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 0;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, CG>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)blocks * 4 * iters * 4 * CG * 2048.0 / (ms * 1e-3) / 1e12;
    if (r > 0 && tf > best) best = tf;
  }
  return best;
}
int main() {
  float* out; hipMalloc(&out, 64);
  printf("4 wavefronts per SIMD; A operand of a k-block (4 MFMAs x CG column tiles) from LDS: none / 4 x ds_read_b32 (compiler: 2 x ds_read2_b32) / 1 x ds_read_b128\n");
  printf("  CG = 1: %6.1f / %6.1f / %6.1f TFLOP/s\n", rate<0, 1>(out), rate<1, 1>(out), rate<2, 1>(out));
  printf("  CG = 2: %6.1f / %6.1f / %6.1f TFLOP/s\n", rate<0, 2>(out), rate<1, 2>(out), rate<2, 2>(out));
  return 0;
}
