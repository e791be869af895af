// mfma_32x32.hip -- v_mfma_f32_32x32x2_f32 beside v_mfma_f32_16x16x4_f32 on gfx950: exactness against an fmaf chain, issue rate,
// and what one instruction's operand fetches cost next to it (the same question as mfma_mix.hip, per 4096 FLOP: one 32x32x2
// needs ONE A and ONE B register per lane where two 16x16x4 need two of each).
//   A: lane l -> A[row l % 32][k l / 32], B: lane l -> B[k l / 32][col l % 32], D: 16 registers, register r of lane l ->
//   D[row 8 (r / 4) + 4 (l / 32) + r % 4][col l % 32].
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_32x32 mfma_32x32.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void exact_kernel(const float* A, const float* B, float* D, int steps) {   // A[steps][64], B[steps][64] as the lanes see them
  const int l = threadIdx.x;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int s = 0; s < steps; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s * 64 + l], B[s * 64 + l], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[r * 64 + l] = acc[r];
}
template <int SHAPE /*32 or 16*/, int V, int L>
__global__ __launch_bounds__(256) void mix_kernel(float* out, int iters) {
  __shared__ float lds[4096];
  f32x16 big[2];
  f32x4 small[4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
  for (int i = 0; i < 4; ++i) small[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  lds[threadIdx.x] = a;
  __syncthreads();
  float l = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // 2 x 4096 FLOP per iteration in both shapes
      if (SHAPE == 32) big[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[i], 0, 0, 0);
      else { small[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, small[2 * i], 0, 0, 0); small[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, small[2 * i + 1], 0, 0, 0); }
      constexpr int N = SHAPE == 32 ? 1 : 2;   // operand fetches scale with the instruction count
#pragma unroll
      for (int j = 0; j < V * N; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
#pragma unroll
      for (int j = 0; j < L * N; ++j) l += lds[(threadIdx.x + 64 * (it & 7) + j) & 4095];
    }
  }
  float s = l;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += big[i][j];
  for (int i = 0; i < 4; ++i) s += small[i][0] + small[i][1] + small[i][2] + small[i][3];
  for (int j = 0; j < 8; ++j) s += v[j];
  if (s == 12345.678f) out[0] = s;
}
template <int SHAPE, int V, int L>
double rate(float* out) {
  const int iters = 4000;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 4;   // 4 wavefronts per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 0;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix_kernel<SHAPE, V, L>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)blocks * 4 * iters * 2 * 4096.0 / (ms * 1e-3) / 1e12;
    if (r > 0 && tf > best) best = tf;
  }
  return best;
}
int main() {
  const int steps = 128;   // K = 256
  std::vector<float> A(steps * 64), B(steps * 64), D(16 * 64), want(16 * 64);
  unsigned s = 777;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd() * 2.0f;
  for (int r = 0; r < 16; ++r)
    for (int l = 0; l < 64; ++l) {
      const int row = 8 * (r / 4) + 4 * (l / 32) + r % 4, col = l % 32;
      float acc = 0.f;
      for (int st = 0; st < steps; ++st)
        for (int kk = 0; kk < 2; ++kk) acc = fmaf(A[st * 64 + kk * 32 + row], B[st * 64 + kk * 32 + col], acc);
      want[r * 64 + l] = acc;
    }
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1 << 16);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(exact_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, steps);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (size_t i = 0; i < D.size(); ++i) bad += std::memcmp(&D[i], &want[i], 4) != 0;
  printf("32x32x2: layout + k-ascending fmaf chain (K = %d): %d of 1024 outputs differ\n", 2 * steps, bad);
  printf("4 wavefronts per SIMD, two independent accumulator sets; per 4096 FLOP: one 32x32x2 with v VALU + l LDS reads, or two 16x16x4 with 2v + 2l\n");
  printf("  v l   32x32x2 TFLOP/s   16x16x4 TFLOP/s\n");
  printf("  0 0   %15.1f   %15.1f\n", rate<32, 0, 0>(dD), rate<16, 0, 0>(dD));
  printf("  1 0   %15.1f   %15.1f\n", rate<32, 1, 0>(dD), rate<16, 1, 0>(dD));
  printf("  2 0   %15.1f   %15.1f\n", rate<32, 2, 0>(dD), rate<16, 2, 0>(dD));
  printf("  0 1   %15.1f   %15.1f\n", rate<32, 0, 1>(dD), rate<16, 0, 1>(dD));
  printf("  2 1   %15.1f   %15.1f\n", rate<32, 2, 1>(dD), rate<16, 2, 1>(dD));
  printf("  4 1   %15.1f   %15.1f\n", rate<32, 4, 1>(dD), rate<16, 4, 1>(dD));
  return bad != 0;
}
