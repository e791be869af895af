// mfma_4x4.hip -- v_mfma_f32_4x4x1_16b_f32 on gfx950: operand layout, exactness against an fmaf chain, issue rate.
// 16 independent blocks per instruction, each D_b[4][4] += A_b[4][1] . B_b[1][4].  Expected layout (checked below):
//   A: lane 4 b + i holds A_b[i];  B: lane 4 b + j holds B_b[j];  D: register r of lane 4 b + j holds D_b[r][j].
// build: hipcc -O2 --offload-arch=gfx950 -o mfma_4x4 mfma_4x4.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float* A, const float* B, float* D, int K) {  // A[K][64], B[K][64] as the lanes see them
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[k * 64 + l], B[k * 64 + l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[r * 64 + l] = acc[r];
}
template <int ACC>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, long long* cyc) {
  f32x4 acc[ACC];
  for (int c = 0; c < ACC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < ACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
  const long long t1 = clock64();
  float s = 0.f;
  for (int c = 0; c < ACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const int K = 256;
  std::vector<float> A(K * 64), B(K * 64), D(256), want(256, 0.f);
  unsigned s = 12345;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd() * 3.0f;
  for (int b = 0; b < 16; ++b)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(A[k * 64 + 4 * b + i], B[k * 64 + 4 * b + j], acc);
        want[i * 64 + 4 * b + j] = acc;  // register i, lane 4 b + j
      }
  float *dA, *dB, *dD;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 1 << 22));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
  CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += std::memcmp(&D[i], &want[i], 4) != 0;
  printf("layout + fmaf-chain exactness (K = %d): %d of 256 outputs differ\n", K, bad);
  long long* dc; CK(hipMalloc(&dc, 8));
  auto rate = [&](auto kern, int acc, int waves_per_simd) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;   // 256 threads = 4 wavefronts = one per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dD, 10, dc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dD, iters, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * acc;
    printf("acc %d, %d waves/SIMD: %.1f shader cycles per MFMA per wave, %.1f TFLOP/s (512 flop each)\n", acc, waves_per_simd, cyc / n,
           n * blocks * 4 * 512.0 / (ms * 1e-3) / 1e12);
  };
  rate(rate_kernel<1>, 1, 1); rate(rate_kernel<2>, 2, 1); rate(rate_kernel<4>, 4, 1); rate(rate_kernel<6>, 6, 1);
  rate(rate_kernel<2>, 2, 2); rate(rate_kernel<4>, 4, 2); rate(rate_kernel<4>, 4, 4);
  return bad != 0;
}
