// Is v_mfma_f32_16x16x4_f32 symmetric in its operands, bit for bit?  D = A.B with A = activations [16 x K], B = weights [K x 16]
// against D^T = B^T.A^T (weights as the A operand, activations as the B operand): the transposed form gives a lane four
// consecutive COLUMNS of one row (one 16-byte store per accumulator) instead of four rows of one column.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench/mfma_swap.hip -o tools/microbench/mfma_swap && tools/microbench/mfma_swap
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 256;
__global__ void k(const float* x /*[16][K]*/, const float* w /*[K][16]*/, float* d0 /*[16][16] x.w*/, float* d1 /*same, from the swapped form*/) {
  const int l = threadIdx.x, i = l & 15, kq = l >> 4;
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  for (int kk = 0; kk < K; kk += 4) {
    const float xv = x[i * K + kk + kq], wv = w[(kk + kq) * 16 + i];
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, wv, a0, 0, 0, 0);   // rows = x rows, cols = w cols
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, a1, 0, 0, 0);   // rows = w cols, cols = x rows
  }
  for (int e = 0; e < 4; ++e) {
    d0[(kq * 4 + e) * 16 + i] = a0[e];          // D[row kq*4+e][col i]
    d1[i * 16 + kq * 4 + e] = a1[e];            // D^T[row kq*4+e = w col][col i = x row] -> stored as D[x row][w col]
  }
}
int main() {
  std::vector<float> x(16 * K), w(K * 16), ref(256);
  unsigned s = 12345;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : x) v = rnd() * 3.1f;
  for (auto& v : w) v = rnd() * 0.37f;
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { float acc = 0; for (int kk = 0; kk < K; ++kk) acc = fmaf(x[r * K + kk], w[kk * 16 + c], acc); ref[r * 16 + c] = acc; }
  float *dx, *dw, *d0, *d1;
  (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&dw, w.size() * 4); (void)hipMalloc(&d0, 1024); (void)hipMalloc(&d1, 1024);
  (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dw, d0, d1);
  std::vector<float> h0(256), h1(256);
  (void)hipMemcpy(h0.data(), d0, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(h1.data(), d1, 1024, hipMemcpyDeviceToHost);
  int bad01 = 0, bad0r = 0;
  for (int i = 0; i < 256; ++i) { bad01 += std::memcmp(&h0[i], &h1[i], 4) != 0; bad0r += std::memcmp(&h0[i], &ref[i], 4) != 0; }
  std::printf("mfma(x, w) vs mfma(w, x) transposed: %d of 256 differ; mfma(x, w) vs the fmaf chain: %d differ\n", bad01, bad0r);
  return bad01 || bad0r;
}
