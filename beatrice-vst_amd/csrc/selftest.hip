// selftest.hip -- device-side exhaustive checks of the packed scalar functions of spec_math.hip.h against their scalar
// definitions (which the parity tests pin to the oracle).  Test infrastructure exported from the product library because
// the functions under test are device code: BeatriceHip_MathSelfTest(which) sweeps all 2^32 float32 bit patterns (NaNs
// excluded) and returns the number of inputs whose results differ in any bit; -1 on a HIP failure.
//   which: 0 exp2 vs exp, 1 tanh2 vs tanh, 2 gelu2 vs gelu, 3 sigmoid2 vs sigmoid
#include <hip/hip_runtime.h>

#include "engine.h"
#include "spec_math.hip.h"

namespace {
template <int WHICH>
__global__ __launch_bounds__(256) void sweep_kernel(unsigned long long* bad, unsigned* first_bad) {
  // thread t of the grid covers patterns 2 (t + k * stride), + 1
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 31); i += stride) {
    const uint32_t b0 = (uint32_t)(2 * i), b1 = b0 + 1;
    const float x0 = __uint_as_float(b0), x1 = __uint_as_float(b1);
    const bsp::f32x2 x{x0, x1};
    bsp::f32x2 got;
    float w0, w1;
    if (WHICH == 0) { got = bsp::exp2(x); w0 = bsp::exp(x0); w1 = bsp::exp(x1); }
    else if (WHICH == 1) { got = bsp::tanh2(x); w0 = bsp::tanh(x0); w1 = bsp::tanh(x1); }
    else if (WHICH == 2) { got = bsp::gelu2(x); w0 = bsp::gelu(x0); w1 = bsp::gelu(x1); }
    else { got = bsp::sigmoid2(x); w0 = bsp::sigmoid(x0); w1 = bsp::sigmoid(x1); }
    const bool n0 = x0 != x0, n1 = x1 != x1;
    if (!n0 && __float_as_uint(got.x) != __float_as_uint(w0)) { ++mine; atomicMin(first_bad, b0); }
    if (!n1 && __float_as_uint(got.y) != __float_as_uint(w1)) { ++mine; atomicMin(first_bad, b1); }
  }
  if (mine) atomicAdd(bad, mine);
}
}  // namespace

extern "C" long long BeatriceHip_MathSelfTest(int which, unsigned* first_bad_bits) {
  unsigned long long* d_bad = nullptr;
  unsigned* d_first = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d_bad), 8) != hipSuccess) return -1;
  if (hipMalloc(reinterpret_cast<void**>(&d_first), 4) != hipSuccess) { (void)hipFree(d_bad); return -1; }
  (void)hipMemset(d_bad, 0, 8);
  (void)hipMemset(d_first, 0xff, 4);
  const dim3 grid(256 * 16), block(256);
  switch (which) {
    case 0: hipLaunchKernelGGL(sweep_kernel<0>, grid, block, 0, 0, d_bad, d_first); break;
    case 1: hipLaunchKernelGGL(sweep_kernel<1>, grid, block, 0, 0, d_bad, d_first); break;
    case 2: hipLaunchKernelGGL(sweep_kernel<2>, grid, block, 0, 0, d_bad, d_first); break;
    case 3: hipLaunchKernelGGL(sweep_kernel<3>, grid, block, 0, 0, d_bad, d_first); break;
    default: (void)hipFree(d_bad); (void)hipFree(d_first); return -1;
  }
  unsigned long long bad = 0;
  unsigned first = 0xffffffffu;
  const bool ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost) == hipSuccess &&
                  hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d_bad);
  (void)hipFree(d_first);
  if (first_bad_bits) *first_bad_bits = first;
  return ok ? (long long)bad : -1;
}
