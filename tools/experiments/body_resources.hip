// Every body of the tick launch as a kernel of its own under the launch's register budget (512 threads, two workgroups per CU):
// registers and spills per body (the table kernel reports only the maximum over all of them).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I beatrice-vst_amd/csrc -c tools/experiments/body_resources.hip -o /tmp/br.o
//   then tools/experiments/body_resources.sh
#include <hip/hip_runtime.h>
#include <type_traits>

#include "tick.hip.h"

template <class Op, int TAG>
__global__ __launch_bounds__(512, 4) void body_kernel(const typename Op::Args a) {
  __shared__ __attribute__((aligned(16))) float lds[Op::LDS_FLOATS > 0 ? Op::LDS_FLOATS : 1];
  if (threadIdx.x == 0) { stepc::pair[0] = 1; stepc::pair[1] = 0; }
  __syncthreads();
  if ((int)threadIdx.x < Op::NTHR) Op::template run_t<false>(a, blockIdx.x, blockIdx.y, lds);
}
#define INST(H, NAME) template __global__ void body_kernel<typename tick::Ops<H>::NAME, H>(const typename tick::Ops<H>::NAME::Args)
#define BOTH(NAME) INST(1, NAME); INST(2, NAME)
BOTH(OpF2); BOTH(OpF3); BOTH(OpF4); BOTH(OpF5); BOTH(OpRB); BOTH(OpOUT); BOTH(OpP1); BOTH(OpP23); BOTH(OpPOUT); BOTH(OpINP);
BOTH(OpUP1); BOTH(OpRES1A); BOTH(OpRES1B); BOTH(OpUP2); BOTH(T1); BOTH(T2); BOTH(T3); BOTH(GruQ); BOTH(GruP); BOTH(Vq);
INST(2, GruQ1); INST(2, GruP1);
template __global__ void body_kernel<F1Op2, 0>(const F1Op2::Args);
template __global__ void body_kernel<FftOp2, 0>(const FftOp2::Args);
template __global__ void body_kernel<HeadOp8, 0>(const HeadOp8::Args);
template __global__ void body_kernel<CondOp2, 0>(const CondOp2::Args);
template __global__ void body_kernel<rc::BlockAOp<1, 1>, 0>(const rc::BlockAArgs);
template __global__ void body_kernel<rc::BlockAOp<8, 1>, 0>(const rc::BlockAArgs);
template __global__ void body_kernel<rc::BlockAOp<1, 2>, 0>(const rc::BlockAArgs);
template __global__ void body_kernel<rc::BlockAOp<8, 2>, 0>(const rc::BlockAArgs);
template __global__ void body_kernel<rc::BlockBOpH<1>, 0>(const rc::BlockBArgs);
template __global__ void body_kernel<rc::BlockBOpH<2>, 0>(const rc::BlockBArgs);
template __global__ void body_kernel<rc::BlockBqOpH<1>, 0>(const rc::BlockBqArgs);
template __global__ void body_kernel<rc::BlockBqOpH<2>, 0>(const rc::BlockBqArgs);
