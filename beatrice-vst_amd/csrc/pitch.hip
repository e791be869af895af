// pitch.hip -- pitch estimator forward pass (MODEL_SPEC 4.2), the body of
// Beatrice20rc0_EstimatePitch1 (reference lib/beatricelib/beatrice.h:266-271) for B streams.
#include "conv_gemm.hip.h"
#include "engine.h"
#include "fused_small.hip.h"

namespace bhip {

bool PitchState::create(int B_, float* shared_in, bool with_params) {
  B = B_;
  std::vector<RingSpec> specs = {
      {&audio, 1, B_IN_HOP, 7},
      {&spec, B_SPEC_BINS, 1, 3}, {&p[0], 128, 1, 3}, {&p[1], 128, 1, 3}, {&p[2], 128, 1, 1},
      {&gi, 384, 1, 1}, {&gh, 384, 1, 1}, {&h, 128, 1, 2}, {&logits, B_PITCH_BINS, 1, 1},
  };
  if (!arena.build(B, specs)) return false;
  if (shared_in) { d_in = shared_in; owns_in = false; }
  else {
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_in), sizeof(float) * B * B_IN_HOP));
    BHIP_TRY(hipMemset(d_in, 0, sizeof(float) * B * B_IN_HOP));
    owns_in = true;
  }
  int** ints[] = {&d_min_q, &d_max_q, &d_prev_q, &d_q_raw, &d_q};
  for (int** p : ints) {
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(p), sizeof(int) * B));
    BHIP_TRY(hipMemset(*p, 0, sizeof(int) * B));
  }
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_feat), sizeof(float) * 4 * B));
  BHIP_TRY(hipMemset(d_feat, 0, sizeof(float) * 4 * B));
  std::vector<int> lo(B, 1), hi(B, B_PITCH_BINS - 1);
  BHIP_TRY(hipMemcpy(d_min_q, lo.data(), sizeof(int) * B, hipMemcpyHostToDevice));
  BHIP_TRY(hipMemcpy(d_max_q, hi.data(), sizeof(int) * B, hipMemcpyHostToDevice));
  if (with_params) {
    std::vector<PitchParams> pp(B, PitchParams{52.0, 1.0, 0.0, 0.0, 0, 0});
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_params), sizeof(PitchParams) * B));
    BHIP_TRY(hipMemcpy(d_params, pp.data(), sizeof(PitchParams) * B, hipMemcpyHostToDevice));
  }
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_hop), sizeof(int)));
  BHIP_TRY(hipMemset(d_hop, 0, sizeof(int)));
  hop = d_hop;
  // hipMemset is asynchronous and runs on the NULL stream, which the (non-blocking) compute streams
  // do not wait for: make every initialisation above visible before the first kernel can start
  BHIP_TRY(hipDeviceSynchronize());
  return true;
}
void PitchState::destroy() {
  arena.release();
  if (owns_in && d_in) (void)hipFree(d_in);
  void* ptrs[] = {d_min_q, d_max_q, d_prev_q, d_q_raw, d_q, d_feat, d_params, d_hop};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  d_in = d_feat = nullptr; d_min_q = d_max_q = d_prev_q = d_q_raw = d_q = d_hop = nullptr; d_params = nullptr;
}

using P1 = Layer<B_SPEC_BINS, 128, 3, 1, 1, 1, PRE_NONE, ACT_GELU, EPI_BIAS, false>;
using P23 = Layer<128, 128, 3, 1, 1, 1, PRE_NONE, ACT_GELU, EPI_BIAS, true>;
using PGATE = Layer<128, 384, 1, 1, 1, 1, PRE_NONE, ACT_NONE, EPI_BIAS, false>;
using POUT = Layer<128, B_PITCH_BINS, 1, 1, 1, 1, PRE_NONE, ACT_NONE, EPI_BIAS, false>;

#define MISC_LAUNCH(NAME, FLOPS, BYTES, KERNEL, GRID, BLOCK, ...)                              \
  launch_site(LaunchInfo{NAME, (double)(FLOPS), (double)(BYTES)}, st,                          \
              [&] { hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, st, __VA_ARGS__); })

void pitch_forward(const PitchWeights& w, const PitchState& s, hipStream_t st) {
  const int B = s.B;
  MISC_LAUNCH("pitch.fft", B * (10.0 * 512 * 10 + 1024 * 2 + 512 * 30), 4.0 * B * (1024 + 160 + 512), pitch_fft_kernel, dim3(B),
              dim3(256), s.d_in, s.audio, s.spec, w.window, w.twiddle, s.hop);
  launch_auto<P1>("pitch.p1", conv_args(s.spec, s.p[0], w.p_w[0], w.p_b[0], s.hop, B), st);
  launch_auto<P23>("pitch.p23", conv_args(s.p[0], s.p[1], w.p_w[1], w.p_b[1], s.hop, B), st);
  launch_auto<P23>("pitch.p23", conv_args(s.p[1], s.p[2], w.p_w[2], w.p_b[2], s.hop, B), st);
  GruArgs ga{s.p[2], s.h, w.gru_wih, w.gru_whh, w.gru_bih, w.gru_bhh, s.hop, B};
  launch_gru<128, 128>("pitch.gru", ga, st);
  launch_auto<POUT>("pitch.out", conv_args(s.h, s.logits, w.out_w, w.out_b, s.hop, B), st);
  PitchHeadArgs a{s.logits.base, s.h, s.d_in, w.voi_w, w.voi_b, s.d_min_q, s.d_max_q, s.d_prev_q,
                  s.d_q_raw, s.d_q, s.d_feat, s.d_params, s.hop};
  MISC_LAUNCH("pitch.head", 25.0 * B * 448, 4.0 * B * (448 + 160 + 128 + 8), pitch_head_kernel, dim3(B), dim3(64), a);
  if (s.advance_hop) MISC_LAUNCH("hop_advance", 0, 4, hop_advance_kernel, dim3(1), dim3(1), s.hop);
}

}  // namespace bhip
