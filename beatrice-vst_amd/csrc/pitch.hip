// pitch.hip -- pitch estimator forward pass (MODEL_SPEC 4.2), the body of
// Beatrice20rc0_EstimatePitch1 (reference lib/beatricelib/beatrice.h:266-271) for B streams and H
// consecutive hops per step.
#include <cstdlib>

#include "chain_layers.hip.h"
#include "team.hip.h"

namespace bhip {

bool PitchState::create(int B_, int H_, float* shared_in, bool with_params, bool pipe_slack, int bins_) {
  B = B_; H = H_; bins = bins_;
  const int x = pipe_slack ? 1 : 0;  // see PhoneState::create; the GRU state is also read by the pitch head, two stages on
  q_slots = pipe_slack ? 2 : 1;
  auto slots = [&](int n0, int hist) { return 1 + (hist + n0 * H - 1) / (n0 * H); };
  std::vector<RingSpec> specs = {
      {&audio, 1, B_IN_HOP * H, slots(B_IN_HOP, B_PITCH_HIST)},
      {&spec, B_SPEC_BINS, H, slots(1, 2) + x}, {&p[0], 128, H, slots(1, 2) + x}, {&p[1], 128, H, slots(1, 2) + x}, {&p[2], 128, H, 1 + x},
      {&h, 128, H, slots(1, 1) + 2 * x}, {&logits, bins, H, 1 + x},
  };
  if (!arena.build(B, specs)) return false;
  if (shared_in) { d_in = shared_in; owns_in = false; }
  else {
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_in), sizeof(float) * (B * H * B_IN_HOP + kMailboxWords)));
    BHIP_TRY(hipMemset(d_in, 0, sizeof(float) * (B * H * B_IN_HOP + kMailboxWords)));
    owns_in = true;
    hop_mailbox = reinterpret_cast<int*>(d_in + (size_t)B * H * B_IN_HOP);  // see PhoneState::create
  }
  int** per_stream[] = {&d_min_q, &d_max_q, &d_prev_q};
  for (int** p : per_stream) {
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(p), sizeof(int) * B));
    BHIP_TRY(hipMemset(*p, 0, sizeof(int) * B));
  }
  // (one stream, one hop per call -- the 1-stream ABI: the raw bin sits right behind the four features, so that EstimatePitch1 fetches
  //  its five result words with ONE copy command instead of two)
  q_raw_in_feat = B == 1 && H == 1 && q_slots == 1;
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_feat), sizeof(float) * (4 * B * H * q_slots + 4)));
  BHIP_TRY(hipMemset(d_feat, 0, sizeof(float) * (4 * B * H * q_slots + 4)));
  if (q_raw_in_feat) d_q_raw = reinterpret_cast<int*>(d_feat + 4);
  int** per_row[] = {&d_q_raw, &d_q};
  for (int** p : per_row) {
    if (p == &d_q_raw && q_raw_in_feat) continue;
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(p), sizeof(int) * B * H * q_slots));
    BHIP_TRY(hipMemset(*p, 0, sizeof(int) * B * H * q_slots));
  }
  std::vector<int> lo(B, 1), hi(B, bins - 1);
  BHIP_TRY(hipMemcpy(d_min_q, lo.data(), sizeof(int) * B, hipMemcpyHostToDevice));
  BHIP_TRY(hipMemcpy(d_max_q, hi.data(), sizeof(int) * B, hipMemcpyHostToDevice));
  if (with_params) {
    std::vector<PitchParams> pp(B, PitchParams{52.0, 1.0, 0.0, 0.0, 0, 0});
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_params), sizeof(PitchParams) * B));
    BHIP_TRY(hipMemcpy(d_params, pp.data(), sizeof(PitchParams) * B, hipMemcpyHostToDevice));
  }
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_hop), 2 * sizeof(int)));
  BHIP_TRY(hipMemset(d_hop, 0, 2 * sizeof(int)));
  hop = d_hop; hop_in = d_hop;
  team_off = false;
  if (B == 1 && H == 1 && hipFuncSetAttribute(reinterpret_cast<const void*>(team::pitch_team_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, team::kLdsFloats * 4) == hipSuccess &&
      team_capacity_ok(reinterpret_cast<const void*>(team::pitch_team_kernel), team::kPitchTeamWgs, team::NTHR, team::kLdsFloats * 4)) {
    // the 1-stream ABI's team launch (team.hip.h), where the device can hold the whole team at once; tag 0 = "never written"
    team_granules = team::kPitchGranules;
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_team_xb), sizeof(unsigned long long) * team::kPitchGranules));
    BHIP_TRY(hipMemset(d_team_xb, 0, sizeof(unsigned long long) * team::kPitchGranules));
    // (pinned host memory, written by the kernel only when a wait was given up: the host reads it after every call for free)
    BHIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&d_team_dead), sizeof(int), hipHostMallocDefault));
    *d_team_dead = 0;
  }
  BHIP_TRY(hipDeviceSynchronize());  // NULL-stream memsets vs non-blocking compute streams
  return true;
}
void PitchState::destroy() {
  arena.release();
  if (owns_in && d_in) (void)hipFree(d_in);
  void* ptrs[] = {d_min_q, d_max_q, d_prev_q, q_raw_in_feat ? nullptr : d_q_raw, d_q, d_feat, d_params, d_hop, d_team_xb};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (d_team_dead) (void)hipHostFree(d_team_dead);
  d_team_xb = nullptr; d_team_dead = nullptr;
  d_in = d_feat = nullptr; d_min_q = d_max_q = d_prev_q = d_q_raw = d_q = d_hop = nullptr; d_params = nullptr;
}

#define MISC_LAUNCH(NAME, FLOPS, BYTES, KERNEL, GRID, BLOCK, ...)                              \
  launch_site(LaunchInfo{NAME, (double)(FLOPS), (double)(BYTES)}, st,                          \
              [&] { hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, st, __VA_ARGS__); })

template <int H>
static void pitch_forward_h(const PitchWeights& w, const PitchState& s, hipStream_t st) {
  using QL = PitchLayers<H>;
  const int B = s.B;
  const FftArgs fa = fft_args(w, s);
  launch_site(fft_info(s), st, [&] { hipLaunchKernelGGL(pitch_fft_kernel, dim3(B, H), dim3(256), 0, st, fa); });
  static const bool no_team = bhip::meas_env("BEATRICE_HIP_NO_TEAM") != nullptr;
  if (H == 1 && B == 1 && s.d_team_xb != nullptr && !no_team && !s.team_off) {   // one stream: the three convolutions as ONE launch (team.hip.h)
    using namespace team;
    PitchTeamArgs a{};
    a.spec = Tensor{s.spec, nullptr};
    for (int i = 0; i < 3; ++i) { a.p[i] = Tensor{s.p[i], s.d_team_xb + 128 * i}; a.p_w[i] = w.p_w[i]; a.p_b[i] = w.p_b[i]; }
    a.hop = s.hop; a.dead = s.d_team_dead;
    launch_site(LaunchInfo{"pitch.team", 2.0 * (1536.0 * 128 + 2 * 384.0 * 128), 4.0 * (1536.0 * 128 + 2 * 384.0 * 128)}, st,
                [&] { hipLaunchKernelGGL(pitch_team_kernel, dim3(kPitchTeamWgs), dim3(NTHR), kLdsFloats * 4, st, a); });
  } else {
    launch_auto<typename QL::P1>("pitch.p1", conv_args(s.spec, s.p[0], w.p_w[0], w.p_b[0], s.hop, B), st);
    launch_auto<typename QL::P23>("pitch.p23", conv_args(s.p[0], s.p[1], w.p_w[1], w.p_b[1], s.hop, B), st);
    launch_auto<typename QL::P23>("pitch.p23", conv_args(s.p[1], s.p[2], w.p_w[2], w.p_b[2], s.hop, B), st);
  }
  for (int t = 0; t < H; ++t) {
    GruArgs ga{s.p[2], s.h, w.gru_wih, w.gru_whh, w.gru_bih, w.gru_bhh, s.hop, B, t};
    launch_gru<128, 128>("pitch.gru", ga, st);
  }
  if (s.bins == 384) {  // the legacy generations' 384 pitch classes (MODEL_SPEC 6.2)
    launch_auto<Layer<128, 384, 1, 1, 1, H, PRE_NONE, ACT_NONE, EPI_BIAS, false>>("pitch.out", conv_args(s.h, s.logits, w.out_w, w.out_b, s.hop, B), st);
  } else {
    launch_auto<typename QL::POUT>("pitch.out", conv_args(s.h, s.logits, w.out_w, w.out_b, s.hop, B), st);
  }
  const PitchHeadArgs a = head_args(w, s);
  launch_site(head_info(s), st, [&] { hipLaunchKernelGGL(pitch_head_kernel, dim3(B), dim3(64), 0, st, a); });
  if (s.advance_hop) MISC_LAUNCH("hop_advance", 0, 4, hop_advance_kernel, dim3(1), dim3(1), s.hop);
}

void pitch_forward(const PitchWeights& w, const PitchState& s, hipStream_t st) {
  switch (s.H) {
    case 1: pitch_forward_h<1>(w, s, st); break;
    case 2: pitch_forward_h<2>(w, s, st); break;
    case 4: pitch_forward_h<4>(w, s, st); break;
    default: pitch_forward_h<8>(w, s, st); break;
  }
}

}  // namespace bhip
