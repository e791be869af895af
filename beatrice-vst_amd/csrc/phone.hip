// phone.hip -- content encoder forward pass (MODEL_SPEC 4.1), the body of
// Beatrice20rc0_ExtractPhone1 (reference lib/beatricelib/beatrice.h:243-247) for B streams and H
// consecutive hops per step (H = 1: the real-time per-hop path; H > 1: block mode, batch.hip).
#include <cstdlib>

#include "chain_layers.hip.h"
#include "team.hip.h"

namespace bhip {

bool PhoneState::create(int B_, int H_, float* shared_in, int out_slots_, bool pipe_slack, int out_ch_) {
  B = B_; H = H_; out_slots = out_slots_; out_ch = out_ch_;
  const int x = pipe_slack ? 1 : 0;  // the reader of a ring may be one step behind its writer
  auto slots = [&](int n0, int hist) { return 1 + (hist + n0 * H - 1) / (n0 * H); };
  std::vector<RingSpec> specs = {
      {&audio, 1, B_IN_HOP * H, slots(B_IN_HOP, 5)},
      {&f[0], 64, 32 * H, slots(32, 4) + x}, {&f[1], 128, 8 * H, slots(8, 2) + x}, {&f[2], 256, 4 * H, slots(4, 2) + x},
      {&f[3], 256, 2 * H, slots(2, 2) + x}, {&f[4], 256, H, slots(1, 4) + x},
      {&rb[0], 256, H, slots(1, 4) + x}, {&rb[1], 256, H, slots(1, 4) + x}, {&rb[2], 256, H, slots(1, 4) + x}, {&rb[3], 256, H, 1 + x},
      {&h, 256, H, slots(1, 1) + x}, {&raw, out_ch, H, 1 + x},
  };
  if (!arena.build(B, specs)) return false;
  if (shared_in) { d_in = shared_in; owns_in = false; }
  else {
    // eight extra words behind the audio: a mailbox the 1-stream ABI fills with the same copy as the audio
    // (step counter | k-NN k | codebook pointers: abi.hip)
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_in), sizeof(float) * (B * H * B_IN_HOP + kMailboxWords)));
    BHIP_TRY(hipMemset(d_in, 0, sizeof(float) * (B * H * B_IN_HOP + kMailboxWords)));
    owns_in = true;
    hop_mailbox = reinterpret_cast<int*>(d_in + (size_t)B * H * B_IN_HOP);
  }
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_phone), sizeof(float) * B * H * out_ch * out_slots));
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_cbT), sizeof(float*) * B * H));
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_cnorm), sizeof(float*) * B * H));
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_vqk), sizeof(int) * B));
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_hop), 2 * sizeof(int)));  // [0] step counter, [1] resident-I/O slot
  BHIP_TRY(hipMemset(d_phone, 0, sizeof(float) * B * H * out_ch * out_slots));
  BHIP_TRY(hipMemset(d_cbT, 0, sizeof(float*) * B * H));
  BHIP_TRY(hipMemset(d_cnorm, 0, sizeof(float*) * B * H));
  BHIP_TRY(hipMemset(d_vqk, 0, sizeof(int) * B));
  BHIP_TRY(hipMemset(d_hop, 0, 2 * sizeof(int)));
  hop = d_hop; hop_in = d_hop;
  team_off = false;
  if (B == 1 && H == 1 && hipFuncSetAttribute(reinterpret_cast<const void*>(team::phone_team_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, team::kLdsFloats * 4) == hipSuccess &&
      team_capacity_ok(reinterpret_cast<const void*>(team::phone_team_kernel), team::NWG, team::NTHR, team::kLdsFloats * 4)) {
    // the 1-stream ABI's team launch (team.hip.h), where the device can hold the whole team at once; tag 0 = "never written"
    team_granules = team::kPhoneGranules;
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_team_xb), sizeof(unsigned long long) * team::kPhoneGranules));
    BHIP_TRY(hipMemset(d_team_xb, 0, sizeof(unsigned long long) * team::kPhoneGranules));
    // (pinned host memory, written by the kernel only when a wait was given up: the host reads it after every call for free)
    BHIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&d_team_dead), sizeof(int), hipHostMallocDefault));
    *d_team_dead = 0;
  }
  // hipMemset is asynchronous and runs on the NULL stream, which the (non-blocking) compute streams
  // do not wait for: make every initialisation above visible before the first kernel can start
  BHIP_TRY(hipDeviceSynchronize());
  return true;
}
void PhoneState::destroy() {
  arena.release();
  if (owns_in && d_in) (void)hipFree(d_in);
  if (d_phone) (void)hipFree(d_phone);
  if (d_cbT) (void)hipFree(d_cbT);
  if (d_cnorm) (void)hipFree(d_cnorm);
  if (d_vqk) (void)hipFree(d_vqk);
  if (d_hop) (void)hipFree(d_hop);
  if (d_team_xb) (void)hipFree(d_team_xb);
  if (d_team_dead) (void)hipHostFree(d_team_dead);
  d_team_xb = nullptr; d_team_dead = nullptr;
  d_in = d_phone = nullptr; d_cbT = d_cnorm = nullptr; d_vqk = d_hop = nullptr;
}

#define MISC_LAUNCH(NAME, FLOPS, BYTES, KERNEL, GRID, BLOCK, ...)                              \
  launch_site(LaunchInfo{NAME, (double)(FLOPS), (double)(BYTES)}, st,                          \
              [&] { hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, st, __VA_ARGS__); })

// k-NN codebook lookup (skipped while no stream uses it) and the module's counter increment
void phone_vq(const PhoneWeights&, const PhoneState& s, hipStream_t st) {
  const int B = s.B, H = s.H;
  if (!s.skip_vq) {
    VqArgs v{H, s.raw, phone_vector_ring(s), s.hop, s.d_cbT, s.d_cnorm, s.d_vqk};
    MISC_LAUNCH("phone.vq", 0 /* k-dependent: 131 kFLOP per stream-hop with k > 0, pass-through at k = 0 */, 4.0 * B * H * 256,
                phone_vq_kernel, dim3(B * H), dim3(512), v);
  }
  if (s.advance_hop) MISC_LAUNCH("hop_advance", 0, 4, hop_advance_kernel, dim3(1), dim3(1), s.hop);
}

template <int H>
static void phone_forward_h(const PhoneWeights& w, const PhoneState& s, hipStream_t st) {
  using PL = PhoneLayers<H>;
  const int B = s.B;
  const F1Args fa = f1_args(w, s);
  static const bool no_team = bhip::meas_env("BEATRICE_HIP_NO_TEAM") != nullptr;
  static const bool no_team_head = bhip::meas_env("BEATRICE_HIP_NO_TEAM_HEAD") != nullptr;   // A/B switch: f1 as a launch of its own again
  const bool use_team = H == 1 && B == 1 && s.d_team_xb != nullptr && !no_team && !s.team_off;
  // the 1-stream ABI's contexts (one counter, nothing to publish): f1 runs at the head of the team launch (team.hip.h with_f1)
  const bool f1_in_team = use_team && !no_team_head && fa.hop == s.hop && fa.hop_publish == nullptr && fa.io_stride == 0;
  if (!f1_in_team) launch_site(f1_info(s), st, [&] { hipLaunchKernelGGL(phone_f1_kernel, dim3(B, H), dim3(256), 0, st, fa); });
  if (use_team) {   // one stream: f2 .. f5 and the residual blocks as ONE launch (team.hip.h)
    using namespace team;
    PhoneTeamArgs a{};
    gran_t* g = s.d_team_xb;
    auto take = [&g](size_t n) { gran_t* p = g; g += n; return p; };
    a.f[0] = Tensor{s.f[0], nullptr};
    a.f[1] = Tensor{s.f[1], take(8 * 128)}; a.f[2] = Tensor{s.f[2], take(4 * 256)}; a.f[3] = Tensor{s.f[3], take(2 * 256)}; a.f[4] = Tensor{s.f[4], take(256)};
    for (int i = 0; i < 4; ++i) { a.rb[i] = Tensor{s.rb[i], take(256)}; a.rb_w[i] = w.rb_w[i]; a.rb_b[i] = w.rb_b[i]; a.f_w[i] = w.f_w[i]; a.f_b[i] = w.f_b[i]; }
    a.hop = s.hop; a.dead = s.d_team_dead;
    a.with_f1 = f1_in_team ? 1 : 0; a.f1 = fa;
    launch_site(LaunchInfo{"phone.team", 2.0 * (8 * 512.0 * 128 + 4 * 512.0 * 256 + 2 * 1024.0 * 256 + 1024.0 * 256 + 4 * 1280.0 * 256),
                           4.0 * (512.0 * 128 + 512.0 * 256 + 2 * 1024.0 * 256 + 4 * 1280.0 * 256)},
                st, [&] { hipLaunchKernelGGL(phone_team_kernel, dim3(NWG), dim3(NTHR), kLdsFloats * 4, st, a); });
  } else {
  launch_auto<typename PL::F2>("phone.f2", conv_args(s.f[0], s.f[1], w.f_w[0], w.f_b[0], s.hop, B), st);
  launch_auto<typename PL::F3>("phone.f3", conv_args(s.f[1], s.f[2], w.f_w[1], w.f_b[1], s.hop, B), st);
  launch_auto<typename PL::F4>("phone.f4", conv_args(s.f[2], s.f[3], w.f_w[2], w.f_b[2], s.hop, B), st);
  launch_auto<typename PL::F5>("phone.f5", conv_args(s.f[3], s.f[4], w.f_w[3], w.f_b[3], s.hop, B), st);
  const Ring* cur = &s.f[4];
  for (int i = 0; i < 4; ++i) {
    launch_auto<typename PL::RBL>("phone.rb", conv_args(*cur, s.rb[i], w.rb_w[i], w.rb_b[i], s.hop, B), st);
    cur = &s.rb[i];
  }
  }
  if (s.after_convs != nullptr) s.after_convs(s.after_convs_arg);
  for (int t = 0; t < H; ++t) {  // the recurrence is sequential over the hops of the step
    GruArgs ga{s.rb[3], s.h, w.gru_wih, w.gru_whh, w.gru_bih, w.gru_bhh, s.hop, B, t};
    launch_gru<256, 256>("phone.gru", ga, st);
  }
  if (s.out_ch == 256) {  // the legacy generations' 256-wide phone vector (MODEL_SPEC 6.1); they have no codebook step
    launch_auto<Layer<256, 256, 1, 1, 1, H, PRE_NONE, ACT_NONE, EPI_BIAS, false>>("phone.out", conv_args(s.h, phone_out_ring(s), w.out_w, w.out_b, s.hop, B), st);
  } else {
    launch_auto<typename PL::OUTL>("phone.out", conv_args(s.h, phone_out_ring(s), w.out_w, w.out_b, s.hop, B), st);
  }
  phone_vq(w, s, st);
}

void phone_forward(const PhoneWeights& w, const PhoneState& s, hipStream_t st) {
  switch (s.H) {
    case 1: phone_forward_h<1>(w, s, st); break;
    case 2: phone_forward_h<2>(w, s, st); break;
    case 4: phone_forward_h<4>(w, s, st); break;
    default: phone_forward_h<8>(w, s, st); break;
  }
}

}  // namespace bhip
