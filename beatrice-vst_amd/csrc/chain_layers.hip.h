// chain_layers.hip.h -- the layer table of the per-hop chain (MODEL_SPEC 4.1, 4.2, 4.4) as Layer<>
// types, and the argument builders of the non-GEMM kernels; shared by the per-module forward passes
// (phone.hip, pitch.hip, wave.hip) and the paired front end (front.hip).  H = hops per step.
#pragma once
#include "conv_gemm.hip.h"
#include "engine.h"
#include "fused_small.hip.h"
#include "wave_tail.hip.h"
#include "rowchain.hip.h"  // launch_auto's many-row branch

namespace bhip {

//                                  CIN NOUT K  S  D  T      PRE       ACT       EPI       RES
template <int H>
struct PhoneLayers {
  using F2 = Layer<64, 128, 8, 4, 1, 8 * H, PRE_NONE, ACT_GELU, EPI_BIAS, false>;
  using F3 = Layer<128, 256, 4, 2, 1, 4 * H, PRE_NONE, ACT_GELU, EPI_BIAS, false>;
  using F4 = Layer<256, 256, 4, 2, 1, 2 * H, PRE_NONE, ACT_GELU, EPI_BIAS, false>;
  using F5 = Layer<256, 256, 4, 2, 1, H, PRE_NONE, ACT_GELU, EPI_BIAS, false>;
  using RBL = Layer<256, 256, 5, 1, 1, H, PRE_NONE, ACT_GELU, EPI_BIAS, true>;
  using OUTL = Layer<256, B_PHONE_CH, 1, 1, 1, H, PRE_NONE, ACT_NONE, EPI_BIAS, false>;
};
template <int H>
struct PitchLayers {
  using P1 = Layer<B_SPEC_BINS, 128, 3, 1, 1, H, PRE_NONE, ACT_GELU, EPI_BIAS, false>;
  using P23 = Layer<128, 128, 3, 1, 1, H, PRE_NONE, ACT_GELU, EPI_BIAS, true>;
  using POUT = Layer<128, B_PITCH_BINS, 1, 1, 1, H, PRE_NONE, ACT_NONE, EPI_BIAS, false>;
};

// waveform generator (MODEL_SPEC 4.4)
namespace wave_layers {
template <int H> using INP = Layer<B_PHONE_CH, B_HID, 1, 1, 1, H, PRE_NONE, ACT_NONE, EPI_BIAS, true>;
template <int D, int H> using C1 = Layer<B_HID, B_HID, 3, 1, D, H, PRE_NONE, ACT_GELU, EPI_BIAS, false>;
template <int H> using C2 = Layer<B_HID, B_HID, 1, 1, 1, H, PRE_NONE, ACT_NONE, EPI_BIAS, true>;
template <int H> using QL = Layer<B_HID, B_HID, 1, 1, 1, H, PRE_NONE, ACT_NONE, EPI_BIAS, false>;
template <int H> using SCORE = Layer<B_HID, B_KV_LEN, 1, 1, 1, H, PRE_NONE, ACT_NONE, EPI_SCALE, false, true>;
template <int CIN, int COUT, int R, int TIN> using UP = Layer<CIN, R * COUT, 2, 1, 1, TIN, PRE_LRELU, ACT_NONE, EPI_BIAS, false>;
template <int C, int D, int T> using RES = Layer<C, C, 3, 1, D, T, PRE_LRELU, ACT_NONE, EPI_BIAS, true>;
using TGQ = TileCfg<1, 1, 1, 2, 1>;  // grouped attention scores: 16 rows x 32 keys, K = 256 (one segment)
}  // namespace wave_layers

static inline F1Args f1_args(const PhoneWeights& w, const PhoneState& s) {
  return F1Args{s.d_in, s.audio, s.f[0], w.f1_w, w.f1_b, s.hop_in, s.hop_publish, s.hop_publish_wave, s.H, s.io_stride};
}
static inline LaunchInfo f1_info(const PhoneState& s) {
  return LaunchInfo{"phone.f1", 2.0 * s.B * s.H * 32 * 64 * 10, 4.0 * s.B * s.H * (160 + 32 * 64)};
}
static inline FftArgs fft_args(const PitchWeights& w, const PitchState& s) {
  return FftArgs{s.d_in, s.audio, s.spec, w.window, w.twiddle, s.hop_in, s.H, s.io_stride};
}
static inline LaunchInfo fft_info(const PitchState& s) {
  return LaunchInfo{"pitch.fft", s.B * s.H * (10.0 * 512 * 10 + 1024 * 2 + 512 * 30), 4.0 * s.B * s.H * (1024 + 160 + 512)};
}
static inline PitchHeadArgs head_args(const PitchWeights& w, const PitchState& s) {
  return PitchHeadArgs{s.H, s.logits, s.h, s.d_in, w.voi_w, w.voi_b, s.d_min_q, s.d_max_q, s.d_prev_q,
                       s.d_q_raw, s.d_q, s.d_feat, s.d_params, s.hop, s.io_stride, s.q_slots, s.B,
                       s.B == 1 && s.H == 1 && s.hop_mailbox != nullptr ? s.h_result : nullptr};
}
static inline LaunchInfo head_info(const PitchState& s) {
  return LaunchInfo{"pitch.head", 25.0 * s.B * s.H * 448, 4.0 * s.B * s.H * (448 + 160 + 128 + 8)};
}
static inline CondArgs cond_args(const WaveWeights& w, const WaveState& s) {
  return CondArgs{s.H, s.d_q, s.d_feat, s.q_slots, s.B, w.pitch_emb, w.feat_w, s.d_add_tab, s.d_add_idx, s.legacy ? nullptr : s.d_frm_tab, s.d_frm_idx, s.e,
                  s.front_hop ? s.front_hop : s.hop, s.front_next_out, s.io_slots, s.legacy ? 384 : B_PITCH_BINS};
}
static inline LaunchInfo cond_info(const WaveState& s) {
  return LaunchInfo{"wave.cond", 11.0 * s.B * s.H * 256, 4.0 * s.B * s.H * 256 * 4};
}

static inline TailArgs tail_args(const WaveWeights& w, const WaveState& s) {
  TailArgs ta{};
  ta.in = s.ya2; ta.state = s.tail.base; ta.fin_w = w.fin_w; ta.fin_b = w.fin_b; ta.d_out = s.d_out; ta.hop = s.hop; ta.io_stride = s.io_stride;
  if (s.B == 1 && s.H == 1) { ta.host_flag = s.h_flag; ta.seq = s.d_seq; }
  ta.w[0] = w.ra_w[1]; ta.b[0] = w.ra_b[1]; ta.w[1] = w.rb_w[1]; ta.b[1] = w.rb_b[1];
  ta.w[2] = w.up_w[2]; ta.b[2] = w.up_b[2]; ta.w[3] = w.ra_w[2]; ta.b[3] = w.ra_b[2]; ta.w[4] = w.rb_w[2]; ta.b[4] = w.rb_b[2];
  ta.w[5] = w.up_w[3]; ta.b[5] = w.up_b[3]; ta.w[6] = w.ra_w[3]; ta.b[6] = w.ra_b[3]; ta.w[7] = w.rb_w[3]; ta.b[7] = w.rb_b[3];
  return ta;
}
static inline LaunchInfo tail_info(const WaveState& s) {
  const double tail_macs = 2.0 * 20 * 192 * 64 + 20.0 * 128 * 128 + 2.0 * 80 * 96 * 32 + 80.0 * 64 * 48 + 2.0 * 240 * 48 * 16 + 240.0 * 112;
  const int rows = s.B * s.H;
  return LaunchInfo{"wave.tail", 2.0 * rows * tail_macs, 4.0 * (52000.0 + s.B * 2 * TAIL_STATE_FLOATS + rows * (22 * 64 + 240))};
}

// phone.out writes the 128-d vector either to the ring the k-NN kernel reads or, when no stream uses
// the codebook (skip_vq), straight to the module's output
static inline Ring phone_vector_ring(const PhoneState& s) { return Ring{s.d_phone, s.out_ch, s.H, s.out_slots}; }
static inline Ring phone_out_ring(const PhoneState& s) { return s.skip_vq ? phone_vector_ring(s) : s.raw; }

}  // namespace bhip
