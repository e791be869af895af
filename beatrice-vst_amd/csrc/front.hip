// front.hip -- content encoder and pitch estimator of one step as ONE chain of paired launches
// (pair.hip.h): the two modules only share the hop's audio (reference processor_core_2.cc:181-189
// calls ExtractPhone1 and EstimatePitch1 back to back on the same 160 samples), so every launch of
// the pitch estimator rides along with a launch of the content encoder, and the waveform generator's
// conditioning mix (needs only the pitch head's output) rides with the next one:
//
//   phone.f1+pitch.fft, f2, f3, f4, f5+pitch.p1, rb0+p2, rb1+p3, rb2+pitch.gru, rb3+pitch.out,
//   phone.gru+pitch.head, phone.out+wave.cond, [phone.vq]
//
// 11 launches instead of 20, identical arithmetic.  The pitch estimator's 768-thread first layer is
// paired with a layer that has few workgroups (the pair runs at the larger workgroup size; pairing it
// with phone.f2's 512 small workgroups cost 14.2 us against 7.4 + 8.6 separately).  Used by the
// batched path in the latency-bound regime (one hop per step, every GEMM in its few-row tiling);
// elsewhere the modules run one after the other (phone.hip, pitch.hip).
#include "chain_layers.hip.h"
#include "pair.hip.h"

namespace bhip {


template <class LA, class LB>
static void pair_conv(const char* na, const ConvArgs& a, const char* nb, const ConvArgs& b, hipStream_t st) {
  using OA = ConvOp<LA, TLat<LA>>;
  using OB = ConvOp<LB, TLat<LB>>;
  launch_pair<OA, OB>(OA::info(na, a), a, OA::grid(a), OB::info(nb, b), b, OB::grid(b), st);
}

bool front_forward(const PhoneWeights& pw, const PhoneState& ps, const PitchWeights& qw, const PitchState& qs,
                   const WaveWeights& ww, const WaveState& ws, hipStream_t st) {
  using PL = PhoneLayers<1>;
  using QL = PitchLayers<1>;
  const int B = ps.B;
  // paired regime: one hop per step and every layer in the few-row tiling (launch_auto's rule, conv_gemm.hip.h)
  if (ps.H != 1 || qs.H != 1 || ws.H != 1 || B * PL::F2::T > 2048 || qs.B != B || ws.B != B) return false;

  launch_pair<F1Op, FftOp>(f1_info(ps), f1_args(pw, ps), dim3(B, 1), fft_info(qs), fft_args(qw, qs), dim3(B, 1), st);
  launch_auto<PL::F2>("phone.f2", conv_args(ps.f[0], ps.f[1], pw.f_w[0], pw.f_b[0], ps.hop, B), st);
  launch_auto<PL::F3>("phone.f3", conv_args(ps.f[1], ps.f[2], pw.f_w[1], pw.f_b[1], ps.hop, B), st);
  launch_auto<PL::F4>("phone.f4", conv_args(ps.f[2], ps.f[3], pw.f_w[2], pw.f_b[2], ps.hop, B), st);
  pair_conv<PL::F5, QL::P1>("phone.f5", conv_args(ps.f[3], ps.f[4], pw.f_w[3], pw.f_b[3], ps.hop, B),
                            "pitch.p1", conv_args(qs.spec, qs.p[0], qw.p_w[0], qw.p_b[0], qs.hop, B), st);
  pair_conv<PL::RBL, QL::P23>("phone.rb", conv_args(ps.f[4], ps.rb[0], pw.rb_w[0], pw.rb_b[0], ps.hop, B),
                              "pitch.p23", conv_args(qs.p[0], qs.p[1], qw.p_w[1], qw.p_b[1], qs.hop, B), st);
  pair_conv<PL::RBL, QL::P23>("phone.rb", conv_args(ps.rb[0], ps.rb[1], pw.rb_w[1], pw.rb_b[1], ps.hop, B),
                              "pitch.p23", conv_args(qs.p[1], qs.p[2], qw.p_w[2], qw.p_b[2], qs.hop, B), st);
  {
    using OA = ConvOp<PL::RBL, TLat<PL::RBL>>;
    using OB = GruOp<128, 128>;
    const ConvArgs a = conv_args(ps.rb[1], ps.rb[2], pw.rb_w[2], pw.rb_b[2], ps.hop, B);
    const GruArgs g{qs.p[2], qs.h, qw.gru_wih, qw.gru_whh, qw.gru_bih, qw.gru_bhh, qs.hop, B, 0};
    launch_pair<OA, OB>(OA::info("phone.rb", a), a, OA::grid(a), OB::info("pitch.gru", g), g, OB::grid(g), st);
  }
  pair_conv<PL::RBL, QL::POUT>("phone.rb", conv_args(ps.rb[2], ps.rb[3], pw.rb_w[3], pw.rb_b[3], ps.hop, B),
                               "pitch.out", conv_args(qs.h, qs.logits, qw.out_w, qw.out_b, qs.hop, B), st);
  {
    using OA = GruOp<256, 256>;
    const GruArgs g{ps.rb[3], ps.h, pw.gru_wih, pw.gru_whh, pw.gru_bih, pw.gru_bhh, ps.hop, B, 0};
    launch_pair<OA, HeadOp>(OA::info("phone.gru", g), g, OA::grid(g), head_info(qs), head_args(qw, qs), dim3(B, 1), st);
  }
  {
    using OA = ConvOp<PL::OUTL, TLat<PL::OUTL>>;
    const ConvArgs a = conv_args(ps.h, phone_out_ring(ps), pw.out_w, pw.out_b, ps.hop, B);
    launch_pair<OA, CondOp>(OA::info("phone.out", a), a, OA::grid(a), cond_info(ws), cond_args(ww, ws), dim3(B, 1), st);
  }
  phone_vq(pw, ps, st);
  return true;
}

}  // namespace bhip
