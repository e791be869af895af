// front.hip -- content encoder and pitch estimator of one step as ONE chain of paired launches
// (fuse.hip.h): the two modules only share the hop's audio (reference processor_core_2.cc:181-189
// calls ExtractPhone1 and EstimatePitch1 back to back on the same 160 samples), so every launch of
// the pitch estimator rides along with a launch of the content encoder, and the waveform generator's
// conditioning mix (needs only the pitch head's output) rides with the next one:
//
//   phone:  f1   f2 f3 f4  f5  rb0 rb1  rb2 rb3 gru x H  out
//   pitch:  fft           p1  p2  p3   gru x H   out head  cond(wave)
//
// After f4 both chains have H + 6 launches left and pair one to one (H = 1: f5+p1, rb0+p2, rb1+p3,
// rb2+pitch.gru, rb3+pitch.out, phone.gru+pitch.head, phone.out+wave.cond): 11 launches instead of 20,
// identical arithmetic.  The pitch estimator's 768-thread first layer is paired with a layer that has few
// workgroups (the pair runs at the larger workgroup size; pairing it with phone.f2's 512 small
// workgroups cost 14.2 us against 7.4 + 8.6 separately).  Used by the batched path while the paired
// layers are in the few-row tiling (B x H <= 2048 rows); elsewhere the modules run one after the other
// (phone.hip, pitch.hip).
#include "chain_layers.hip.h"
#include "fuse.hip.h"

namespace bhip {


template <class LA, class LB>
static void pair_conv(const char* na, const ConvArgs& a, const char* nb, const ConvArgs& b, hipStream_t st) {
  using OA = ConvOp<LA, TLat<LA>>;
  using OB = ConvOp<LB, TLat<LB>>;
  fuse::launch(st, fuse::part<OA>(OA::info(na, a), a, OA::grid(a)), fuse::part<OB>(OB::info(nb, b), b, OB::grid(b)));
}

template <int H>
static void front_forward_h(const PhoneWeights& pw, const PhoneState& ps, const PitchWeights& qw, const PitchState& qs,
                            const WaveWeights& ww, const WaveState& ws, hipStream_t st) {
  using PL = PhoneLayers<H>;
  using QL = PitchLayers<H>;
  using RB = ConvOp<typename PL::RBL, TLat<typename PL::RBL>>;
  using POUT = ConvOp<typename QL::POUT, TLat<typename QL::POUT>>;
  using OUT = ConvOp<typename PL::OUTL, TLat<typename PL::OUTL>>;
  using PGRU = GruOp<256, 256>;
  using QGRU = GruOp<128, 128>;
  const int B = ps.B;

  fuse::launch(st, fuse::part<F1Op>(f1_info(ps), f1_args(pw, ps), dim3(B, H)), fuse::part<FftOp>(fft_info(qs), fft_args(qw, qs), dim3(B, H)));
  launch_auto<typename PL::F2>("phone.f2", conv_args(ps.f[0], ps.f[1], pw.f_w[0], pw.f_b[0], ps.hop, B), st);
  launch_auto<typename PL::F3>("phone.f3", conv_args(ps.f[1], ps.f[2], pw.f_w[1], pw.f_b[1], ps.hop, B), st);
  launch_auto<typename PL::F4>("phone.f4", conv_args(ps.f[2], ps.f[3], pw.f_w[2], pw.f_b[2], ps.hop, B), st);
  pair_conv<typename PL::F5, typename QL::P1>("phone.f5", conv_args(ps.f[3], ps.f[4], pw.f_w[3], pw.f_b[3], ps.hop, B),
                                              "pitch.p1", conv_args(qs.spec, qs.p[0], qw.p_w[0], qw.p_b[0], qs.hop, B), st);
  pair_conv<typename PL::RBL, typename QL::P23>("phone.rb", conv_args(ps.f[4], ps.rb[0], pw.rb_w[0], pw.rb_b[0], ps.hop, B),
                                                "pitch.p23", conv_args(qs.p[0], qs.p[1], qw.p_w[1], qw.p_b[1], qs.hop, B), st);
  pair_conv<typename PL::RBL, typename QL::P23>("phone.rb", conv_args(ps.rb[0], ps.rb[1], pw.rb_w[1], pw.rb_b[1], ps.hop, B),
                                                "pitch.p23", conv_args(qs.p[1], qs.p[2], qw.p_w[2], qw.p_b[2], qs.hop, B), st);

  // the remaining H + 3 launches of each chain, zipped:
  //   phone: rb2, rb3, gru(0..H-1), out          pitch: gru(0..H-1), out, head, cond
  const ConvArgs rb2 = conv_args(ps.rb[1], ps.rb[2], pw.rb_w[2], pw.rb_b[2], ps.hop, B);
  const ConvArgs rb3 = conv_args(ps.rb[2], ps.rb[3], pw.rb_w[3], pw.rb_b[3], ps.hop, B);
  const ConvArgs pout = conv_args(ps.h, phone_out_ring(ps), pw.out_w, pw.out_b, ps.hop, B);
  const ConvArgs qout = conv_args(qs.h, qs.logits, qw.out_w, qw.out_b, qs.hop, B);
  auto pgru = [&](int t) { return GruArgs{ps.rb[3], ps.h, pw.gru_wih, pw.gru_whh, pw.gru_bih, pw.gru_bhh, ps.hop, B, t}; };
  auto qgru = [&](int t) { return GruArgs{qs.p[2], qs.h, qw.gru_wih, qw.gru_whh, qw.gru_bih, qw.gru_bhh, qs.hop, B, t}; };
  for (int i = 0; i < H + 3; ++i) {
    // kinds at position i -- phone: 0 rb, 1 gru, 2 out;  pitch: 0 gru, 1 out, 2 head, 3 cond
    const int pk = i < 2 ? 0 : (i < 2 + H ? 1 : 2), qk = i < H ? 0 : i - H + 1;
    if (pk == 0 && qk == 0) {
      const ConvArgs& a = i == 0 ? rb2 : rb3;
      const GruArgs g = qgru(i);
      fuse::launch(st, fuse::part<RB>(RB::info("phone.rb", a), a, RB::grid(a)), fuse::part<QGRU>(QGRU::info("pitch.gru", g), g, QGRU::grid(g)));
    } else if (pk == 0 && qk == 1) {  // H = 1 only
      fuse::launch(st, fuse::part<RB>(RB::info("phone.rb", rb3), rb3, RB::grid(rb3)), fuse::part<POUT>(POUT::info("pitch.out", qout), qout, POUT::grid(qout)));
    } else if (pk == 1 && qk == 0) {  // H = 4 only
      const GruArgs a = pgru(i - 2), g = qgru(i);
      fuse::launch(st, fuse::part<PGRU>(PGRU::info("phone.gru", a), a, PGRU::grid(a)), fuse::part<QGRU>(QGRU::info("pitch.gru", g), g, QGRU::grid(g)));
    } else if (pk == 1 && qk == 1) {
      const GruArgs a = pgru(i - 2);
      fuse::launch(st, fuse::part<PGRU>(PGRU::info("phone.gru", a), a, PGRU::grid(a)), fuse::part<POUT>(POUT::info("pitch.out", qout), qout, POUT::grid(qout)));
    } else if (pk == 1 && qk == 2) {
      const GruArgs a = pgru(i - 2);
      fuse::launch(st, fuse::part<PGRU>(PGRU::info("phone.gru", a), a, PGRU::grid(a)), fuse::part<HeadOp>(head_info(qs), head_args(qw, qs), dim3(B, 1)));
    } else {  // pk == 2 && qk == 3
      fuse::launch(st, fuse::part<OUT>(OUT::info("phone.out", pout), pout, OUT::grid(pout)), fuse::part<CondOp>(cond_info(ws), cond_args(ww, ws), dim3(B * H, 1)));
    }
  }
  phone_vq(pw, ps, st);
}

bool front_forward(const PhoneWeights& pw, const PhoneState& ps, const PitchWeights& qw, const PitchState& qs,
                   const WaveWeights& ww, const WaveState& ws, hipStream_t st) {
  // paired regime: every PAIRED layer in the few-row tiling (launch_auto's rule, conv_gemm.hip.h)
  if (qs.H != ps.H || ws.H != ps.H || ps.B * ps.H > 2048 || qs.B != ps.B || ws.B != ps.B) return false;
  switch (ps.H) {
    case 1: front_forward_h<1>(pw, ps, qw, qs, ww, ws, st); break;
    case 2: front_forward_h<2>(pw, ps, qw, qs, ww, ws, st); break;
    case 4: front_forward_h<4>(pw, ps, qw, qs, ww, ws, st); break;
    default: front_forward_h<8>(pw, ps, qw, qs, ww, ws, st); break;
  }
  return true;
}

}  // namespace bhip
