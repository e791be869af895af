// batch_tick.hip.h -- the batch's tick mode: the workgroup table of one tick, the per-tick host work (settings snapshots, step pairs, launch), drain, enable / leave
// (Part of batch.hip's translation unit: included there, after struct BeatriceBatch and the helpers above it; not a stand-alone header.)
#pragma once

// ---- tick pipelining (tick.hip.h) ---------------------------------------------------------------------------------
#ifdef BEATRICE_HIP_MEASUREMENT_BUILD
// MEASUREMENT BUILDS ONLY (results are wrong by construction): the tick launch with only the body TYPES of this mask (tick::BodyType
// bit numbers), for per-body instruction counters (tools/debug/tick_inst_by_body.py); BeatriceBatchMeas_TickOnlyTypes below
static unsigned long long g_tick_only_types = ~0ull;
#endif
template <int H>
bool tick_build_table_h(BeatriceBatch* b, const bool sparse) {
  using namespace tick;
  using O = Ops<H>;
  using Tab = typename O::Tab;
  using OpF2 = typename O::OpF2; using OpF3 = typename O::OpF3; using OpF4 = typename O::OpF4; using OpF5 = typename O::OpF5; using OpRB = typename O::OpRB;
  using OpOUT = typename O::OpOUT; using OpP1 = typename O::OpP1; using OpP23 = typename O::OpP23; using OpPOUT = typename O::OpPOUT; using OpINP = typename O::OpINP;
  using OpUP1 = typename O::OpUP1; using OpRES1A = typename O::OpRES1A; using OpRES1B = typename O::OpRES1B; using OpUP2 = typename O::OpUP2;
  using OpF4s = typename O::OpF4s; using OpF5s = typename O::OpF5s; using OpRBs = typename O::OpRBs; using OpP1s = typename O::OpP1s; using OpUP1s = typename O::OpUP1s;
  using T1 = typename O::T1; using T2 = typename O::T2; using T3 = typename O::T3; using T1s = typename O::T1s; using T2s = typename O::T2s;
  using GruQ = typename O::GruQ; using GruP = typename O::GruP; using GruQ1 = typename O::GruQ1; using GruP1 = typename O::GruP1;
  using GruQm = typename O::GruQm; using GruPm = typename O::GruPm;
  State& k = b->tk;
  auto tb = std::make_unique<typename O::Builder>();
  const PhoneWeights& pw = b->phone_m->w;
  const PitchWeights& qw = b->pitch_m->w;
  const WaveWeights& ww = b->wave_m->w;
  const PhoneState& ps = b->phone;
  const PitchState& qs = b->pitch;
  const WaveState& ws = b->wave;
  const int B = b->B;
  const Plan pl = k.plan;
  // measurement aid, MEASUREMENT BUILDS ONLY (tools/debug/build_variant.sh <name> -DBEATRICE_HIP_MEASUREMENT_BUILD; the
  // product library has no switch that changes results): BEATRICE_HIP_TICK_DROP=<bit mask> leaves groups of bodies out of
  // the launch
#ifdef BEATRICE_HIP_MEASUREMENT_BUILD
  static const int drop = bhip::meas_env("BEATRICE_HIP_TICK_DROP") ? std::atoi(bhip::meas_env("BEATRICE_HIP_TICK_DROP")) : 0;
#else
  constexpr int drop = 0;
#endif
  auto keep = [](int group) { return ((drop >> group) & 1) == 0; };
#ifdef BEATRICE_HIP_MEASUREMENT_BUILD
  tb->only = g_tick_only_types;
#endif
  // (every body takes its step counter from the launch's StepPairs -- a null counter pointer says so, ring.h stepc)
  auto hp = [&](int) -> const int* { return nullptr; };
  auto conv = [&](const Ring& in, const Ring& out, const float* w, const float* bias, int stage) { return conv_args(in, out, w, bias, hp(stage), B); };
  // (workgroups are dispatched in this order: the longest-running bodies first)
  // The GRU cells of a step's hops, linked inside the launch (tick.hip.h): hop 0 publishes into link 0, hop t polls link t - 1 and
  // publishes into link t, the last hop only polls.  A cell must find its predecessor's states published when it starts (it holds a
  // slot while it waits), so the hops sit far apart in dispatch order: hop 0 ahead of every other body, the later ones at the points
  // marked gru_at() below, each behind at least one more round of the launch's workgroups.
  // (GruArgs::passes: two row groups per workgroup from 512 streams on; BEATRICE_HIP_GRU_PASSES overrides, for measurements)
  static const int gru_passes_env = bhip::meas_env("BEATRICE_HIP_GRU_PASSES") ? std::atoi(bhip::meas_env("BEATRICE_HIP_GRU_PASSES")) : 0;
  const int gru_passes = gru_passes_env > 0 ? gru_passes_env : (B >= 512 ? 2 : 1);
  auto link_p = [&](int t) { return k.d_link_p + (size_t)t * B * 256; };
  auto link_q = [&](int t) { return k.d_link_q + (size_t)t * B * 128; };
  auto add_pgru = [&](int t) {
    if constexpr (H > 1) {
      GruArgs g{ps.rb[3], ps.h, pw.gru_wih, pw.gru_whh, pw.gru_bih, pw.gru_bhh, hp(Plan::PGRU), B, t, t + 1 < H ? link_p(t) : nullptr, t > 0 ? link_p(t - 1) : nullptr, k.h_link_dead, gru_passes};
      if (t == 0) tb->template add<T_PGRU>(GruP::info("phone.gru", g), g, GruP::grid(g), Plan::PGRU, keep(1), 19);
      else if (t == H - 1) tb->template add<T_PGRU1>(GruP1::info("phone.gru", g), g, GruP1::grid(g), Plan::PGRU, keep(1), 19);
      else if constexpr (H > 2) tb->template add<T_PGRUM>(GruPm::info("phone.gru", g), g, GruPm::grid(g), Plan::PGRU, keep(1), 19);
    }
  };
  auto add_qgru = [&](int t) {
    if constexpr (H > 1) {
      GruArgs g{qs.p[2], qs.h, qw.gru_wih, qw.gru_whh, qw.gru_bih, qw.gru_bhh, hp(Plan::QGRU), B, t, t + 1 < H ? link_q(t) : nullptr, t > 0 ? link_q(t - 1) : nullptr, k.h_link_dead, gru_passes};
      if (t == 0) tb->template add<T_QGRU>(GruQ::info("pitch.gru", g), g, GruQ::grid(g), Plan::QGRU, keep(1), 12);
      else if (t == H - 1) tb->template add<T_QGRU1>(GruQ1::info("pitch.gru", g), g, GruQ1::grid(g), Plan::QGRU, keep(1), 12);
      else if constexpr (H > 2) tb->template add<T_QGRUM>(GruQm::info("pitch.gru", g), g, GruQm::grid(g), Plan::QGRU, keep(1), 12);
    }
  };
  // gru_at(point): the hops placed at insertion point 0 (behind the conditioned blocks), 1 (behind the tail), 2 / 3 (where the second
  // hop of a two-hop step has always been: behind phone.f2 / wave.inp).  H = 2: hop 1 at 2 (phone) and 3 (pitch); H = 4: hops 1, 2 at
  // 0, 1 and hop 3 at 2 / 3
  auto gru_at = [&](int point) {
    if constexpr (H == 2) { if (point == 2) add_pgru(1); if (point == 3) add_qgru(1); }
    if constexpr (H == 4) {
      if (point == 0) { add_pgru(1); add_qgru(1); }
      if (point == 1) { add_pgru(2); add_qgru(2); }
      if (point == 2) add_pgru(3);
      if (point == 3) add_qgru(3);
    }
  };
  add_pgru(0); add_qgru(0);
  // ---- longest workgroups first (measured, two per CU): f5 48 us, p1 46, f4 45, rb 42, block halves 41 / 38, tail 36
  if (!sparse) {
    { const ConvArgs a = conv(ps.f[3], ps.f[4], pw.f_w[3], pw.f_b[3], Plan::F5); tb->template add<T_F5>(OpF5::info("phone.f5", a), a, OpF5::grid(a), Plan::F5, keep(6), 30); }
    { const ConvArgs a = conv(qs.spec, qs.p[0], qw.p_w[0], qw.p_b[0], Plan::P1); tb->template add<T_P1>(OpP1::info("pitch.p1", a), a, OpP1::grid(a), Plan::P1, keep(2), 41); }
    { const ConvArgs a = conv(ps.f[2], ps.f[3], pw.f_w[2], pw.f_b[2], Plan::F4); tb->template add<T_F4>(OpF4::info("phone.f4", a), a, OpF4::grid(a), Plan::F4, keep(6), 28); }
    for (int i = 0; i < 4; ++i) {
      const ConvArgs a = conv(i == 0 ? ps.f[4] : ps.rb[i - 1], ps.rb[i], pw.rb_w[i], pw.rb_b[i], Plan::RB0 + i);
      tb->template add<T_RB>(OpRB::info("phone.rb", a), a, OpRB::grid(a), Plan::RB0 + i, keep(6), 39);
    }
  } else {   // (the sparse table: one row tile per workgroup, tick.hip.h)
    { const ConvArgs a = conv(ps.f[3], ps.f[4], pw.f_w[3], pw.f_b[3], Plan::F5); tb->template add<T_F5S>(OpF5s::info("phone.f5", a), a, OpF5s::grid(a), Plan::F5, keep(6), 30); }
    { const ConvArgs a = conv(qs.spec, qs.p[0], qw.p_w[0], qw.p_b[0], Plan::P1); tb->template add<T_P1S>(OpP1s::info("pitch.p1", a), a, OpP1s::grid(a), Plan::P1, keep(2), 41); }
    { const ConvArgs a = conv(ps.f[2], ps.f[3], pw.f_w[2], pw.f_b[2], Plan::F4); tb->template add<T_F4S>(OpF4s::info("phone.f4", a), a, OpF4s::grid(a), Plan::F4, keep(6), 28); }
    for (int i = 0; i < 4; ++i) {
      const ConvArgs a = conv(i == 0 ? ps.f[4] : ps.rb[i - 1], ps.rb[i], pw.rb_w[i], pw.rb_b[i], Plan::RB0 + i);
      tb->template add<T_RBS>(OpRBs::info("phone.rb", a), a, OpRBs::grid(a), Plan::RB0 + i, keep(6), 39);
    }
  }
  // conditioned blocks: two row-local chains each
  for (int blk = 0; blk < B_NBLOCKS; ++blk) {
    const WaveState::Scratch& sc = ws.scr[blk];  // one scratch set per block: all four blocks are in flight at once
    const int s0 = pl.blk(blk);
    const rc::BlockBArgs ba{sc.xa, ws.x[blk + 1], ww.q_w[blk], ww.q_b[blk], ww.o_w[blk], ww.o_b[blk], ws.d_kt[blk], ws.d_v[blk],
                            b->dev_view<int>(b->off.perm[blk]), b->dev_view<int>(b->off.tile_slot[blk]), hp(s0 + 1)};
    tb->template add<T_BLKB>(LaunchInfo{"wave.blk.b", 2.0 * B * H * (256.0 * 256 * 2 + 256.0 * 384 * 2), 4.0 * (2.0 * 256 * 256 + 2.0 * 256 * 384 + B * H * 3.0 * 256)}, ba,
                    dim3(ws.n_tiles_max, 1), s0 + 1, keep(5), 43.0);
    if (quads_on(b)) {  // rows without 15 neighbours on their K/V slot: one workgroup per quad (rebuild_tiles decides which rows)
      const rc::BlockBqArgs bq{sc.xa, ws.x[blk + 1], ww.q_w[blk], ww.q_b[blk], ww.o_w[blk], ww.o_b[blk], ws.d_ktp[blk], ws.d_vp[blk],
                               b->dev_view<int>(b->off.qperm[blk]), b->dev_view<int>(b->off.qslot[blk]), hp(s0 + 1)};
      // (a slot leaves at most 7 rows = 2 quads to this list: <= n_slots workgroups of two quads)
      tb->template add<T_BLKBQ>(LaunchInfo{"wave.blk.bq", 0.0, 0.0}, bq, dim3(std::min(2 * ws.n_tiles_max, ws.n_slots), 1), s0 + 1, keep(5), 42.0);
    }
  }
  for (int blk = 0; blk < B_NBLOCKS; ++blk) {
    const rc::BlockAArgs aa{ws.x[blk], ws.scr[blk].xa, ww.c1_w[blk], ww.c1_b[blk], ww.c2_w[blk], ww.c2_b[blk], hp(pl.blk(blk)), B};
    switch (blk) {
      case 0: tb->template add<T_BLKA1>(rc::BlockAOp<1, H>::info(aa), aa, rc::BlockAOp<1, H>::grid(aa), pl.blk(blk), keep(5), 38); break;
      case 1: tb->template add<T_BLKA2>(rc::BlockAOp<2, H>::info(aa), aa, rc::BlockAOp<2, H>::grid(aa), pl.blk(blk), keep(5), 38); break;
      case 2: tb->template add<T_BLKA4>(rc::BlockAOp<4, H>::info(aa), aa, rc::BlockAOp<4, H>::grid(aa), pl.blk(blk), keep(5), 38); break;
      default: tb->template add<T_BLKA8>(rc::BlockAOp<8, H>::info(aa), aa, rc::BlockAOp<8, H>::grid(aa), pl.blk(blk), keep(5), 38); break;
    }
  }
  gru_at(0);
  // (the k-NN lookup: one workgroup per stream walks the step's hops against the stream's codebook -- 35 us at four hops per step with
  //  64 speakers' codebooks in play.  At the end of the table, where round 2 had put it by its one-hop cost, it ENDED the launch: 256
  //  workgroups alone on the chip from 229 to 283 us of configs[3]'s tick, profiles/r06_notes.md section 3.  Present only while some stream has k > 0)
  { const VqArgs a{H, ps.raw, phone_vector_ring(ps), hp(Plan::VQ), ps.d_cbT, ps.d_cnorm, ps.d_vqk}; tb->template add<T_VQ>(LaunchInfo{"phone.vq", 0, 4.0 * B * H * 256}, a, dim3(B, 1), Plan::VQ, !ps.skip_vq, 9.0 * H, true); }
  if (!pl.split_tail) {
    TailArgs ta = tail_args(ww, ws); ta.hop = hp(pl.tail()); tb->template add<T_TAIL>(tail_info(ws), ta, dim3(B, 1), pl.tail(), keep(4), 41, true);
  } else {
    // the tail as three stages, several streams per workgroup (tail_stages.hip.h); same weights, same state block
    tst::StageArgs t1{}, t2{}, t3{};
    t1.in = ws.ya2; t1.out = ws.ya3; t2.in = ws.ya3; t2.out = ws.ya4; t3.in = ws.ya4;
    for (tst::StageArgs* t : {&t1, &t2, &t3}) { t->state = ws.tail.base; t->hop = nullptr; t->B = B; }
    t1.w[0] = ww.ra_w[1]; t1.b[0] = ww.ra_b[1]; t1.w[1] = ww.rb_w[1]; t1.b[1] = ww.rb_b[1]; t1.w[2] = ww.up_w[2]; t1.b[2] = ww.up_b[2];
    t2.w[0] = ww.ra_w[2]; t2.b[0] = ww.ra_b[2]; t2.w[1] = ww.rb_w[2]; t2.b[1] = ww.rb_b[2]; t2.w[2] = ww.up_w[3]; t2.b[2] = ww.up_b[3];
    t3.w[0] = ww.ra_w[3]; t3.b[0] = ww.ra_b[3]; t3.w[1] = ww.rb_w[3]; t3.b[1] = ww.rb_b[3];
    t3.fin_w = ww.fin_w; t3.fin_b = ww.fin_b; t3.d_out = ws.d_out; t3.io_stride = ws.io_stride;
    if (!sparse) {
      tb->template add<T_TAIL1>(T1::info(t1), t1, T1::grid(t1), pl.tail(), keep(4), 28, true);
      tb->template add<T_TAIL2>(T2::info(t2), t2, T2::grid(t2), pl.tail() + 1, keep(4), 11.5 * H, true);
    } else {   // (never occupied while the sparse table is in use; kept so that the two tables hold the same stages)
      tb->template add<T_TAIL1S>(T1s::info(t1), t1, T1s::grid(t1), pl.tail(), keep(4), 28, true);
      tb->template add<T_TAIL2S>(T2s::info(t2), t2, T2s::grid(t2), pl.tail() + 1, keep(4), 11.5 * H, true);
    }
    tb->template add<T_TAIL3>(T3::info(t3), t3, T3::grid(t3), pl.tail() + 2, keep(4), 6.5 * H, true);
  }
  gru_at(1);
  // ---- everything else, longest workgroups first (they start when the heavy ones above leave their slots)
  // (the pitch head: few workgroups that walk a step's hops one after the other -- 10 us per workgroup at two hops per step, 21 at four;
  //  at the end of the table, where its instruction count would put it, it ENDED the launch: round 5's timelines, profiles/r05_notes.md)
  { PitchHeadArgs a = head_args(qw, qs); a.hop = hp(Plan::HEAD); tb->template add<T_HEAD>(head_info(qs), a, dim3((B + 7) / 8, 1), Plan::HEAD, keep(0), 7.5 * H); }
  { const ConvArgs a = conv(ps.f[1], ps.f[2], pw.f_w[1], pw.f_b[1], Plan::F3); tb->template add<T_F3>(OpF3::info("phone.f3", a), a, OpF3::grid(a), Plan::F3, keep(6), 15); }
  if (!sparse) { const ConvArgs a = conv(ws.x[4], ws.ya1, ww.up_w[0], ww.up_b[0], pl.up1()); tb->template add<T_UP1>(OpUP1::info("wave.up1", a), a, OpUP1::grid(a), pl.up1(), keep(7), 14); }
  else { const ConvArgs a = conv(ws.x[4], ws.ya1, ww.up_w[0], ww.up_b[0], pl.up1()); tb->template add<T_UP1S>(OpUP1s::info("wave.up1", a), a, OpUP1s::grid(a), pl.up1(), keep(7), 14); }
  { const ConvArgs a = conv(ws.yb1, ws.yc1, ww.rb_w[0], ww.rb_b[0], pl.up1() + 2); tb->template add<T_RES1B>(OpRES1B::info("wave.res1b", a), a, OpRES1B::grid(a), pl.up1() + 2, keep(7), 13); }
  { const ConvArgs a = conv(ws.ya1, ws.yb1, ww.ra_w[0], ww.ra_b[0], pl.up1() + 1); tb->template add<T_RES1A>(OpRES1A::info("wave.res1a", a), a, OpRES1A::grid(a), pl.up1() + 1, keep(7), 13); }
  { const ConvArgs a = conv(ws.yc1, ws.ya2, ww.up_w[1], ww.up_b[1], pl.up1() + 3); tb->template add<T_UP2>(OpUP2::info("wave.up2", a), a, OpUP2::grid(a), pl.up1() + 3, keep(7), 13); }
  { const ConvArgs a = conv(ps.f[0], ps.f[1], pw.f_w[0], pw.f_b[0], Plan::F2); tb->template add<T_F2>(OpF2::info("phone.f2", a), a, OpF2::grid(a), Plan::F2, keep(6), 15); }
  if constexpr (H == 1) { const GruArgs g{ps.rb[3], ps.h, pw.gru_wih, pw.gru_whh, pw.gru_bih, pw.gru_bhh, hp(Plan::PGRU), B, 0, nullptr, nullptr, nullptr, gru_passes}; tb->template add<T_PGRU>(GruP::info("phone.gru", g), g, GruP::grid(g), Plan::PGRU, keep(1), 19); }
  gru_at(2);
  { const ConvArgs a = conv(qs.h, qs.logits, qw.out_w, qw.out_b, Plan::POUT); tb->template add<T_POUT>(OpPOUT::info("pitch.out", a), a, OpPOUT::grid(a), Plan::POUT, keep(2), 11); }
  for (int i = 0; i < 2; ++i) {
    const ConvArgs a = conv(qs.p[i], qs.p[i + 1], qw.p_w[i + 1], qw.p_b[i + 1], Plan::P2 + i);
    tb->template add<T_P23>(OpP23::info("pitch.p23", a), a, OpP23::grid(a), Plan::P2 + i, keep(2), 10);
  }
  { const Ring phone_in{ws.d_phone, B_PHONE_CH, H, ws.front_slots}; ConvArgs a = conv(phone_in, ws.x[0], ww.inp_w, ww.inp_b, Plan::INP); a.res = ws.e; tb->template add<T_INP>(OpINP::info("wave.inp", a), a, OpINP::grid(a), Plan::INP, keep(3), 7.5); }
  if constexpr (H == 1) { const GruArgs g{qs.p[2], qs.h, qw.gru_wih, qw.gru_whh, qw.gru_bih, qw.gru_bhh, hp(Plan::QGRU), B, 0, nullptr, nullptr, nullptr, gru_passes}; tb->template add<T_QGRU>(GruQ::info("pitch.gru", g), g, GruQ::grid(g), Plan::QGRU, keep(1), 12); }
  gru_at(3);
  { FftArgs a = fft_args(qw, qs); a.hop = hp(Plan::FFT); tb->template add<T_FFT>(fft_info(qs), FftArgs2{a, B}, dim3((B + 1) / 2, H), Plan::FFT, keep(0), 8.7); }
  { const ConvArgs a = conv(ps.h, phone_out_ring(ps), pw.out_w, pw.out_b, Plan::OUT); tb->template add<T_OUT>(OpOUT::info("phone.out", a), a, OpOUT::grid(a), Plan::OUT, keep(3), 4.5); }
  { F1Args a = f1_args(pw, ps); a.hop = hp(Plan::F1); a.hop_publish = nullptr; a.hop_publish_wave = nullptr; tb->template add<T_F1>(f1_info(ps), F1Args2{a, B}, dim3((B + 1) / 2, H), Plan::F1, keep(0), 3.7); }
  { CondArgs a = cond_args(ww, ws); a.hop = hp(Plan::COND); a.hop_next_out = nullptr; tb->template add<T_COND>(cond_info(ws), a, dim3((B * H + 1) / 2, 1), Plan::COND, keep(0), 1.5); }
  if (!tb->ok) return false;
  // (XCD-aware placement -- bodies with many weights pinned to one XCD, or to the XCDs of equal index modulo 2 / 4, so that their weights
  //  stay in fewer L2s -- was measured in rounds 2 and 4: memory-side traffic / 4, the launch 4-8 % slower; the switch went with the
  //  span lookup in round 5.  Its successor is TableBuilder::two_halves below: the same idea at the granularity that costs nothing)
  // Dispatch order by workgroup (fuse::WgDesc, one descriptor per workgroup read by a scalar load): the bodies in the order added above,
  // every body a contiguous run.  (Round 5 also measured the short workgroups dealt AMONG the long MFMA-dense ones: 7.7 % slower,
  // profiles/r05_notes.md section 4; round 6 the short ones as a KERNEL OF THEIR OWN at twice the occupancy, beside or behind the dense
  // one: 8-12 % slower, profiles/r06_notes.md section 1, tools/experiments/r06_two_kernel_tick.patch.)
  static const int halves_mode = bhip::meas_env("BEATRICE_HIP_TICK_HALVES") ? std::atoi(bhip::meas_env("BEATRICE_HIP_TICK_HALVES")) : 2;   // 0: every body on all XCDs; 2 (default): halves; 4: quarters
  {
    std::vector<fuse::WgDesc> desc;
    bool pinned[fuse::kMaxSpans];
    int group[fuse::kMaxSpans];
    for (int i = 0; i < tb->t.n_spans; ++i) {   // the bodies with the weights, each on one half of the chip (TableBuilder::two_halves)
      group[i] = -1;
      switch (tb->t.span[i].type) {
        case T_PGRU: case T_PGRU1: case T_PGRUM: pinned[i] = true; group[i] = 0; break;
        case T_QGRU: case T_QGRU1: case T_QGRUM: pinned[i] = true; group[i] = 1; break;
        case T_F2: case T_F3: case T_F4: case T_F5: case T_P1: case T_RB: case T_F4S: case T_F5S: case T_P1S: case T_RBS:
        case T_BLKA1: case T_BLKA2: case T_BLKA4: case T_BLKA8: case T_BLKB: case T_BLKBQ:
        case T_UP1: case T_UP1S: case T_RES1A: case T_RES1B: case T_UP2: case T_POUT: case T_P23: case T_INP: case T_OUT: pinned[i] = true; break;
        default: pinned[i] = false; break;   // (the tail stages and the per-stream bodies: hardly any weights)
      }
    }
    if (halves_mode != 0 && !sparse) {   // (the sparse table only ever runs partly filled ticks)
      desc = tb->two_halves(pinned, group, halves_mode == 4 ? 4 : 2);
    } else {
      desc = tb->in_span_order();
    }
    auto upload = [&](const std::vector<fuse::WgDesc>& v, fuse::WgDesc*& d_desc, size_t& cap) {
      if (v.size() > cap) {
        if (d_desc) (void)hipFree(d_desc);
        d_desc = nullptr;
        cap = 0;
        BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_desc), sizeof(fuse::WgDesc) * v.size()));
        cap = v.size();
      }
      if (!v.empty()) BHIP_TRY(hipMemcpy(d_desc, v.data(), sizeof(fuse::WgDesc) * v.size(), hipMemcpyHostToDevice));
      return true;
    };
    if (!sparse) {   // (the order of partly filled ticks: tick_launch)
      const std::vector<fuse::WgDesc> plain = tb->in_span_order();
      if (!upload(plain, k.d_desc_plain, k.desc_plain_cap)) return false;
      k.table_total_plain = (int)plain.size();
      // the fill / drain shapes without the empty stages' workgroups (tick::State::d_desc_ranges)
      for (auto& row : k.range_n) for (int& n : row) n = 0;
      const int n_stages = pl.count();
      // ... and, from kRangeHalvesFrom occupied stages on, in the halves order worked out for exactly the stages they hold (round 6: a
      // nearly full tick gains from the halves what a full one does; profiles/r06_notes.md section 9)
      static const int range_halves_from = bhip::meas_env("BEATRICE_HIP_TICK_RANGE_HALVES") ? std::atoi(bhip::meas_env("BEATRICE_HIP_TICK_RANGE_HALVES")) : kRangeHalvesFrom;
      if (plain.size() <= 16384 && n_stages <= kMaxStages) {
        std::vector<fuse::WgDesc> all;
        for (int shape = 0; shape < 2; ++shape)
          for (int s0 = 0; s0 < n_stages; ++s0) {
            k.range_off[shape][s0] = all.size();
            const int occupied = shape == 0 ? s0 + 1 : n_stages - s0;
            if (halves_mode != 0 && occupied >= range_halves_from) {
              const std::vector<fuse::WgDesc> part = shape == 0 ? tb->two_halves(pinned, group, 2, 0, s0) : tb->two_halves(pinned, group, 2, s0, 255);
              all.insert(all.end(), part.begin(), part.end());
            } else {
              for (const fuse::WgDesc& d : plain) {
                const int stage = (d.arg >> 16) & 0xff;
                if (shape == 0 ? stage <= s0 : stage >= s0) all.push_back(d);
              }
            }
            k.range_n[shape][s0] = (int)(all.size() - k.range_off[shape][s0]);
          }
        if (!upload(all, k.d_desc_ranges, k.desc_ranges_cap)) return false;
      }
    }
    tb->t.total = (int)desc.size();   // (two_halves may add filler indices)
    if (!upload(desc, sparse ? k.d_desc_sparse : k.d_desc, sparse ? k.desc_sparse_cap : k.desc_cap)) return false;
  }
  if (bhip::meas_env("BEATRICE_HIP_TICK_TRACE")) {
    if (k.d_trace) (void)hipFree(k.d_trace);
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&k.d_trace), sizeof(unsigned long long) * 3 * tb->t.total));
    tb->t.trace = k.d_trace;
  }
  if (sparse) {
    tb->t.trace = nullptr;
    BHIP_TRY(hipMemcpy(k.d_table_sparse, &tb->t, sizeof(Tab), hipMemcpyHostToDevice));
    k.table_sparse_total = tb->t.total;
    return true;
  }
  BHIP_TRY(hipMemcpy(k.d_table, &tb->t, sizeof(Tab), hipMemcpyHostToDevice));
  k.table_total = tb->t.total;
  k.table_flops = tb->flops;
  k.table_bytes = tb->bytes;
  // who reads which part of the settings block, and where its private copy lives
  k.consumers.clear();
  unsigned char* d = b->settings.d;
  k.consumers.push_back(Consumer{Plan::VQ, b->off.cbT, b->off.min_q - b->off.cbT, d + b->off.cbT, -1});
  k.consumers.push_back(Consumer{Plan::HEAD, b->off.min_q, b->off.add_idx - b->off.min_q, d + b->off.min_q, -1});
  k.consumers.push_back(Consumer{Plan::COND, b->off.add_idx, b->off.front_bytes - b->off.add_idx, d + b->off.add_idx, -1});
  for (int blk = 0; blk < B_NBLOCKS; ++blk) {
    const size_t lo = b->off.perm[blk], hi = blk + 1 < B_NBLOCKS ? b->off.perm[blk + 1] : b->off.front_bytes + b->off.wave_bytes;
    k.consumers.push_back(Consumer{pl.blk(blk) + 1, lo, hi - lo, d + lo, -1});
  }
  k.table_dirty = false;
  return true;
}
bool tick_build_table(BeatriceBatch* b, const bool sparse = false) {
  switch (b->H) {
    case 1: return tick_build_table_h<1>(b, sparse);
    case 2: return tick_build_table_h<2>(b, sparse);
    case 4: return tick_build_table_h<4>(b, sparse);
    default: return false;
  }
}
// the launch of one tick: the table kernel of the batch's hops per step
static void tick_launch(BeatriceBatch* b, const bool sparse, const bool full, hipStream_t st, const fuse::StepPairs& pairs) {
  tick::State& k = b->tk;
  const void* t = sparse ? k.d_table_sparse : k.d_table;
  // full = every stage has a step: the order that confines the bodies with the weights to halves of the chip; a partly filled tick
  // (fill, drain) runs the same table in plain span order -- without the workgroups of its empty stages where the occupied stages
  // are 0 .. k or k .. last (tick::State::d_desc_ranges)
  const fuse::WgDesc* desc = sparse ? k.d_desc_sparse : (full ? k.d_desc : k.d_desc_plain);
  int total = sparse ? k.table_sparse_total : (full ? k.table_total : k.table_total_plain);
  static const bool no_ranges = bhip::meas_env("BEATRICE_HIP_TICK_NO_RANGES") != nullptr;   // A/B switch for measurements
  if (!sparse && !full && !no_ranges && k.d_desc_ranges != nullptr) {
    const int n_stages = k.plan.count();
    int lo = n_stages, hi = -1, occupied = 0;
    for (int s = 0; s < n_stages; ++s) if (pairs.hop[s] >= 0) { lo = s < lo ? s : lo; hi = s; ++occupied; }
    if (occupied > 0 && occupied == hi - lo + 1) {
      const int shape = lo == 0 ? 0 : (hi == n_stages - 1 ? 1 : -1);
      const int at = shape == 0 ? hi : lo;
      if (shape >= 0 && k.range_n[shape][at] > 0) { desc = k.d_desc_ranges + k.range_off[shape][at]; total = k.range_n[shape][at]; }
    }
  }
  if (total <= 0) return;
  // (k.ragged: the second instance of the launch, once a stream has sat a step out -- at several hops per step a stream sits a WHOLE step out)
  if (b->H == 1) fuse::launch_table_w<4>(static_cast<const tick::Ops<1>::Tab*>(t), desc, total, st, pairs, k.ragged);
  else if (b->H == 2) fuse::launch_table_w<4>(static_cast<const tick::Ops<2>::Tab*>(t), desc, total, st, pairs, k.ragged);
  else fuse::launch_table_w<4>(static_cast<const tick::Ops<4>::Tab*>(t), desc, total, st, pairs, k.ragged);
}

// One tick: every stage advances by one step; `feeding` = a new step enters at stage 0.
// BEATRICE_HIP_TICK_HOSTPROF=1: host time of tick_run by section, printed when the process ends (measurement aid)
struct HostProf {
  static constexpr int N = 5;
  static bool on() { static const bool v = bhip::meas_env("BEATRICE_HIP_TICK_HOSTPROF") != nullptr; return v; }
  struct Totals { double us[N] = {}; long long calls = 0; ~Totals() { if (calls) std::fprintf(stderr, "tick_run host us per call: settings/kv %.1f, table %.1f, snapshot upload %.1f, copies %.1f, launch %.1f (%lld calls)\n", us[0] / calls, us[1] / calls, us[2] / calls, us[3] / calls, us[4] / calls, calls); } };
  static Totals& totals() { static Totals t; return t; }
  std::chrono::steady_clock::time_point t0;
  HostProf() { if (on()) { t0 = std::chrono::steady_clock::now(); totals().calls += 1; } }
  void lap(int i) {
    if (!on()) return;
    const auto t1 = std::chrono::steady_clock::now();
    totals().us[i] += std::chrono::duration<double, std::micro>(t1 - t0).count();
    t0 = t1;
  }
};
bool tick_run(BeatriceBatch* b, bool feeding) {
  using namespace tick;
  State& k = b->tk;
  HostProf prof;
  if (feeding) {
    advance_kv(b);
    draw_codebooks(b);
    if (b->vq_dirty) { update_vq_mode(b); b->vq_dirty = false; }   // (drains the pipeline if the k-NN stage comes or goes)
  }
  prof.lap(0);
  if (k.table_dirty && !(tick_build_table(b, true) && tick_build_table(b, false))) return false;   // (the full table last: it clears table_dirty)
  prof.lap(1);
  hipStream_t st = b->stream;
  Copy upload{nullptr, nullptr, 0};
  int upload_stage = -1;
  if (feeding) {
    bool dirty = b->front_dirty || k.snap_cur < 0;
    for (bool w : b->wave_dirty) dirty = dirty || w;
    if (dirty) {  // a new version of the settings: one upload into the next slot of the snapshot ring
      const int serial = k.snap_next++;
      const size_t off = 0, len = k.snap_bytes;
      unsigned char* dst = k.d_snap + (size_t)(serial % kRing) * k.snap_bytes;
      // through a ring of pinned staging copies, so that the host may run several settings changes ahead of the device
      // (the batch's two-deep mirror would make every second change wait for the copy of the change before it); the
      // tick's prologue kernel reads the staging copy straight from host memory (tick.hip.h)
      const int si = serial % State::kStaging;
      if (k.stage_pending[si]) { if (!hip_ok(hipEventSynchronize(k.stage_ev[si]), "tick settings staging")) return false; }
      unsigned char* src = k.h_stage + (size_t)si * k.snap_bytes;
      std::memcpy(src, b->settings.h + off, len);
      upload = Copy{dst, src, (int)len};
      upload_stage = si;
      k.snap_cur = serial;
      b->front_dirty = false;
      for (bool& w : b->wave_dirty) w = false;
    }
    const long long u = k.n_fed;
    k.fed_step[k.tick % kRing] = u;
    k.snap_of_step[u % kRing] = k.snap_cur;
    k.hop_of_step[u % kRing] = b->hop_host;
    k.io_of_step[u % kRing] = b->io_host;
  } else {
    k.fed_step[k.tick % kRing] = -1;
  }
  // ragged steps: the flags of BeatriceBatch_SetSilentStreams name the streams that sit THIS step out
  Copy hv_upload{nullptr, nullptr, 0};
  int hv_stage = -1;
  if (feeding) {
    BeatriceBatch::SilentRule& sr = b->silent;
    if (!k.ragged && sr.on && sr.any_next) {   // the first stream to stand still: from here on every stream has its own counter
      if (!k.d_hopv) {
        k.row = (b->B + 3) & ~3;
        if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&k.d_hopv), sizeof(int) * kRing * k.row), "tick hopv") ||
            !hip_ok(hipHostMalloc(reinterpret_cast<void**>(&k.h_hopv), sizeof(int) * State::kStaging * k.row, hipHostMallocDefault), "tick hopv staging"))
          return false;
        for (hipEvent_t& e : k.hv_ev) if (!hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "tick hopv event")) return false;
      }
      k.hop_s.assign(b->B, b->hop_host);
      k.ragged = true;
    }
    const long long u = k.n_fed;
    k.step_ragged[u % kRing] = k.ragged;
    if (k.ragged) {
      const int si = (int)(u % State::kStaging);
      if (k.hv_pending[si]) { if (!hip_ok(hipEventSynchronize(k.hv_ev[si]), "tick hopv staging")) return false; k.hv_pending[si] = false; }
      int* h = k.h_hopv + (size_t)si * k.row;
      for (int s = 0; s < b->B; ++s) {
        const bool out = sr.any_next && sr.next[s];
        h[s] = out ? -1 : k.hop_s[s];
        if (!out) k.hop_s[s] = hop_next(k.hop_s[s]);
      }
      for (int s = b->B; s < k.row; ++s) h[s] = -1;
      hv_upload = Copy{reinterpret_cast<unsigned char*>(k.d_hopv + (size_t)(u % kRing) * k.row), reinterpret_cast<const unsigned char*>(h), (int)(sizeof(int) * k.row)};
      hv_stage = si;
    }
    if (sr.any_next) { std::fill(sr.next.begin(), sr.next.end(), 0); sr.any_next = false; }
  }
  prof.lap(2);
  Prolog p{};
  p.n_stages = k.plan.count();
  auto step_at = [&k](int stage) -> long long {
    const long long t2 = k.tick - stage;
    return t2 >= 0 ? k.fed_step[t2 % kRing] : -1;
  };
  for (int s = 0; s < p.n_stages; ++s) {
    const long long u = step_at(s);
    p.hop[s] = u < 0 ? -1 : k.hop_of_step[u % kRing];
    p.io[s] = u < 0 ? 0 : k.io_of_step[u % kRing];
  }
  for (Consumer& c : k.consumers) {
    const long long u = step_at(c.stage);
    if (u < 0) continue;
    const int want = k.snap_of_step[u % kRing];
    if (want == c.held) continue;
    p.copy[p.n_copies++] = Copy{c.dst, k.d_snap + (size_t)(want % kRing) * k.snap_bytes + c.off, (int)c.bytes};
    c.held = want;
  }
  if (upload.bytes > 0) {  // first, so that entry order = age; (a consumer never needs the snapshot uploaded in its own tick: none sits at stage 0)
    for (int i = p.n_copies; i > 0; --i) p.copy[i] = p.copy[i - 1];
    p.copy[0] = upload;
    p.n_copies += 1;
  }
  if (hv_upload.bytes > 0) p.copy[p.n_copies++] = hv_upload;   // (read by stage 0 of the launch that follows: the prologue runs before it)
  if (p.n_copies > 0) {  // (only on ticks where the settings changed or a change arrives at a consumer)
    int chunks = 0;
    for (int i = 0; i < p.n_copies; ++i) { p.first_chunk[i] = chunks; chunks += (p.copy[i].bytes + kCopyChunk - 1) / kCopyChunk; }
    p.first_chunk[p.n_copies] = chunks;
    hipLaunchKernelGGL(prologue_kernel, dim3(chunks), dim3(256), 0, st, p);
    if (upload_stage >= 0) {
      if (!hip_ok(hipEventRecord(k.stage_ev[upload_stage], st), "tick settings event")) return false;
      k.stage_pending[upload_stage] = true;
    }
    if (hv_stage >= 0) {
      if (!hip_ok(hipEventRecord(k.hv_ev[hv_stage], st), "tick hopv event")) return false;
      k.hv_pending[hv_stage] = true;
    }
  }
  if (b->r48.on && (feeding || b->r48.deferred_slot >= 0)) {
    // one launch for both ends of the 48 kHz wrapper: the block entering the pipeline -> its 16 kHz hop, straight into the
    // resident slot; and the step the PREVIOUS tick completed leaves through the up-sampler and the 480-sample FIFO
    // (resample.h:346-361: the block emitted for step j carries the model output of step j - 1) into the 48 kHz slot of step j
    BeatriceBatch::Resident48& r = b->r48;
    Wrap48TickArgs wa{};
    wa.channels = r.channels; wa.st = b->d_w48; wa.coef_down = b->d_coef_down; wa.coef_up = b->d_coef_up; wa.H = b->H;
    auto counters_of = [&k](long long u) -> const int* { return u >= 0 && k.step_ragged[u % kRing] ? k.d_hopv + (size_t)(u % kRing) * k.row : nullptr; };
    if (feeding) {
      wa.n_pre = b->B;
      wa.in48 = r.d_in48 + (size_t)b->io_host * b->B * b->H * r.channels * 480;
      wa.in16 = r.d_in16 + (size_t)b->io_host * b->B * b->H * B_IN_HOP;
      wa.hv_pre = counters_of(k.n_fed);   // (the step being fed; its row was written by the prologue launch above)
    }
    if (r.deferred_slot >= 0) {
      wa.n_post = b->B;
      wa.out48 = r.d_out48 + (size_t)r.deferred_slot * b->B * b->H * r.channels * 480;
      wa.model_out = r.d_out24 + (size_t)r.deferred_slot * b->B * b->H * B_OUT_HOP;
      wa.hv_post = counters_of(r.deferred_step);
      wa.in48_post = r.d_in48 + (size_t)r.deferred_slot * b->B * b->H * r.channels * 480;
      r.deferred_slot = -1;
    }
    hipLaunchKernelGGL(wrap48_tick_kernel, dim3(wrap48_tick_grid(wa)), dim3(256), 0, st, wa);
  }
  prof.lap(3);
  fuse::StepPairs pairs;
  pairs.hopv = k.ragged ? k.d_hopv : nullptr;
  pairs.n_streams = k.row;
  for (int s = 0; s < fuse::kMaxStepPairs; ++s) {
    pairs.hop[s] = s < p.n_stages ? p.hop[s] : -1; pairs.io[s] = s < p.n_stages ? p.io[s] : 0;
    const long long u = s < p.n_stages ? step_at(s) : -1;
    pairs.hv[s] = u >= 0 && k.step_ragged[u % kRing] ? (int)(u % kRing) : -1;
  }
  // The sparse table (tick.hip.h) while only front-end stages have a step -- the first ticks of a fill: measured per tick on a
  // 20-step run (tools/debug/fill_drain.sh) 31 us against 42-45; from the first conditioned block on, and in the drain, the
  // half-size bodies LOSE (51-54 us against 45-48; two streams per tail workgroup 39 against 37), so there the full table runs
  int highest = -1;
  for (int s = 0; s < p.n_stages; ++s) if (p.hop[s] >= 0) highest = s;
  static const bool no_sparse = bhip::meas_env("BEATRICE_HIP_TICK_NO_SPARSE") != nullptr;   // A/B switch for measurements
  // (one hop per step only: at two hops per step it measured no difference, at four the half-size bodies cost 1 % of a 20-step run --
  //  5 196 against 5 138 us of launches, profiles/r05_notes.md)
  const bool sparse = highest >= 0 && highest < Plan::BLK0 && !no_sparse && b->H == 1;
  int occupied = 0;
  for (int s = 0; s < p.n_stages; ++s) occupied += p.hop[s] >= 0 ? 1 : 0;
  tick_launch(b, sparse, occupied == p.n_stages, st, pairs);
  if (b->r48.on) {  // the step this tick completed: its 48 kHz block is produced by the wrapper launch of the next tick (or of the drain)
    const long long u = step_at(k.plan.count() - 1);
    if (u >= 0) { b->r48.deferred_slot = k.io_of_step[u % kRing]; b->r48.deferred_step = u; }
  }
  if (feeding) {
    b->last_parity = b->hop_host % 3;
    b->last_hop = b->hop_host;
    b->hop_host = hop_next(b->hop_host);
    b->io_host = (b->io_host + 1) % b->io_slots;
    b->steps_enqueued += 1;
    k.n_fed += 1;
    k.last_feed_tick = k.tick;
  }
  prof.lap(4);
  k.tick += 1;
  b->inflight = true;
  return hip_ok(hipGetLastError(), "tick launch");
}
// Ragged steps leave every stream at a step counter of its own; at a DRAINED point (no step inside the pipeline) the batch is brought
// back to one counter: the rings of every stream that lags are rotated forward by its deficit (ring_rotate_kernel), after which all
// streams stand at the batch's counter -- the common launch runs again (the ragged instance costs +16 %), and the in-order chain,
// which has one counter for all streams, can take the batch over (leaving tick mode used to be refused for good).
static bool tick_relevel(BeatriceBatch* b) {
  using namespace tick;
  State& k = b->tk;
  if (!k.ragged) return true;
  std::vector<int> shift(b->B, 0);
  bool any = false;
  for (int s = 0; s < b->B; ++s) {
    int d = b->hop_host - k.hop_s[s];
    if (d < 0) d += B_HOP_WRAP;
    shift[s] = d;
    any = any || d != 0;
  }
  if (any) {
    if (!k.d_ring_table) {
      std::vector<FreezeRing> rings;
      for (const RingArena* a : {&b->phone.arena, &b->pitch.arena, &b->wave.arena})
        for (const Ring* r : a->rings) {
          if (r->m > kMaxRotateSlots) return false;
          rings.push_back(FreezeRing{r->base, r->n * r->C, r->m, 0});
        }
      k.n_ring_table = (int)rings.size();
      if (!hip_ok(hipMalloc(&k.d_ring_table, sizeof(FreezeRing) * rings.size()), "ring table") ||
          !hip_ok(hipMemcpy(k.d_ring_table, rings.data(), sizeof(FreezeRing) * rings.size(), hipMemcpyHostToDevice), "ring table up") ||
          !hip_ok(hipMalloc(reinterpret_cast<void**>(&k.d_shift), sizeof(int) * b->B), "ring shifts"))
        return false;
    }
    if (!hip_ok(hipMemcpyAsync(k.d_shift, shift.data(), sizeof(int) * b->B, hipMemcpyHostToDevice, b->stream), "ring shifts up")) return false;   // (pageable source: staged before the call returns)
    hipLaunchKernelGGL(ring_rotate_kernel, dim3(k.n_ring_table, b->B), dim3(256), 0, b->stream, static_cast<const FreezeRing*>(k.d_ring_table), k.d_shift);
    if (!hip_ok(hipGetLastError(), "ring rotate")) return false;
  }
  k.hop_s.assign(b->B, b->hop_host);
  k.ragged = false;
  for (bool& r : k.step_ragged) r = false;
  return true;
}
// ticks without new input until the last step fed has left the last stage
// output half of one call of the any-rate wrapper around the ticks (BeatriceBatch_BindResidentBlocks): its block's inner
// samples gathered from the resident model outputs, second resampling direction, output gain, into the call's slot
bool rb_post(BeatriceBatch* b, const BeatriceBatch::ResidentBlocks::Job& j) {
  BeatriceBatch::ResidentBlocks& r = b->rb;
  const size_t nt = b->wrap.taps_down.size();
  const int slot = (int)(j.call % r.n_slots), ge = (int)(j.call % r.ring);
  if (r.ragged)   // clocks per stream: the call's records name every stream's directions, sample count and block
    hipLaunchKernelGGL(wrapn::wrapr_post_kernel, dim3(b->B), dim3(256), 0, b->stream, r.d_out24, b->B, r.d_map, r.map_ring, b->d_wrap,
                       r.h_gains + (size_t)ge * 2 * b->B + b->B, b->rw.d_taps, r.h_rs + (size_t)ge * b->B,
                       r.d_out + (size_t)slot * b->B * r.cell, r.channels);
  else   // (the gain segments where the host wrote them: pinned memory)
    hipLaunchKernelGGL(wrapn::wrap_post_kernel, dim3(b->B), dim3(256), 0, b->stream, r.d_out24, r.io_slots, b->B, b->H, j.t0, b->d_wrap,
                       r.h_gains + (size_t)ge * 2 * b->B + b->B, b->d_wrap_taps + (j.dout.decimate ? 0 : nt), j.dout,
                       r.d_out + (size_t)slot * b->B * r.channels * r.n, r.channels);
  // (the event frees the ring entry for the host)
  if (!hip_ok(hipEventRecord(r.gain_ev[ge], b->stream), "wrapper gain event")) return false;
  r.ev_recorded[ge] = 1;
  return hip_ok(hipGetLastError(), "wrapper output half");
}
bool tick_drain(BeatriceBatch* b) {
  bool ok = true;
  if (b->tk.on && b->tk.d_trace && b->tk.last_feed_tick == b->tk.tick - 1 && b->tk.n_fed > b->tk.plan.count()) {
    // measurement aid: the tick just enqueued had every stage busy; dump its per-workgroup timeline (100 MHz wall clock)
    std::vector<unsigned long long> tr((size_t)3 * b->tk.table_total);
    if (hip_ok(hipStreamSynchronize(b->stream), "trace sync") &&
        hip_ok(hipMemcpy(tr.data(), b->tk.d_trace, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost), "trace copy"))
      if (FILE* f = std::fopen(bhip::meas_env("BEATRICE_HIP_TICK_TRACE"), "w")) {
        for (size_t i = 0; i < tr.size(); i += 3) std::fprintf(f, "%llu %llu %llu\n", tr[i], tr[i + 1], tr[i + 2]);
        std::fclose(f);
      }
  }
  while (ok && b->tk.on && b->tk.tick <= b->tk.last_feed_tick + b->tk.plan.count() - 1) ok = tick_run(b, false);
  if (ok && b->r48.on && b->r48.deferred_slot >= 0) {  // the 48 kHz block of the step the last tick completed
    BeatriceBatch::Resident48& r = b->r48;
    Wrap48TickArgs wa{};
    wa.channels = r.channels; wa.st = b->d_w48; wa.coef_down = b->d_coef_down; wa.coef_up = b->d_coef_up; wa.H = b->H;
    wa.n_post = b->B;
    wa.out48 = r.d_out48 + (size_t)r.deferred_slot * b->B * b->H * r.channels * 480;
    wa.model_out = r.d_out24 + (size_t)r.deferred_slot * b->B * b->H * B_OUT_HOP;
    wa.hv_post = r.deferred_step >= 0 && b->tk.step_ragged[r.deferred_step % tick::kRing] ? b->tk.d_hopv + (size_t)(r.deferred_step % tick::kRing) * b->tk.row : nullptr;
    wa.in48_post = r.d_in48 + (size_t)r.deferred_slot * b->B * b->H * r.channels * 480;
    r.deferred_slot = -1;
    hipLaunchKernelGGL(wrap48_tick_kernel, dim3(wrap48_tick_grid(wa)), dim3(256), 0, b->stream, wa);
    ok = hip_ok(hipGetLastError(), "wrap48 flush");
  }
  // every model hop that entered the pipeline has left it: the output halves still owed, in order.  With several hops per step
  // the hops of a step that is not full yet have not entered: the calls that end on them stay owed until later calls fill the
  // step (BeatriceBatch_ResidentBlocksOwed)
  while (ok && b->rb.on && !b->rb.jobs.empty() && b->rb.jobs.front().last_hop() < b->rb.hops_fed()) {
    ok = rb_post(b, b->rb.jobs.front());
    b->rb.jobs.pop_front();
  }
  if (ok && b->tk.on && b->tk.ragged) ok = tick_relevel(b);   // (nothing is in flight: back to one counter and the common launch)
  return ok;
}
int tick_enable(BeatriceBatch* b, bool on) {
  using namespace tick;
  State& k = b->tk;
  if (on == k.on) return 0;
  if (on) {
    // one 10 ms hop per step, resident I/O with enough slots
    // that a step's input is still there when the pitch head reads it nine ticks on and outputs have somewhere to land
    if ((b->H != 1 && b->H != 2 && b->H != 4) || b->B > 4096 || b->io_slots < k.plan.count() + 1 || b->io_slots > stepc::kImmediateMaxSlot + 1) return -1;   // (the slot index travels in 12 bits of an immediate, ring.h stepc)
    if (!sync_all(b)) return -2;
    if (b->pipelined) { drop_graph(b); set_plan(b, 1); }
    k.snap_bytes = b->off.front_bytes + b->off.wave_bytes;
    if (!k.d_table) {
      const size_t tab_bytes = std::max(std::max(sizeof(Ops<1>::Tab), sizeof(Ops<2>::Tab)), sizeof(Ops<4>::Tab));
      if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&k.d_table), tab_bytes), "tick table") ||
          !hip_ok(hipMalloc(reinterpret_cast<void**>(&k.d_table_sparse), tab_bytes), "tick sparse table") ||
          !hip_ok(hipMalloc(reinterpret_cast<void**>(&k.d_snap), k.snap_bytes * kRing), "tick snapshots") ||
          !hip_ok(hipHostMalloc(reinterpret_cast<void**>(&k.h_stage), k.snap_bytes * State::kStaging, hipHostMallocDefault), "tick staging"))
        return -2;
      for (hipEvent_t& e : k.stage_ev) if (!hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "tick staging event")) return -2;
    }
    if (b->H > 1 && !k.d_link_p) {   // the granules between the GRU cells of a step's hops (tick.hip.h); tag 0 = never written
      const size_t nq = (size_t)(b->H - 1) * b->B * 128, np = (size_t)(b->H - 1) * b->B * 256;
      if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&k.d_link_q), sizeof(unsigned long long) * nq), "tick gru link") ||
          !hip_ok(hipMalloc(reinterpret_cast<void**>(&k.d_link_p), sizeof(unsigned long long) * np), "tick gru link") ||
          !hip_ok(hipMemset(k.d_link_q, 0, sizeof(unsigned long long) * nq), "tick gru link") || !hip_ok(hipMemset(k.d_link_p, 0, sizeof(unsigned long long) * np), "tick gru link") ||
          !hip_ok(hipHostMalloc(reinterpret_cast<void**>(&k.h_link_dead), sizeof(int), hipHostMallocDefault), "tick gru link flag"))
        return -2;
      *k.h_link_dead = 0;
    }
    // the plain-order K / V copies the quad bodies read (rowchain.hip.h block_bq_body) exist in tick mode only: ~3 MB per
    // speaker that the in-order, stage-pipelined and large-batch modes never read
    if (!b->wave.legacy && !b->wave.d_ktp[0]) {
      const size_t kvf = (size_t)b->wave.n_slots * B_HID * B_KV_LEN;
      bool ok = true;
      for (int blk = 0; blk < B_NBLOCKS && ok; ++blk)
        ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&b->wave.d_ktp[blk]), sizeof(float) * kvf), "plain K") &&
             hip_ok(hipMalloc(reinterpret_cast<void**>(&b->wave.d_vp[blk]), sizeof(float) * kvf), "plain V") &&
             hip_ok(hipMemset(b->wave.d_ktp[blk], 0, sizeof(float) * kvf), "plain K zero") && hip_ok(hipMemset(b->wave.d_vp[blk], 0, sizeof(float) * kvf), "plain V zero");
      // every entry that may hold a table (real speakers and morph slots): re-projected from the raw embeddings the batch keeps
      if (ok && b->n_speakers > 0) ok = project_speakers(b, 0, b->max_speakers);
      if (!ok) {
        for (int blk = 0; blk < B_NBLOCKS; ++blk) {
          if (b->wave.d_ktp[blk]) (void)hipFree(b->wave.d_ktp[blk]);
          if (b->wave.d_vp[blk]) (void)hipFree(b->wave.d_vp[blk]);
          b->wave.d_ktp[blk] = b->wave.d_vp[blk] = nullptr;
        }
        return -2;
      }
    }
    if (!hip_ok(hipDeviceSynchronize(), "tick sync")) return -2;
    k.tick = 0; k.n_fed = 0; k.last_feed_tick = -1000; k.snap_cur = -1; k.snap_next = 0;
    k.ragged = false;
    for (bool& r : k.step_ragged) r = false;
    for (long long& f : k.fed_step) f = -1;
    k.table_dirty = true;
    k.on = true;
    for (int blk = 0; blk < B_NBLOCKS; ++blk) rebuild_tiles(b, blk);  // (tick mode cuts the attention rows into tiles AND quads)
    return 0;
  }
  if (!sync_all(b)) return -2;  // drains; streams that have sat steps out come back to the batch's counter (tick_relevel)
  if (k.ragged) return -2;
  k.on = false;
  if (b->silent.on && !b->silent.d_rings) b->silent.on = false;   // (the rule was enabled for tick mode only)
  for (int blk = 0; blk < B_NBLOCKS; ++blk) {   // (see above: tick mode's own copies of the K / V tables)
    if (b->wave.d_ktp[blk]) (void)hipFree(b->wave.d_ktp[blk]);
    if (b->wave.d_vp[blk]) (void)hipFree(b->wave.d_vp[blk]);
    b->wave.d_ktp[blk] = b->wave.d_vp[blk] = nullptr;
  }
  k.table_dirty = true;
  for (int blk = 0; blk < B_NBLOCKS; ++blk) rebuild_tiles(b, blk);
  // the in-order chain reads its counters from device memory: hand them the host's values
  const int pair[2] = {b->hop_host, b->io_host};
  if (!hip_ok(hipMemcpy(b->d_hop_next, pair, sizeof(pair), hipMemcpyHostToDevice), "tick leave")) return -2;
  b->front_dirty = true;
  for (bool& w : b->wave_dirty) w = true;
  return 0;
}

