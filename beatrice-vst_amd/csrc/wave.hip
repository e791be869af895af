// wave.hip -- waveform generator forward pass (MODEL_SPEC 4.4), the body of
// Beatrice20rc0_GenerateWaveform1 (reference lib/beatricelib/beatrice.h:301-307) for B streams.
#include <cstdlib>

#include "chain_layers.hip.h"
#include "team.hip.h"
#include "tail_stages.hip.h"  // the tail as three multi-stream kernels (large batches in order; the tick launch has them as bodies)
#include "rowchain.hip.h"

namespace bhip {

// a tail stage (tail_stages.hip.h) as a launch of its own
template <class Op>
static __global__ __launch_bounds__(tst::NTHR, 4) void tail_stage_kernel(const tst::StageArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[Op::LDS_FLOATS];
  Op::run(a, blockIdx.x, 0, lds);
}

static const int kBlockDil[B_NBLOCKS] = {1, 2, 4, 8};
static unsigned long long* g_team_trace = nullptr;   // (BEATRICE_HIP_TEAM_TRACE: one buffer for the process, leaked at exit)

bool WaveState::create(int B_, int H_, int n_slots_, int n_add_, int n_frm_, float* shared_phone, int* shared_q,
                       float* shared_feat, int front_slots_, bool pipe_slack, bool legacy_) {
  B = B_; H = H_; n_slots = n_slots_; n_add = n_add_; n_frm = n_frm_; front_slots = front_slots_; legacy = legacy_;
  const int ps = pipe_slack ? 1 : 0;  // every layer its own pipeline stage (tick mode): readers run a step behind
  boundary_slots = front_slots_ > 1 ? 2 : 0;  // a batch may cut the module into pipeline stages
  const int xs = boundary_slots;
  auto fit = [](int m) { while (B_HOP_WRAP % m != 0) ++m; return m; };
  const int rows = B * H;
  n_tiles_max = (rows + 15) / 16 + n_slots;  // rows grouped by slot: at most one partial tile per slot
  auto slots = [&](int n0, int hist) { return 1 + (hist + n0 * H - 1) / (n0 * H); };
  std::vector<RingSpec> specs = {
      {&e, B_HID, H, front_slots},
      {&x[0], B_HID, H, fit(slots(1, 2 * kBlockDil[0]) + xs)}, {&x[1], B_HID, H, fit(slots(1, 2 * kBlockDil[1]) + xs)},
      {&x[2], B_HID, H, fit(slots(1, 2 * kBlockDil[2]) + xs)}, {&x[3], B_HID, H, fit(slots(1, 2 * kBlockDil[3]) + xs)},
      {&x[4], B_HID, H, fit(slots(1, 1) + xs)},
  };
  for (int i = 0; i < (boundary_slots ? kScratchSets : 1); ++i) {
    // (xa is read again, as the residual, by the block's last layer: four stages after it is written)
    specs.push_back({&scr[i].h1, B_HID, H, 1 + ps}); specs.push_back({&scr[i].xa, B_HID, H, 1 + 4 * ps}); specs.push_back({&scr[i].q, B_HID, H, 1 + ps});
    specs.push_back({&scr[i].sc, B_KV_LEN, H, 1 + ps}); specs.push_back({&scr[i].o, B_HID, H, 1 + ps});
  }
  specs.push_back({&ya1, 128, 5 * H, slots(5, 2) + ps});   // history 2 (res1a, k3)
  specs.push_back({&yb1, 128, 5 * H, slots(5, 6) + ps});   // history 6 (res1b, k3 dil 3)
  specs.push_back({&yc1, 128, 5 * H, slots(5, 1) + ps});   // history 1 (up2)
  specs.push_back({&ya2, 64, 20 * H, fit(slots(20, 2) + xs)});  // history 2 (first layer of the fused tail)
  specs.push_back({&tail, TAIL_STATE_FLOATS, 1, 1});
  if (pipe_slack) {  // the tail as three pipeline stages (tail_stages.hip.h): writer and reader are one tick apart
    specs.push_back({&ya3, 32, 80 * H, 2});
    specs.push_back({&ya4, 16, 240 * H, 2});
  }
  if (!arena.build(B, specs)) return false;
  if (shared_phone) { d_phone = shared_phone; d_q = shared_q; d_feat = shared_feat; owns_inputs = false; }
  else {
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_phone), sizeof(float) * rows * B_PHONE_CH * front_slots));
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_q), sizeof(int) * rows));
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_feat), sizeof(float) * rows * 4));
    BHIP_TRY(hipMemset(d_phone, 0, sizeof(float) * rows * B_PHONE_CH * front_slots));
    BHIP_TRY(hipMemset(d_q, 0, sizeof(int) * rows));
    BHIP_TRY(hipMemset(d_feat, 0, sizeof(float) * rows * 4));
    owns_inputs = true;
  }
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), sizeof(float) * rows * B_OUT_HOP));
  BHIP_TRY(hipMemset(d_out, 0, sizeof(float) * rows * B_OUT_HOP));
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_add_tab), sizeof(float) * n_add * B_HID));
  BHIP_TRY(hipMemset(d_add_tab, 0, sizeof(float) * n_add * B_HID));
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_frm_tab), sizeof(float) * n_frm * B_HID));
  BHIP_TRY(hipMemset(d_frm_tab, 0, sizeof(float) * n_frm * B_HID));
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_add_idx), sizeof(int) * B));
  BHIP_TRY(hipMemset(d_add_idx, 0, sizeof(int) * B));
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_frm_idx), sizeof(int) * B));
  BHIP_TRY(hipMemset(d_frm_idx, 0, sizeof(int) * B));
  std::vector<int> perm((size_t)n_tiles_max * 16, -1), slot(n_tiles_max, -1);
  for (int r = 0; r < rows; ++r) perm[r] = r;
  for (int t = 0; t < (rows + 15) / 16; ++t) slot[t] = 0;
  for (int blk = 0; blk < B_NBLOCKS; ++blk) {
    const size_t kvf = (size_t)n_slots * B_HID * B_KV_LEN;
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_kt[blk]), sizeof(float) * kvf));
    BHIP_TRY(hipMemset(d_kt[blk], 0, sizeof(float) * kvf));
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_v[blk]), sizeof(float) * kvf));
    BHIP_TRY(hipMemset(d_v[blk], 0, sizeof(float) * kvf));
    // (the plain-order copies d_ktp / d_vp of these tables belong to tick mode only: the batch allocates and fills them when
    //  tick mode is entered and frees them when it is left, batch_tick.hip.h tick_enable)
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_perm[blk]), sizeof(int) * perm.size()));
    BHIP_TRY(hipMemcpy(d_perm[blk], perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice));
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_tile_slot[blk]), sizeof(int) * slot.size()));
    BHIP_TRY(hipMemcpy(d_tile_slot[blk], slot.data(), sizeof(int) * slot.size(), hipMemcpyHostToDevice));
  }
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_hop), 2 * sizeof(int)));  // [0] step counter, [1] resident-I/O slot
  BHIP_TRY(hipMemset(d_hop, 0, 2 * sizeof(int)));
  hop = d_hop;
  team_off = false;
  if (B == 1 && H == 1 && !legacy && hipFuncSetAttribute(reinterpret_cast<const void*>(team::wave_team_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, team::kLdsFloats * 4) == hipSuccess &&
      team_capacity_ok(reinterpret_cast<const void*>(team::wave_team_kernel), team::NWG, team::NTHR, team::kLdsFloats * 4)) {
    // the 1-stream ABI's team launch (team.hip.h), where the device can hold the whole team at once; tag 0 = "never written"
    team_granules = team::kWaveGranules;
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_team_xb), sizeof(unsigned long long) * team::kWaveGranules));
    BHIP_TRY(hipMemset(d_team_xb, 0, sizeof(unsigned long long) * team::kWaveGranules));
    // (pinned host memory, written by the kernel only when a wait was given up: the host reads it after every call for free)
    BHIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&d_team_dead), sizeof(int), hipHostMallocDefault));
    *d_team_dead = 0;
    if (bhip::meas_env("BEATRICE_HIP_TEAM_TRACE")) {   // measurement aid: per-stage stamps of workgroup 0 (BeatriceHip_TeamTraceDump)
      BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&g_team_trace), sizeof(unsigned long long) * 1024));
      BHIP_TRY(hipMemset(g_team_trace, 0, sizeof(unsigned long long) * 1024));
    }
  }
  // hipMemset is asynchronous and runs on the NULL stream, which the (non-blocking) compute streams
  // do not wait for: make every initialisation above visible before the first kernel can start
  BHIP_TRY(hipDeviceSynchronize());
  return true;
}
void WaveState::destroy() {
  arena.release();
  if (owns_inputs) { if (d_phone) (void)hipFree(d_phone); if (d_q) (void)hipFree(d_q); if (d_feat) (void)hipFree(d_feat); }
  if (d_team_xb) (void)hipFree(d_team_xb);
  if (d_team_dead) (void)hipHostFree(d_team_dead);
  d_team_xb = nullptr; d_team_dead = nullptr;
  void* ptrs[] = {d_out, d_add_tab, d_frm_tab, d_add_idx, d_frm_idx, d_hop};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (int b = 0; b < B_NBLOCKS; ++b) {
    void* q4[] = {d_kt[b], d_v[b], d_perm[b], d_tile_slot[b], d_ktp[b], d_vp[b]};
    for (void* p : q4) if (p) (void)hipFree(p);
    d_kt[b] = d_v[b] = d_ktp[b] = d_vp[b] = nullptr; d_perm[b] = d_tile_slot[b] = nullptr;
  }
  d_phone = d_feat = d_out = d_add_tab = d_frm_tab = nullptr;
  d_q = d_add_idx = d_frm_idx = d_hop = nullptr;
}

using namespace wave_layers;

#define MISC_LAUNCH(NAME, FLOPS, BYTES, KERNEL, GRID, BLOCK, ...)                              \
  launch_site(LaunchInfo{NAME, (double)(FLOPS), (double)(BYTES)}, st,                          \
              [&] { hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, st, __VA_ARGS__); })

template <int D, int H>
static void launch_c1(const WaveWeights& w, const WaveState& s, int blk, const Ring& h1, hipStream_t st) {
  static const char* const names[9] = {"", "wave.blk.c1.d1", "wave.blk.c1.d2", "", "wave.blk.c1.d4", "", "", "", "wave.blk.c1.d8"};
  launch_auto<C1<D, H>>(names[D], conv_args(s.x[blk], h1, w.c1_w[blk], w.c1_b[blk], s.hop, s.B), st);
}

// One stream, one hop (the 1-stream C-ABI): input mix, the four conditioned blocks and upsampler stage 1 + the stage-2 transposed
// conv -- 29 layers -- as ONE launch of a team of workgroups (team.hip.h).  BEATRICE_HIP_NO_TEAM=1: the per-layer launches (A/B, parity).
static bool team_on() {
  static const bool off = bhip::meas_env("BEATRICE_HIP_NO_TEAM") != nullptr;
  return !off;
}
static void launch_wave_team(const WaveWeights& w, const WaveState& s, hipStream_t st, const CondArgs* cond) {
  using namespace team;
  const WaveState::Scratch& k = s.scr[0];
  WaveTeamArgs a{};
  gran_t* g = s.d_team_xb;
  auto take = [&g](size_t n) { gran_t* p = g; g += n; return p; };
  a.phone_in = Ring{s.d_phone, B_PHONE_CH, 1, s.front_slots};
  a.e = s.e;
  for (int i = 0; i <= B_NBLOCKS; ++i) a.x[i] = Tensor{s.x[i], take(256)};
  for (int b = 0; b < B_NBLOCKS; ++b) {
    a.h1[b] = Tensor{k.h1, take(256)}; a.xa[b] = Tensor{k.xa, take(256)}; a.q[b] = Tensor{k.q, take(256)}; a.o[b] = Tensor{k.o, take(256)};
    a.sc[b] = Tensor{k.sc, take(384)};
    a.c1_w[b] = w.c1_w[b]; a.c1_b[b] = w.c1_b[b]; a.c2_w[b] = w.c2_w[b]; a.c2_b[b] = w.c2_b[b];
    a.q_w[b] = w.q_w[b]; a.q_b[b] = w.q_b[b]; a.o_w[b] = w.o_w[b]; a.o_b[b] = w.o_b[b];
    a.kt[b] = s.d_kt[b]; a.v[b] = s.d_v[b]; a.tile_slot[b] = s.d_tile_slot[b];
  }
  a.ya1 = Tensor{s.ya1, take(640)}; a.yb1 = Tensor{s.yb1, take(640)}; a.yc1 = Tensor{s.yc1, take(640)}; a.ya2 = Tensor{s.ya2, nullptr};
  a.inp_w = w.inp_w; a.inp_b = w.inp_b;
  a.up_w[0] = w.up_w[0]; a.up_b[0] = w.up_b[0]; a.up_w[1] = w.up_w[1]; a.up_b[1] = w.up_b[1];
  a.ra_w = w.ra_w[0]; a.ra_b = w.ra_b[0]; a.rb_w = w.rb_w[0]; a.rb_b = w.rb_b[0];
  a.hop = s.hop;
  a.dead = s.d_team_dead;
  a.stamps = g_team_trace;
  if (cond != nullptr) { a.with_cond = 1; a.cond = *cond; }   // (the conditioning mix at the head of the launch)
  launch_site(LaunchInfo{"wave.team", 2.0 * (128.0 * 256 + 4 * (768.0 * 256 + 3 * 256.0 * 256 + 2 * 256.0 * 384) + 512.0 * 640 + 2 * 5 * 384.0 * 128 + 5 * 256.0 * 256),
                         4.0 * (128.0 * 256 + 4 * (768.0 * 256 + 3 * 256.0 * 256 + 2 * 256.0 * 384) + 512.0 * 640 + 2 * 384.0 * 128 + 256.0 * 256)},
              st, [&] { hipLaunchKernelGGL(wave_team_kernel, dim3(NWG), dim3(NTHR), kLdsFloats * 4, st, a); });
}

template <int H>
static void wave_forward_h(const WaveWeights& w, const WaveState& s, hipStream_t st, bool cond_done, WavePart part) {
  const int B = s.B, rows = s.B * H;
  const WaveState::Scratch& k = s.scr[part.scratch];
  auto in_part = [&part](int seg) { return seg >= part.first && seg <= part.last; };
  ConvArgs a;
  const bool use_team = H == 1 && B == 1 && s.d_team_xb != nullptr && !s.team_off && part.first <= 1 && part.last >= 6 && team_on();
  const CondArgs ca = cond_args(w, s);
  static const bool no_team_head = bhip::meas_env("BEATRICE_HIP_NO_TEAM_HEAD") != nullptr;   // A/B switch: wave.cond as a launch of its own again
  // the 1-stream ABI's contexts (one counter, no pair to hand on): the conditioning mix runs at the head of the team launch
  const bool cond_in_team = use_team && !cond_done && !no_team_head && ca.hop == s.hop && ca.hop_next_out == nullptr;
  if (!cond_done && !cond_in_team) launch_site(cond_info(s), st, [&] { hipLaunchKernelGGL(wave_cond_kernel, dim3(rows), dim3(256), 0, st, ca); });
  if (use_team) {
    launch_wave_team(w, s, st, cond_in_team ? &ca : nullptr);
    part.first = 7;   // what is left: the fused tail
  }
  if (in_part(1)) {
    const Ring phone_in{s.d_phone, s.legacy ? 256 : B_PHONE_CH, H, s.front_slots};
    a = conv_args(phone_in, s.x[0], w.inp_w, w.inp_b, s.hop, B);
    a.res = s.e;
    if (s.legacy) launch_auto<Layer<256, B_HID, 1, 1, 1, H, PRE_NONE, ACT_NONE, EPI_BIAS, true>>("wave.inp", a, st);
    else launch_auto<INP<H>>("wave.inp", a, st);
  }
  // The conditioned blocks as two stream-stationary kernels each (rowchain.hip.h) instead of six per-layer launches: the
  // per-layer launches win while a launch cannot fill the chip (256 streams: 0.293 vs 0.413 ms per step), the row-local
  // kernels once 16 streams per workgroup do (8192 streams: 84-89 TFLOP/s per block half against 34-61 for the six
  // layers; 3.40 -> 3.82 M frames/s); even at 2048.  BEATRICE_HIP_ROWCHAIN=1 / =0 forces one or the other (measurements, parity tests).
  static const char* const rc_env = bhip::meas_env("BEATRICE_HIP_ROWCHAIN");
  const bool rowchain = rc_env != nullptr ? rc_env[0] != '0' : B >= 2048;
  for (int blk = 0; blk < B_NBLOCKS; ++blk) {
    if (!in_part(2 + blk)) continue;
    if (rowchain && H == 1 && !s.legacy) {
      const rc::BlockAArgs aa{s.x[blk], k.xa, w.c1_w[blk], w.c1_b[blk], w.c2_w[blk], w.c2_b[blk], s.hop, B};
      const dim3 ga((B + 15) / 16);
      launch_site(rc::BlockAOp<1>::info(aa), st, [&] {
        switch (blk) {
          case 0: hipLaunchKernelGGL(rc::block_a_kernel<1>, ga, dim3(rc::NTHR), 0, st, aa); break;
          case 1: hipLaunchKernelGGL(rc::block_a_kernel<2>, ga, dim3(rc::NTHR), 0, st, aa); break;
          case 2: hipLaunchKernelGGL(rc::block_a_kernel<4>, ga, dim3(rc::NTHR), 0, st, aa); break;
          default: hipLaunchKernelGGL(rc::block_a_kernel<8>, ga, dim3(rc::NTHR), 0, st, aa); break;
        }
      });
      const rc::BlockBArgs ba{k.xa, s.x[blk + 1], w.q_w[blk], w.q_b[blk], w.o_w[blk], w.o_b[blk], s.d_kt[blk], s.d_v[blk],
                              s.d_perm[blk], s.d_tile_slot[blk], s.hop};
      launch_site(LaunchInfo{"wave.blk.b", 2.0 * rows * (256.0 * 256 * 2 + 256.0 * 384 * 2), 4.0 * (2.0 * 256 * 256 + 2.0 * 256 * 384 + rows * 3.0 * 256)}, st,
                  [&] { hipLaunchKernelGGL(rc::block_b_kernel, dim3(s.n_tiles_max), dim3(rc::NTHR), 0, st, ba); });
      continue;
    }
    switch (blk) {
      case 0: launch_c1<1, H>(w, s, blk, k.h1, st); break;
      case 1: launch_c1<2, H>(w, s, blk, k.h1, st); break;
      case 2: launch_c1<4, H>(w, s, blk, k.h1, st); break;
      default: launch_c1<8, H>(w, s, blk, k.h1, st); break;
    }
    if (s.legacy) {  // MODEL_SPEC 6.3: the block ends here -- x' = x + Linear(h), no attention half
      a = conv_args(k.h1, s.x[blk + 1], w.c2_w[blk], w.c2_b[blk], s.hop, B);
      a.res = s.x[blk];
      launch_auto<C2<H>>("wave.blk.c2o", a, st);
      continue;
    }
    a = conv_args(k.h1, k.xa, w.c2_w[blk], w.c2_b[blk], s.hop, B);
    a.res = s.x[blk];
    launch_auto<C2<H>>("wave.blk.c2o", a, st);  // c2 and o are the same kernel symbol: one profile row
    launch_auto<QL<H>>("wave.blk.q", conv_args(k.xa, k.q, w.q_w[blk], w.q_b[blk], s.hop, B), st);
    a = conv_args(k.q, k.sc, s.d_kt[blk], nullptr, s.hop, B);
    a.scale = 0.0625f; a.perm = s.d_perm[blk]; a.tile_slot = s.d_tile_slot[blk];
    a.w_slot_stride = (size_t)B_HID * B_KV_LEN;
    launch_conv<SCORE<H>, TGQ>("wave.blk.attn_qk", a, s.n_tiles_max, st);
    AttnPvArgs pa{k.sc, s.d_v[blk], k.o, s.d_perm[blk], s.d_tile_slot[blk], s.hop};
    MISC_LAUNCH("wave.blk.attn_pv", 2.0 * rows * 384 * 256 + 25.0 * rows * 384, 4.0 * (384.0 * 256 + rows * (384 + 256)),
                attn_pv_kernel, dim3(s.n_tiles_max, B_HID / 32), dim3(256), pa);
    a = conv_args(k.o, s.x[blk + 1], w.o_w[blk], w.o_b[blk], s.hop, B);
    a.res = k.xa;
    launch_auto<C2<H>>("wave.blk.c2o", a, st);
  }
  // upsampler: stage 1 and the stage-2 transposed conv as batched GEMMs (few rows per stream, large
  // weights), everything after that in one per-stream kernel
  if (in_part(6)) {
    launch_auto<UP<256, 128, 5, H>>("wave.up1", conv_args(s.x[4], s.ya1, w.up_w[0], w.up_b[0], s.hop, B), st);
    launch_auto<RES<128, 1, 5 * H>>("wave.res1a", conv_args(s.ya1, s.yb1, w.ra_w[0], w.ra_b[0], s.hop, B), st);
    launch_auto<RES<128, 3, 5 * H>>("wave.res1b", conv_args(s.yb1, s.yc1, w.rb_w[0], w.rb_b[0], s.hop, B), st);
    launch_auto<UP<128, 64, 4, 5 * H>>("wave.up2", conv_args(s.yc1, s.ya2, w.up_w[1], w.up_b[1], s.hop, B), st);
  }
  if (!in_part(7)) return;
  if (rowchain && H == 1 && s.ya3.base != nullptr) {
    // many streams: the three multi-stream tail stages (tail_stages.hip.h; rows = (stream, frame) fill the MFMA tiles, a third of
    // the fused kernel's VALU work) as launches of their own -- the same state block, so a batch may change between this,
    // the fused kernel and the tick launch at any step
    tst::StageArgs t1{}, t2{}, t3{};
    t1.in = s.ya2; t1.out = s.ya3; t2.in = s.ya3; t2.out = s.ya4; t3.in = s.ya4;
    for (tst::StageArgs* t : {&t1, &t2, &t3}) { t->state = s.tail.base; t->hop = s.hop; t->B = B; }
    t1.w[0] = w.ra_w[1]; t1.b[0] = w.ra_b[1]; t1.w[1] = w.rb_w[1]; t1.b[1] = w.rb_b[1]; t1.w[2] = w.up_w[2]; t1.b[2] = w.up_b[2];
    t2.w[0] = w.ra_w[2]; t2.b[0] = w.ra_b[2]; t2.w[1] = w.rb_w[2]; t2.b[1] = w.rb_b[2]; t2.w[2] = w.up_w[3]; t2.b[2] = w.up_b[3];
    t3.w[0] = w.ra_w[3]; t3.b[0] = w.ra_b[3]; t3.w[1] = w.rb_w[3]; t3.b[1] = w.rb_b[3];
    t3.fin_w = w.fin_w; t3.fin_b = w.fin_b; t3.d_out = s.d_out; t3.io_stride = s.io_stride;
    launch_site(tst::T1Op::info(t1), st, [&] { hipLaunchKernelGGL(tail_stage_kernel<tst::T1Op>, tst::T1Op::grid(t1), dim3(tst::NTHR), 0, st, t1); });
    launch_site(tst::T2Op::info(t2), st, [&] { hipLaunchKernelGGL(tail_stage_kernel<tst::T2Op>, tst::T2Op::grid(t2), dim3(tst::NTHR), 0, st, t2); });
    launch_site(tst::T3Op::info(t3), st, [&] { hipLaunchKernelGGL(tail_stage_kernel<tst::T3Op>, tst::T3Op::grid(t3), dim3(tst::NTHR), 0, st, t3); });
  } else {
    const TailArgs ta = tail_args(w, s);
    launch_site(tail_info(s), st, [&] { hipLaunchKernelGGL(wave_tail_kernel<H>, dim3(B), dim3(tail::NTHR), 0, st, ta); });
  }
  if (s.advance_hop) MISC_LAUNCH("hop_advance", 0, 4, hop_advance_kernel, dim3(1), dim3(1), s.hop);
}

// the conditioning mix alone (a batch runs it with the front end of the step)
void wave_cond(const WaveWeights& w, const WaveState& s, hipStream_t st) {
  const CondArgs ca = cond_args(w, s);
  launch_site(cond_info(s), st, [&] { hipLaunchKernelGGL(wave_cond_kernel, dim3(s.B * s.H), dim3(256), 0, st, ca); });
}

// measurement aid: the stamps of the last team launch (see team.hip.h)
extern "C" int BeatriceHip_TeamTraceDump(unsigned long long* out, int cap) {
  if (!g_team_trace || hipDeviceSynchronize() != hipSuccess) return -1;
  unsigned long long all[1024];
  if (hipMemcpy(all, g_team_trace, sizeof(all), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  int n = (int)all[1023];
  n = n < cap ? n : cap;
  for (int i = 0; i < n; ++i) out[i] = all[i];
  return n;
}

void wave_forward(const WaveWeights& w, const WaveState& s, hipStream_t st, bool cond_done, WavePart part) {
  switch (s.H) {
    case 1: wave_forward_h<1>(w, s, st, cond_done, part); break;
    case 2: wave_forward_h<2>(w, s, st, cond_done, part); break;
    case 4: wave_forward_h<4>(w, s, st, cond_done, part); break;
    default: wave_forward_h<8>(w, s, st, cond_done, part); break;
  }
}

}  // namespace bhip
