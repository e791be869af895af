// batch_wrappers.hip.h -- the host-side wrappers of the batch API: 48 kHz blocks (in order and around the ticks), any host rate with gains (in order and around the ticks)
// (Part of batch.hip's translation unit: included there, after struct BeatriceBatch and the helpers above it; not a stand-alone header.)
#pragma once

// ---- 48 kHz blocks with the wrapper on the device ----------------------------------------------
static bool step_48k(BeatriceBatch* b, const float* d_in48, float* d_out48, int channels) {
  BeatriceBatch::SilentRule& sr = b->silent;
  const unsigned char* frozen = nullptr;
  int flag_slot = -1;
  if (sr.on && sr.any_next) {   // this step's flags travel to the device through a ring of staging copies; a slot is reused only
    const int e = (int)(sr.steps % BeatriceBatch::SilentRule::kDepth);   // after the step that read it has finished (its event)
    if (sr.flag_pending[e]) { BHIP_TRY(hipEventSynchronize(sr.flag_ev[e])); sr.flag_pending[e] = false; }
    unsigned char* h = sr.h_flags + (size_t)e * b->B;
    std::memcpy(h, sr.next.data(), b->B);
    unsigned char* d = sr.d_flags + (size_t)e * b->B;
    BHIP_TRY(hipMemcpyAsync(d, h, b->B, hipMemcpyHostToDevice, b->stream));
    frozen = d;
    flag_slot = e;
  }
  if (sr.on) ++sr.steps;
  // the FIFO of the reference emits the PREVIOUS block's model output first (resample.h:346-361)
  hipLaunchKernelGGL(wrap48_post_kernel, dim3(b->B), dim3(256), 0, b->stream, b->d_w48, b->d_coef_up, d_out48, channels, frozen, d_in48);
  hipLaunchKernelGGL(wrap48_pre_kernel, dim3(b->B), dim3(256), 0, b->stream, d_in48, channels, b->d_w48, b->d_coef_down, b->d_in, frozen);
  if (frozen)
    hipLaunchKernelGGL(freeze_save_kernel, dim3(sr.n_rings, b->B), dim3(256), 0, b->stream, sr.d_rings, sr.d_keep, b->B, b->pitch.d_prev_q, sr.d_keep_prev_q);
  sr.in_block_step = true;
  const bool ok = step_device(b, nullptr, nullptr);   // (advance_kv / draw_codebooks skip the flagged streams)
  sr.in_block_step = false;
  if (frozen) {
    std::fill(sr.next.begin(), sr.next.end(), 0);
    sr.any_next = false;
  }
  if (!ok) return false;
  if (frozen)
    hipLaunchKernelGGL(freeze_fix_kernel, dim3(sr.n_rings, b->B), dim3(256), 0, b->stream, sr.d_rings, sr.d_keep, b->B, frozen, b->last_hop,
                       b->pitch.d_prev_q, sr.d_keep_prev_q);
  hipLaunchKernelGGL(wrap48_latch_kernel, dim3((b->B * 240 + 255) / 256), dim3(256), 0, b->stream, b->d_w48, b->wave.d_out, b->B, frozen);
  if (flag_slot >= 0) { BHIP_TRY(hipEventRecord(sr.flag_ev[flag_slot], b->stream)); sr.flag_pending[flag_slot] = true; }
  return hip_ok(hipGetLastError(), "wrap48");
}
static bool freeze_prepare(BeatriceBatch* b);
static void silent_release(BeatriceBatch* b) {
  BeatriceBatch::SilentRule& sr = b->silent;
  for (hipEvent_t& e : sr.flag_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  if (sr.d_flags) (void)hipFree(sr.d_flags);
  if (sr.h_flags) (void)hipHostFree(sr.h_flags);
  if (sr.d_rings) (void)hipFree(sr.d_rings);
  if (sr.d_keep) (void)hipFree(sr.d_keep);
  if (sr.d_keep_prev_q) (void)hipFree(sr.d_keep_prev_q);
  sr = BeatriceBatch::SilentRule{};
}
// The shell's rule "a block whose down-mix is all zeros is not converted" (reference src/vst/processor.cc:204-214), per stream,
// for the in-order 48 kHz blocks: BeatriceBatch_ConvertBlocks48k (host buffers) finds the silent streams itself, exactly as the
// shell does (every sample of (L + R) * 0.5, or of L, equal to 0.0f); with device buffers the caller names them for the next
// block with BeatriceBatch_SetSilentStreams.  A silent stream's output block is its (zero) down-mix on every channel; its model
// state, its wrapper state (filter histories, the pending 10 ms of the FIFO), its pending key/value installs and its codebook
// lottery stand still, as if the block had never existed.
int BeatriceBatch_EnableSilentBlockRule(BeatriceBatch* b, int enable) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::SilentRule& sr = b->silent;
  // a resident-block binding around the ticks owns the rule's flags while it is bound (BindResidentBlocksRagged names the streams
  // whose FIFO did not fire through silent.next; switching the rule off underneath it would advance every stream on every fired
  // step while the key/value installs and the codebook lottery still skip the flagged ones -- ADVICE r05): refused in both directions
  if (b->rb.on) return -1;
  if (!enable) {
    if (sr.on) { (void)sync_all(b); if (b->rw.ready) sr.on = false; else silent_release(b); }   // (the per-stream wrapper keeps the freeze machinery)
    return 0;
  }
  if (sr.on) return 0;
  if (b->tk.on) {   // tick mode (plain resident I/O): the flagged streams sit the next STEP out -- their step counters stand still and
    if (b->hs.on || b->rb.on) return -1;   // travel with every step from then on (batch_tick.hip.h, ragged steps); with the 48 kHz wrapper around the ticks
                                            // (BindResidentIO48k) that is the shell's rule on its blocks -- at two / four blocks per step on a stream's WHOLE step
    sr.next.assign(b->B, 0);
    sr.any_next = false;
    sr.on = true;
    return 0;
  }
  if (b->H != 1 || b->pipelined || b->tk.on || b->io_slots > 0) return -1;   // the in-order chain, one block per step
  if (!sync_all(b)) return -2;
  if (!freeze_prepare(b)) { silent_release(b); return -2; }
  sr.on = true;
  drop_graph(b);
  return 0;
}
// what a step that leaves some streams standing needs (the silent-block rule of the 48 kHz blocks, and the per-stream clocks of
// BeatriceBatch_ProcessBlocksRagged): the list of state rings to put back, room for the single-slot ones, the flag staging
static bool freeze_prepare(BeatriceBatch* b) {
  BeatriceBatch::SilentRule& sr = b->silent;
  if (sr.d_rings) return true;
  std::vector<FreezeRing> rings;
  size_t keep = 0;
  for (const RingArena* a : {&b->phone.arena, &b->pitch.arena, &b->wave.arena})
    for (const Ring* r : a->rings) {
      FreezeRing f{r->base, r->n * r->C, r->m, 0};
      if (r->m == 1) { f.keep_off = keep; keep += (size_t)b->B * f.slot_floats; }
      rings.push_back(f);
    }
  sr.n_rings = (int)rings.size();
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&sr.d_rings), sizeof(FreezeRing) * rings.size()), "silent rings") &&
            hip_ok(hipMemcpy(sr.d_rings, rings.data(), sizeof(FreezeRing) * rings.size(), hipMemcpyHostToDevice), "silent rings up") &&
            hip_ok(hipMalloc(reinterpret_cast<void**>(&sr.d_keep), sizeof(float) * std::max<size_t>(keep, 1)), "silent keep") &&
            hip_ok(hipMalloc(reinterpret_cast<void**>(&sr.d_keep_prev_q), sizeof(int) * b->B), "silent prev_q") &&
            hip_ok(hipMalloc(reinterpret_cast<void**>(&sr.d_flags), BeatriceBatch::SilentRule::kDepth * (size_t)b->B), "silent flags") &&
            hip_ok(hipHostMalloc(reinterpret_cast<void**>(&sr.h_flags), BeatriceBatch::SilentRule::kDepth * (size_t)b->B, hipHostMallocDefault), "silent flags host");
  for (hipEvent_t& e : sr.flag_ev) ok = ok && hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "silent flag event");
  if (!ok) return false;
  sr.next.assign(b->B, 0);
  sr.any_next = false;
  return true;
}
// streams whose NEXT 48 kHz block is silent by the shell's rule (flags[B], non-zero = silent); cleared by that block
int BeatriceBatch_SetSilentStreams(BeatriceBatch* b, const unsigned char* flags) {
  if (!b || !b->ok) return -2;
  if (!b->silent.on || !flags) return -1;
  if ((b->H != 1 && !b->tk.on) || b->pipelined || (!b->tk.on && b->io_slots > 0) || (b->tk.on && (b->hs.on || b->rb.on))) return -1;   // (the rule's modes: EnableSilentBlockRule)
  b->silent.any_next = false;
  for (int s = 0; s < b->B; ++s) { b->silent.next[s] = flags[s] ? 1 : 0; b->silent.any_next = b->silent.any_next || flags[s]; }
  return 0;
}
// Throughput form of the 48 kHz wrapper: n_slots resident 48 kHz blocks per direction, the tick pipeline between them.
// Block k (BeatriceBatch_ConvertBlocks48kDevice(b, NULL, NULL, channels)) is read from slot k mod n_slots; its converted
// block lands in the same slot of d_out48 BeatriceBatch_TickStages() - 1 calls later (or after BeatriceBatch_Synchronize).
// Same samples as the in-order BeatriceBatch_ConvertBlocks48kDevice.  NULL pointers unbind.
int BeatriceBatch_BindResidentIO48k(BeatriceBatch* b, const float* d_in48, float* d_out48, int channels, int n_slots) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::Resident48& r = b->r48;
  if (r.on) {
    if (!sync_all(b)) return -2;
    const int rc = tick_enable(b, false);
    if (rc) return rc;
    (void)BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0);
    if (r.d_in16) (void)hipFree(r.d_in16);
    if (r.d_out24) (void)hipFree(r.d_out24);
    r = BeatriceBatch::Resident48{};
  }
  if (!d_in48 && !d_out48) return 0;
  if (!d_in48 || !d_out48 || channels < 1 || channels > 2 || n_slots < b->tk.plan.count() + 1 || b->H > tick::kMaxHops || b->io_slots > 0 || b->pipelined ||
      b->tk.on || b->hs.on || b->silent.on)   // (the silent-block rule is an in-order mode: switch it off first)
    return -1;
  // (with H hops per step a slot holds H blocks per stream, [B][H][channels][480], and a call converts them all)
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_in16), sizeof(float) * n_slots * b->B * b->H * B_IN_HOP), "r48 in16") &&
            hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_out24), sizeof(float) * n_slots * b->B * b->H * B_OUT_HOP), "r48 out24") &&
            hip_ok(hipMemset(r.d_in16, 0, sizeof(float) * n_slots * b->B * b->H * B_IN_HOP), "r48 zero");
  ok = ok && BeatriceBatch_BindResidentIO(b, r.d_in16, r.d_out24, n_slots) == 0 && tick_enable(b, true) == 0;
  if (!ok) {
    (void)tick_enable(b, false);
    (void)BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0);
    if (r.d_in16) (void)hipFree(r.d_in16);
    if (r.d_out24) (void)hipFree(r.d_out24);
    r = BeatriceBatch::Resident48{};
    return -2;
  }
  r.d_in48 = d_in48; r.d_out48 = d_out48; r.channels = channels; r.n_slots = n_slots; r.on = true;
  return 0;
}
int BeatriceBatch_ConvertBlocks48kDevice(BeatriceBatch* b, const float* d_in, float* d_out, int channels) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (b->r48.on) return (!d_in && !d_out && channels == b->r48.channels) ? (tick_run(b, true) ? 0 : -2) : -1;
  if (channels < 1 || channels > 2 || !d_in || !d_out || b->H != 1 || b->io_slots > 0 || b->pipelined || b->tk.on) return -1;  // per 10 ms block, in order
  return step_48k(b, d_in, d_out, channels) ? 0 : -2;
}
int BeatriceBatch_ConvertBlocks48k(BeatriceBatch* b, const float* in, float* out, int channels) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (channels < 1 || channels > 2 || !in || !out || b->H != 1 || b->io_slots > 0 || b->pipelined || b->tk.on) return -1;
  const size_t n = (size_t)b->B * channels * 480;
  float* h_in = b->h_io48;
  float* h_out = b->h_io48 + (size_t)b->B * 2 * 480;
  float* d_in = b->d_io48;
  float* d_out = b->d_io48 + (size_t)b->B * 2 * 480;
  if (b->silent.on) {   // the shell's test, per stream: every sample of the down-mix equal to 0.0f
    BeatriceBatch::SilentRule& sr = b->silent;
    sr.any_next = false;
    for (int st = 0; st < b->B; ++st) {
      const float* src = in + (size_t)st * channels * 480;
      bool sil = true;
      for (int i = 0; i < 480 && sil; ++i) {
        float m = src[i];
        if (channels >= 2) { m = m + src[480 + i]; m = m * 0.5f; }
        sil = !(m != 0.0f);
      }
      sr.next[st] = sil ? 1 : 0;
      sr.any_next = sr.any_next || sil;
    }
  }
  std::memcpy(h_in, in, sizeof(float) * n);
  bool ok = hip_ok(hipMemcpyAsync(d_in, h_in, sizeof(float) * n, hipMemcpyHostToDevice, b->stream), "in48");
  ok = ok && step_48k(b, d_in, d_out, channels);
  ok = ok && hip_ok(hipMemcpyAsync(h_out, d_out, sizeof(float) * n, hipMemcpyDeviceToHost, b->stream), "out48");
  ok = hip_ok(hipStreamSynchronize(b->stream), "sync") && ok;
  b->inflight = false;
  if (ok) std::memcpy(out, h_out, sizeof(float) * n);
  else std::memset(out, 0, sizeof(float) * n);
  return ok ? 0 : -2;
}

// ---- host-rate blocks with the whole wrapper on the device (wrapper.hip.h) -------------------------------------------
namespace { constexpr int kInnerStride = wrapn::kMaxSamples + 64; }
int BeatriceBatch_ConfigureWrapper(BeatriceBatch* b, double sample_rate) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (b->H != 1 && b->H != 2 && b->H != 4) return -1;   // (several hops per step: for BeatriceBatch_BindResidentBlocks, the wrapper around the ticks)
  if (b->rb.on) return -1;                              // resident blocks are bound: the binding restarts the wrapper itself when it is made and released
  if (!sync_all(b)) return -2;
  if (!b->wrap.configure(sample_rate)) return -1;  // rate <= 0, or a ratio whose filter history exceeds the state block
  const int B = b->B;
  const size_t nt = b->wrap.taps_down.size();
  bool ok = true;
  if (!b->d_wrap) {
    ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_wrap), sizeof(wrapn::StreamState) * B), "wrap state") &&
         hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_wrap_inner), sizeof(float) * B * kInnerStride), "wrap inner") &&
         hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_wrap_io), sizeof(float) * B * 4 * wrapn::kMaxSamples), "wrap io") &&
         hip_ok(hipHostMalloc(reinterpret_cast<void**>(&b->h_wrap_io), sizeof(float) * B * 4 * wrapn::kMaxSamples, hipHostMallocDefault), "wrap io host") &&
         b->wrap_gains.alloc_host(2 * (size_t)B) &&
         hip_ok(hipMalloc(reinterpret_cast<void**>(&b->wrap_gains.d), sizeof(wrapn::GainSeg) * 2 * B), "wrap gains");
    b->gain_in.assign(B, wrapn::GainClock());
    b->gain_out.assign(B, wrapn::GainClock());
  }
  if (b->d_wrap_taps) { (void)hipFree(b->d_wrap_taps); b->d_wrap_taps = nullptr; }
  ok = ok && hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_wrap_taps), sizeof(float) * 2 * nt), "wrap taps") &&
       hip_ok(hipMemcpy(b->d_wrap_taps, b->wrap.taps_down.data(), sizeof(float) * nt, hipMemcpyHostToDevice), "taps down") &&
       hip_ok(hipMemcpy(b->d_wrap_taps + nt, b->wrap.taps_up.data(), sizeof(float) * nt, hipMemcpyHostToDevice), "taps up") &&
       hip_ok(hipMemset(b->d_wrap, 0, sizeof(wrapn::StreamState) * B), "wrap state0") && hip_ok(hipDeviceSynchronize(), "wrap sync");
  b->wrap_gains_constant = false;
  // (a new rate restarts the resampler and the FIFO as the reference's SetSampleRate does; the gains keep their state,
  //  now ramping at the new rate: reference processor_core_2.cc:421-429)
  return ok ? 0 : -2;
}
// reference ProcessorCore2::SetInputGain / SetOutputGain (processor_core_2.cc:488-496): the target; the ramp follows at 2 dB/ms
int BeatriceBatch_SetInputGain(BeatriceBatch* b, int stream, double db) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (stream < -1 || stream >= b->B || b->gain_in.empty()) return -1;
  for (int s = (stream < 0 ? 0 : stream); s < (stream < 0 ? b->B : stream + 1); ++s) b->gain_in[s].target_db = db;
  return 0;
}
int BeatriceBatch_SetOutputGain(BeatriceBatch* b, int stream, double db) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (stream < -1 || stream >= b->B || b->gain_out.empty()) return -1;
  for (int s = (stream < 0 ? 0 : stream); s < (stream < 0 ? b->B : stream + 1); ++s) b->gain_out[s].target_db = db;
  return 0;
}
static bool wrap_chunk(BeatriceBatch* b, const float* d_in, float* d_out, int channels, int n) {
  using namespace wrapn;
  const int B = b->B;
  hipStream_t st = b->stream;
  WrapPlan& w = b->wrap;
  // gains: this call's segment per stream; the device copy is refreshed unless it already holds the same constants
  bool all_constant = true;
  GainSeg* seg = b->wrap_gains.h;
  for (int s = 0; s < B; ++s) {
    const GainSeg gi = b->gain_in[s].advance(n, w.rate), go = b->gain_out[s].advance(n, w.rate);
    all_constant = all_constant && gi.step == 1.0 && go.step == 1.0 && seg[s].step == 1.0 && seg[B + s].step == 1.0 &&
                   seg[s].amp0 == gi.amp0 && seg[B + s].amp0 == go.amp0;
    seg[s] = gi;
    seg[B + s] = go;
  }
  if (!(all_constant && b->wrap_gains_constant)) {
    const size_t off = 0, len = 2 * (size_t)B;
    GainSeg* dst = nullptr;
    if (!b->wrap_gains.push_parts(st, 1, &off, &len, &dst)) return false;
    b->wrap_gains_constant = all_constant;
  }
  const size_t nt = w.taps_down.size();
  const Dir din = w.to_inner(n);
  const int m = din.n_out;
  if (m < 0 || m > kMaxSamples) return false;
  hipLaunchKernelGGL(wrap_in_kernel, dim3(B), dim3(256), 0, st, d_in, channels, n, b->d_wrap, b->wrap_gains.d, b->d_wrap_taps + (din.decimate ? 0 : nt), din,
                     b->d_wrap_inner, kInnerStride);
  for (int at = 0; at < m;) {  // the exact-480 FIFO; a model hop every time it fills
    const int take = std::min(kBlock - w.fill, m - at);
    const int fires = w.fill + take == kBlock ? 1 : 0;
    hipLaunchKernelGGL(wrap_fifo_kernel, dim3(B), dim3(256), 0, st, b->d_wrap_inner, kInnerStride, b->d_wrap, at, w.fill, take, fires, b->d_in, B_IN_HOP);
    if (fires) {
      if (!step_device(b, nullptr, nullptr)) return false;
      hipLaunchKernelGGL(wrap_refill_kernel, dim3((B * kBlock + 255) / 256), dim3(256), 0, st, b->d_wrap, b->wave.d_out, B);
      w.fill = 0;
    } else {
      w.fill += take;
    }
    at += take;
  }
  const Dir dout = w.to_outer(m);
  if (dout.n_out != n) return false;  // the two clocks are coupled so that a block comes back with its own length
  hipLaunchKernelGGL(wrap_out_kernel, dim3(B), dim3(256), 0, st, b->d_wrap_inner, kInnerStride, b->d_wrap, b->wrap_gains.d + B,
                     b->d_wrap_taps + (dout.decimate ? 0 : nt), dout, d_out, channels);
  return hip_ok(hipGetLastError(), "wrapper launch");
}
static int wrap_max_chunk(const BeatriceBatch* b) {  // host samples per launch so that neither side exceeds the kernels' LDS buffers
  const double r = b->wrap.rate / 48000.0;
  return std::max(1, (int)std::floor((wrapn::kMaxSamples - 8) * std::min(1.0, r)));
}
// in / out: [B][channels][n] planar at the configured host rate; any n >= 1 (long blocks are processed in pieces)
int BeatriceBatch_ProcessBlocksDevice(BeatriceBatch* b, const float* d_in, float* d_out, int channels, int n) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (b->rb.on) {
    if (b->rb.ragged || d_in || d_out || channels != b->rb.channels || n != b->rb.n) return -1;
    if (b->rb.dead) return -2;
    if (!rb_step(b)) { b->rb.dead = true; return -2; }   // (rb_step advances gain clocks, resampler phases and the FIFO before its checks: a failed call leaves the binding unusable, not silently skewed)
    return 0;
  }
  if (!b->wrap.ready || channels < 1 || channels > 2 || !d_in || !d_out || n < 1 || b->H != 1 || b->io_slots > 0 || b->pipelined || b->tk.on) return -1;
  const int piece = wrap_max_chunk(b);
  if (n <= piece) return wrap_chunk(b, d_in, d_out, channels, n) ? 0 : -2;
  return -1;  // the planar layout [B][channels][n] cannot be cut without copies: callers pass blocks of at most `piece` samples
}
int BeatriceBatch_MaxWrapperBlock(const BeatriceBatch* b) { return b && b->ok && b->wrap.ready ? wrap_max_chunk(b) : 0; }

// ---- clocks per stream: a batch whose streams come from different hosts -------------------------------------------------------------
// rates[B]: the host rate of every stream (streams of equal rate share their tap tables).  Restarts every stream's resampler
// pair and FIFO as SetSampleRate does; the gains keep their state.  In-order mode, one hop per step.
int BeatriceBatch_ConfigureWrapperRates(BeatriceBatch* b, const double* rates) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (!rates || b->H != 1 || b->pipelined || b->tk.on || b->io_slots > 0) return -1;
  if (!sync_all(b)) return -2;
  BeatriceBatch::RaggedWrap& r = b->rw;
  const int B = b->B;
  std::vector<wrapn::WrapPlan> classes;
  std::vector<int> cls(B);
  for (int s = 0; s < B; ++s) {
    int c = -1;
    for (size_t i = 0; i < classes.size(); ++i) if (classes[i].rate == rates[s]) { c = (int)i; break; }
    if (c < 0) {
      wrapn::WrapPlan p;
      if (!p.configure(rates[s])) return -1;
      classes.push_back(p);
      c = (int)classes.size() - 1;
    }
    cls[s] = c;
  }
  // the uniform wrapper's per-stream state, inner buffer, I/O staging and gain mirrors are shared with this mode
  if (!b->d_wrap) { const int rc = BeatriceBatch_ConfigureWrapper(b, rates[0]); if (rc) return rc; }
  if (!freeze_prepare(b)) return -2;
  std::vector<float> taps;
  r.taps_down_off.clear(); r.taps_up_off.clear();
  for (const wrapn::WrapPlan& p : classes) {
    r.taps_down_off.push_back((int)taps.size()); taps.insert(taps.end(), p.taps_down.begin(), p.taps_down.end());
    r.taps_up_off.push_back((int)taps.size()); taps.insert(taps.end(), p.taps_up.begin(), p.taps_up.end());
  }
  if (r.d_taps) { (void)hipFree(r.d_taps); r.d_taps = nullptr; }
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_taps), sizeof(float) * taps.size()), "ragged taps") &&
            hip_ok(hipMemcpy(r.d_taps, taps.data(), sizeof(float) * taps.size(), hipMemcpyHostToDevice), "ragged taps up");
  if (ok && !r.d_rs) {
    ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_rs), sizeof(wrapn::RagStream) * r.kStage * B), "ragged records") &&
         hip_ok(hipHostMalloc(reinterpret_cast<void**>(&r.h_rs), sizeof(wrapn::RagStream) * r.kStage * B, hipHostMallocDefault), "ragged records host") &&
         hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_frozen), (size_t)wrapn::kMaxChunks * B), "ragged flags");
    for (hipEvent_t& e : r.ev) ok = ok && hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "ragged event");
  }
  ok = ok && hip_ok(hipMemset(b->d_wrap, 0, sizeof(wrapn::StreamState) * B), "ragged state0") && hip_ok(hipDeviceSynchronize(), "ragged sync");
  if (!ok) return -2;
  r.classes = classes;
  r.cls = cls;
  r.clk.assign(B, BeatriceBatch::RaggedWrap::Clock{0, 0, 0});
  for (int s = 0; s < B; ++s) r.clk[s] = BeatriceBatch::RaggedWrap::Clock{classes[cls[s]].hi - 1, classes[cls[s]].hi - 1, 0};
  b->wrap.ready = false;          // (the uniform entry points are off until BeatriceBatch_ConfigureWrapper is called again)
  b->wrap_gains_constant = false;
  drop_graph(b);
  r.ready = true;
  return 0;
}
// One active stream's share of a call with per-stream clocks: its gain segments, the two resampling directions and the pieces its inner
// samples are cut into at the 480-sample FIFO, from the stream's own clocks (which advance).  false: the plan does not fit the kernels'
// buffers -- the caller puts every clock of the call back.
static bool rag_plan(BeatriceBatch* b, int s, int n, wrapn::RagStream& q, wrapn::GainSeg& seg_in, wrapn::GainSeg& seg_out) {
  using namespace wrapn;
  BeatriceBatch::RaggedWrap& r = b->rw;
  WrapPlan& p = r.classes[r.cls[s]];
  BeatriceBatch::RaggedWrap::Clock& c = r.clk[s];
  p.phase_down = c.phase_down; p.phase_up = c.phase_up; p.fill = c.fill;   // the class's plan does the clock arithmetic on the stream's state
  seg_in = b->gain_in[s].advance(n, p.rate);
  seg_out = b->gain_out[s].advance(n, p.rate);
  q.din = p.to_inner(n);
  const int m = q.din.n_out;
  if (m < 0 || m > kMaxSamples) return false;
  int fill = p.fill, nc = 0;
  for (int at = 0; at < m;) {
    const int take = std::min(kBlock - fill, m - at);
    if (nc >= kMaxChunks) return false;
    q.at[nc] = (short)at; q.fill[nc] = (short)fill; q.take[nc] = (short)take;
    q.fires[nc] = fill + take == kBlock ? 1 : 0;
    fill = q.fires[nc] ? 0 : fill + take;
    at += take;
    ++nc;
  }
  q.n_chunks = nc;
  p.fill = fill;
  q.dout = p.to_outer(m);
  if (q.dout.n_out != n) return false;
  q.taps_in = q.din.decimate ? r.taps_down_off[r.cls[s]] : r.taps_up_off[r.cls[s]];
  q.taps_out = q.dout.decimate ? r.taps_down_off[r.cls[s]] : r.taps_up_off[r.cls[s]];
  c.phase_down = p.phase_down; c.phase_up = p.phase_up; c.fill = p.fill;
  return true;
}
// One call = for every stream s a block of n_samples[s] host samples at ITS rate (0: the stream sits this call out).  in / out:
// the streams' planar blocks [channels][n_samples[s]] one after the other.  apply_silent_rule != 0: a block whose down-mix is
// all zeros is not converted (src/vst/processor.cc:204-214): nothing of that stream moves -- gains, resampler clocks, FIFO,
// model state, key/value installs, codebook lottery -- and its output block is zeros.  Streams fire their model hops when
// THEIR 480-sample FIFO fills; a step runs for the streams that fire in it and leaves the others standing.
int BeatriceBatch_ProcessBlocksRagged(BeatriceBatch* b, const float* in, float* out, int channels, const int* n_samples, int apply_silent_rule) {
  using namespace wrapn;
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::RaggedWrap& r = b->rw;
  if (!r.ready || !in || !out || !n_samples || channels < 1 || channels > 2 || b->H != 1 || b->io_slots > 0 || b->pipelined || b->tk.on) return -1;
  const int B = b->B;
  hipStream_t st = b->stream;
  size_t total = 0;
  for (int s = 0; s < B; ++s) {
    if (n_samples[s] < 0) return -1;
    if (n_samples[s] > 0 && n_samples[s] > std::max(1, (int)std::floor((kMaxSamples - 8) * std::min(1.0, r.classes[r.cls[s]].rate / 48000.0)))) return -1;
    total += (size_t)channels * n_samples[s];
  }
  const int e = (int)(r.calls % r.kStage);
  if (r.pending[e]) { if (!hip_ok(hipEventSynchronize(r.ev[e]), "ragged staging")) return -2; r.pending[e] = false; }
  RagStream* rs = r.h_rs + (size_t)e * B;
  GainSeg* seg = b->wrap_gains.h;
  int max_chunks = 0;
  size_t off = 0;
  // (all or nothing: the host-side clocks of the streams planned so far go back to where they were when a later stream's plan does
  //  not fit -- otherwise host and device state of those streams would part ways)
  r.clk_undo = r.clk;
  r.gain_undo_in.assign(b->gain_in.begin(), b->gain_in.end());
  r.gain_undo_out.assign(b->gain_out.begin(), b->gain_out.end());
  auto refuse = [&]() { r.clk = r.clk_undo; std::copy(r.gain_undo_in.begin(), r.gain_undo_in.end(), b->gain_in.begin()); std::copy(r.gain_undo_out.begin(), r.gain_undo_out.end(), b->gain_out.begin()); return -2; };
  for (int s = 0; s < B; ++s) {
    RagStream& q = rs[s];
    q = RagStream{};
    const int n = n_samples[s];
    q.io_off = (long long)off; q.n = n;
    bool active = n > 0;
    if (active && apply_silent_rule) {   // the shell's own test on the down-mix
      const float* src = in + off;
      bool sil = true;
      for (int i = 0; i < n && sil; ++i) {
        float m = src[i];
        if (channels >= 2) { m = m + src[n + i]; m = m * 0.5f; }
        sil = !(m != 0.0f);
      }
      active = !sil;
    }
    off += (size_t)channels * n;
    q.active = active ? 1 : 0;
    if (!active) { seg[s] = GainSeg{1.0, 1.0, 1.0}; seg[B + s] = GainSeg{1.0, 1.0, 1.0}; continue; }
    if (!rag_plan(b, s, n, q, seg[s], seg[B + s])) return refuse();
    const int nc = q.n_chunks;
    max_chunks = std::max(max_chunks, nc);
  }
  // uploads: the per-stream records of this call, the gain segments, the audio
  RagStream* d_rs = r.d_rs + (size_t)e * B;
  bool ok = hip_ok(hipMemcpyAsync(d_rs, rs, sizeof(RagStream) * B, hipMemcpyHostToDevice, st), "ragged records up");
  { const size_t o0 = 0, len = 2 * (size_t)B; GainSeg* dst = nullptr; ok = ok && b->wrap_gains.push_parts(st, 1, &o0, &len, &dst); }
  b->wrap_gains_constant = false;
  float* h_in = b->h_wrap_io;
  float* h_out = b->h_wrap_io + (size_t)B * 2 * kMaxSamples;
  float* d_in = b->d_wrap_io;
  float* d_out = b->d_wrap_io + (size_t)B * 2 * kMaxSamples;
  std::memcpy(h_in, in, sizeof(float) * total);
  ok = ok && hip_ok(hipMemcpyAsync(d_in, h_in, sizeof(float) * std::max<size_t>(total, 1), hipMemcpyHostToDevice, st), "ragged in");
  if (!ok) return -2;
  hipLaunchKernelGGL(wrapr_in_kernel, dim3(B), dim3(256), 0, st, d_in, channels, b->d_wrap, b->wrap_gains.d, r.d_taps, d_rs, b->d_wrap_inner, kInnerStride);
  BeatriceBatch::SilentRule& sr = b->silent;
  for (int ci = 0; ci < max_chunks && ok; ++ci) {
    bool any_fire = false;
    for (int s = 0; s < B; ++s) {
      const bool fires = rs[s].active && ci < rs[s].n_chunks && rs[s].fires[ci];
      sr.next[s] = fires ? 0 : 1;
      any_fire = any_fire || fires;
    }
    unsigned char* d_flags = r.d_frozen + (size_t)ci * B;
    hipLaunchKernelGGL(wrapr_fifo_kernel, dim3(B), dim3(256), 0, st, b->d_wrap_inner, kInnerStride, b->d_wrap, d_rs, ci, b->d_in, d_flags, nullptr, 0, 0);
    if (!any_fire) { std::fill(sr.next.begin(), sr.next.end(), 0); continue; }
    bool any_frozen = false;
    for (int s = 0; s < B; ++s) any_frozen = any_frozen || sr.next[s];
    sr.any_next = any_frozen;
    if (any_frozen)
      hipLaunchKernelGGL(freeze_save_kernel, dim3(sr.n_rings, B), dim3(256), 0, st, sr.d_rings, sr.d_keep, B, b->pitch.d_prev_q, sr.d_keep_prev_q);
    sr.in_block_step = true;
    ok = step_device(b, nullptr, nullptr);   // (advance_kv / draw_codebooks skip the streams flagged in sr.next)
    sr.in_block_step = false;
    std::fill(sr.next.begin(), sr.next.end(), 0);
    sr.any_next = false;
    if (!ok) break;
    if (any_frozen)
      hipLaunchKernelGGL(freeze_fix_kernel, dim3(sr.n_rings, B), dim3(256), 0, st, sr.d_rings, sr.d_keep, B, d_flags, b->last_hop, b->pitch.d_prev_q, sr.d_keep_prev_q);
    hipLaunchKernelGGL(wrapr_refill_kernel, dim3((B * kBlock + 255) / 256), dim3(256), 0, st, b->d_wrap, b->wave.d_out, B, d_flags);
  }
  if (ok) {
    hipLaunchKernelGGL(wrapr_out_kernel, dim3(B), dim3(256), 0, st, b->d_wrap_inner, kInnerStride, b->d_wrap, b->wrap_gains.d + B, r.d_taps, d_rs, d_out, channels);
    ok = hip_ok(hipGetLastError(), "ragged wrapper launch") && hip_ok(hipEventRecord(r.ev[e], st), "ragged event");
    r.pending[e] = ok;
  }
  r.calls += 1;
  ok = ok && hip_ok(hipMemcpyAsync(h_out, d_out, sizeof(float) * std::max<size_t>(total, 1), hipMemcpyDeviceToHost, st), "ragged out");
  ok = hip_ok(hipStreamSynchronize(st), "ragged sync") && ok;
  b->inflight = false;
  if (ok) std::memcpy(out, h_out, sizeof(float) * total);
  else std::memset(out, 0, sizeof(float) * total);
  return ok ? 0 : -2;
}

// ---- the same wrapper around the TICK pipeline (throughput form, resident blocks) ------------------------------------------------
// One call = one host-rate block per stream from slot `call mod n_slots` of d_in: gains and the first resampling direction, the
// 480-sample accumulation, a model hop into the tick pipeline every time it fills (one tick per hop, at least one tick per
// call so that a hop is out of the pipeline TickStages() - 1 calls after it went in); then the output half of the call made
// `delay` = TickStages() - 1 calls ago, into ITS slot of d_out.  Everything that is control is on the host, as in wrap_chunk.
static void rb_release(BeatriceBatch* b) {
  BeatriceBatch::ResidentBlocks& r = b->rb;
  if (r.d_in16) (void)hipFree(r.d_in16);
  if (r.d_out24) (void)hipFree(r.d_out24);
  if (r.h_gains) (void)hipHostFree(r.h_gains);
  if (r.gain_ev) { for (int i = 0; i < r.ring; ++i) if (r.gain_ev[i]) (void)hipEventDestroy(r.gain_ev[i]); delete[] r.gain_ev; }
  if (r.h_rs) (void)hipHostFree(r.h_rs);
  if (r.d_map) (void)hipFree(r.d_map);
  if (r.d_zero) (void)hipFree(r.d_zero);
  r = BeatriceBatch::ResidentBlocks{};
}
// leaves either form of the resident blocks: the pipeline drained, tick mode off, the wrapper restarted at its rate(s)
static int rb_unbind(BeatriceBatch* b) {
  BeatriceBatch::ResidentBlocks& r = b->rb;
  if (!sync_all(b)) return -2;
  const int rc = tick_enable(b, false);
  if (rc) return rc;
  (void)BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0);
  const bool ragged = r.ragged;
  rb_release(b);
  if (ragged) {
    b->silent.on = false;   // (the flags of the ragged steps were the binding's own)
    std::vector<double> rates(b->B);
    for (int s = 0; s < b->B; ++s) rates[s] = b->rw.classes[b->rw.cls[s]].rate;
    (void)BeatriceBatch_ConfigureWrapperRates(b, rates.data());
  } else {
    (void)BeatriceBatch_ConfigureWrapper(b, b->wrap.rate);   // the wrapper restarts, as after a sample-rate change (the gains keep their state)
  }
  return 0;
}
// Calls after which a call's output block is out.  One hop per step: every call runs at least one tick, so TickStages() - 1 calls
// behind the call that fed its newest hop.  H hops per step: ticks run only when a step is full (an idle tick per call would carry a
// quarter-filled pipeline and cost what a full one does) -- the newest hop waits until up to H - 1 further hops have filled its step
// and TickStages() - 1 further steps have gone in behind it, each call bringing at least m_lo inner samples (the two clocks make a
// call's count vary by one).  -1: blocks too short to bound it (H > 1 only).
static int rb_delay(const BeatriceBatch* b, int n) {
  const int stages = b->tk.plan.count();
  if (b->H == 1) return stages - 1;
  const long long m_lo = (long long)std::floor(n * 48000.0 / b->wrap.rate) - 1;
  if (m_lo < 1) return -1;
  const long long hops = (long long)(b->H - 1) + (long long)(stages - 1) * b->H;
  return (int)((hops * wrapn::kBlock + m_lo - 1) / m_lo) + 1;   // (+ 1: the output half rides in the launch of the call after)
}
// synthetic: a call BeatriceBatch_FlushResidentBlocks makes up -- a block of zeros that is no call of the caller's: it reads no slot, owes no
// output block and does not count (the call counter stands), but moves the wrapper and fires hops as a block of silence would
static bool rb_step(BeatriceBatch* b, const bool synthetic) {
  using namespace wrapn;
  BeatriceBatch::ResidentBlocks& r = b->rb;
  const int B = b->B, n = r.n, H = r.H;
  hipStream_t st = b->stream;
  WrapPlan& w = b->wrap;
  const long long call = r.calls;
  const int ge = (int)(call % r.ring);
  // this call's gain segments: input half now, output half when its job runs.  The kernels read them where they are written (pinned
  // memory): the entry is free again once the output half that read it last has run
  if (r.ev_recorded[ge]) { if (!hip_ok(hipEventSynchronize(r.gain_ev[ge]), "wrapper gain ring")) return false; r.ev_recorded[ge] = 0; }
  GainSeg* seg = r.h_gains + (size_t)ge * 2 * B;
  for (int s = 0; s < B; ++s) { seg[s] = b->gain_in[s].advance(n, w.rate); seg[B + s] = b->gain_out[s].advance(n, w.rate); }
  const size_t nt = w.taps_down.size();
  WrapCallArgs a{};
  a.din = w.to_inner(n);
  const int m = a.din.n_out;
  if (m < 0 || m > kMaxSamples) return false;
  const Dir dout = w.to_outer(m);
  if (dout.n_out != n) return false;
  a.src = synthetic ? r.d_zero : r.d_in + (size_t)(call % r.n_slots) * B * r.channels * n;
  a.channels = r.channels; a.n = n; a.st = b->d_wrap; a.gain_in = seg; a.taps_in = b->d_wrap_taps + (a.din.decimate ? 0 : nt);
  a.inner = b->d_wrap_inner; a.stride = kInnerStride; a.in16 = r.d_in16; a.row16 = H * B_IN_HOP; a.B = B; a.H = H;
  // the 480-sample accumulation; a model hop every time it fills (the per-stream FIFO array holds it), into its place: hop
  // hops_fired % H of the step that goes in next -- the steps this call fills take the resident slots from io_host on
  int steps = 0, slot = b->io_host;
  for (int at = 0; at < m;) {
    const int take = std::min(kBlock - w.fill, m - at);
    const int fires = w.fill + take == kBlock ? 1 : 0;
    if (a.n_chunks >= kMaxChunks) return false;
    const int c = a.n_chunks++;
    a.at[c] = (short)at; a.fill[c] = (short)w.fill; a.take[c] = (short)take; a.fires[c] = (unsigned char)fires;
    a.in16_off[c] = (long long)((size_t)slot * B * H + (size_t)(r.hops_fired % H)) * B_IN_HOP;
    if (fires) {
      ++r.hops_fired;
      if (r.hops_fired % H == 0) { ++steps; slot = (slot + 1) % r.io_slots; }   // the step is full
      w.fill = 0;
    } else {
      w.fill += take;
    }
    at += take;
  }
  // several hops per step: the output half of the oldest call that is due rides in this launch (its hops left the ticks with the
  // previous call at the latest: `delay` counts this call)
  const BeatriceBatch::ResidentBlocks::Job* due = nullptr;
  if (H > 1 && !r.jobs.empty() && r.jobs.front().call + r.delay <= call && r.jobs.front().last_hop() < r.hops_done) due = &r.jobs.front();
  int due_ge = -1;
  if (due) {
    due_ge = (int)(due->call % r.ring);
    a.n_post = B; a.out24 = r.d_out24; a.io_slots = r.io_slots; a.t0 = due->t0; a.gain_out = r.h_gains + (size_t)due_ge * 2 * B + B;
    a.taps_out = b->d_wrap_taps + (due->dout.decimate ? 0 : nt); a.dout = due->dout;
    a.out = r.d_out + (size_t)(due->call % r.n_slots) * B * r.channels * n;
  }
  hipLaunchKernelGGL(wrap_call_kernel, dim3(B + a.n_post), dim3(256), 0, st, a);
  if (due) {
    if (!hip_ok(hipEventRecord(r.gain_ev[due_ge], st), "wrapper gain event")) return false;
    r.ev_recorded[due_ge] = 1;
    r.jobs.pop_front();
  }
  // (never more than one call comes due per call under the bind-time bound; should the queue have fallen behind, the rest follow here)
  while (H > 1 && !r.jobs.empty() && r.jobs.front().call + r.delay <= call && r.jobs.front().last_hop() < r.hops_done) {
    if (!rb_post(b, r.jobs.front())) return false;
    r.jobs.pop_front();
  }
  for (int i = 0; i < steps; ++i) if (!tick_run(b, true)) return false;   // the full steps: into the pipeline
  if (steps == 0 && H == 1 && !tick_run(b, false)) return false;          // one hop per step: the pipeline advances with every call
  r.hops_done = r.hops_fed();
  if (!synthetic) r.jobs.push_back(BeatriceBatch::ResidentBlocks::Job{call, r.t48, dout});
  r.t48 += m;
  if (!synthetic) r.calls = call + 1;
  bool ok = hip_ok(hipGetLastError(), "wrapper launch");
  // one hop per step: after `delay` calls a call's hops are out, its output half runs behind this call's tick
  while (ok && H == 1 && !r.jobs.empty() && r.jobs.front().call + r.delay <= call) {
    ok = rb_post(b, r.jobs.front());
    r.jobs.pop_front();
  }
  b->inflight = true;
  return ok;
}
// d_in / d_out: [n_slots][B][channels][n] planar blocks at the configured host rate (BeatriceBatch_ConfigureWrapper first).
// Call k (BeatriceBatch_ProcessBlocksDevice(b, NULL, NULL, channels, n)) reads slot k mod n_slots; its output block is in the
// same slot of d_out BeatriceBatch_ResidentBlocksDelay() calls later (or after BeatriceBatch_Synchronize).  Same samples as
// the in-order BeatriceBatch_ProcessBlocksDevice.  n_slots > delay + 1.  NULL pointers unbind.
int BeatriceBatch_BindResidentBlocks(BeatriceBatch* b, const float* d_in, float* d_out, int channels, int n, int n_slots) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::ResidentBlocks& r = b->rb;
  if (r.on) { const int rc = rb_unbind(b); if (rc) return rc; }
  if (!d_in && !d_out) return 0;
  const int stages = b->tk.plan.count();
  const int H = b->H;
  if (!b->wrap.ready || !d_in || !d_out || channels < 1 || channels > 2 || n < 1 || n > wrap_max_chunk(b) || (H != 1 && H != 2 && H != 4) ||
      b->io_slots > 0 || b->pipelined || b->tk.on || b->hs.on || b->r48.on || b->silent.on)
    return -1;
  const int delay = rb_delay(b, n);
  if (delay < 0 || n_slots < delay + 2) return -1;
  // binding restarts the resampler pair and the FIFO (their in-order form keeps processed samples in the FIFO, this one does not)
  if (BeatriceBatch_ConfigureWrapper(b, b->wrap.rate) != 0) return -2;
  // model hops a call can fire: ceil(inner samples / 480) + 1; a hop's resident output is read until `delay` calls after the
  // call in which the NEXT hop fired
  // call in which the NEXT hop fired.  With H hops per step a slot holds a step: the hops of (delay + 2) calls are that many / H steps.
  const int m_max = (int)std::ceil(n * 48000.0 / b->wrap.rate) + 2, hops_per_call = (m_max + wrapn::kBlock - 1) / wrapn::kBlock + 1;
  r.delay = delay;
  r.ring = r.delay + 3;
  r.H = H;
  r.io_slots = std::max(stages + 1, H == 1 ? (r.delay + 2) * hops_per_call + 2 : ((r.delay + 2) * hops_per_call + H - 1) / H + 3);
  if (r.io_slots > stepc::kImmediateMaxSlot + 1) { r = BeatriceBatch::ResidentBlocks{}; return -1; }   // (tick mode's limit on resident slots)
  const int B = b->B;
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_in16), sizeof(float) * r.io_slots * B * H * B_IN_HOP), "rb in16") &&
            hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_out24), sizeof(float) * r.io_slots * B * H * B_OUT_HOP), "rb out24") &&
            hip_ok(hipMemset(r.d_in16, 0, sizeof(float) * r.io_slots * B * H * B_IN_HOP), "rb zero") &&
            hip_ok(hipMemset(r.d_out24, 0, sizeof(float) * r.io_slots * B * H * B_OUT_HOP), "rb zero") &&
            hip_ok(hipHostMalloc(reinterpret_cast<void**>(&r.h_gains), sizeof(wrapn::GainSeg) * r.ring * 2 * B, hipHostMallocDefault), "rb gains host");
  if (ok) {
    r.gain_ev = new hipEvent_t[r.ring]();
    for (int i = 0; i < r.ring && ok; ++i) ok = hip_ok(hipEventCreateWithFlags(&r.gain_ev[i], hipEventDisableTiming), "rb event");
    r.ev_recorded.assign(r.ring, 0);
  }
  ok = ok && hip_ok(hipDeviceSynchronize(), "rb sync") && BeatriceBatch_BindResidentIO(b, r.d_in16, r.d_out24, r.io_slots) == 0 && tick_enable(b, true) == 0;
  if (!ok) {
    (void)tick_enable(b, false);
    (void)BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0);
    rb_release(b);
    return -2;
  }
  r.d_in = d_in; r.d_out = d_out; r.channels = channels; r.n = n; r.n_slots = n_slots; r.on = true;
  return 0;
}
// ---- the wrapper with clocks PER STREAM around the tick pipeline (BeatriceBatch_ConfigureWrapperRates, then
// BeatriceBatch_BindResidentBlocksRagged / BeatriceBatch_ProcessBlocksRaggedDevice): what the reference gives every plugin instance
// (src/common/resample.h:401-438), in the throughput form.  A call = for every stream a block of n_samples[s] host samples at ITS
// rate from its cell of slot `call mod n_slots`; a stream fires a model hop when ITS 480-sample FIFO fills, and the step that goes
// into the ticks then carries the streams that fired -- the others sit it out with their own step counters (the tick launch's ragged
// steps, batch_tick.hip.h), so a stream's hops ride in steps of their own and slot_map remembers which.  Output half `delay` =
// TickStages() - 1 calls later, from the records the call left on the device.
static int rbr_step(BeatriceBatch* b, const int* n_samples) {
  using namespace wrapn;
  BeatriceBatch::ResidentBlocks& r = b->rb;
  BeatriceBatch::RaggedWrap& rw = b->rw;
  const int B = b->B;
  hipStream_t st = b->stream;
  for (int s = 0; s < B; ++s) {
    const int lim = std::max(1, (int)std::floor((kMaxSamples - 8) * std::min(1.0, rw.classes[rw.cls[s]].rate / 48000.0)));
    if (n_samples[s] < 0 || n_samples[s] > std::min(r.max_samples, lim)) return -1;
  }
  const long long call = r.calls;
  const int ge = (int)(call % r.ring);
  // (records and gain segments are read by the kernels where they are written, pinned memory: the ring entry is free again once the
  //  output half that read it last has run)
  if (r.ev_recorded[ge]) { if (!hip_ok(hipEventSynchronize(r.gain_ev[ge]), "wrapper record ring")) return -2; r.ev_recorded[ge] = 0; }
  RagStream* rs = r.h_rs + (size_t)ge * B;
  GainSeg* seg = r.h_gains + (size_t)ge * 2 * B;
  // all or nothing, as BeatriceBatch_ProcessBlocksRagged: a plan that does not fit puts every clock of the call back
  rw.clk_undo = rw.clk;
  rw.gain_undo_in.assign(b->gain_in.begin(), b->gain_in.end());
  rw.gain_undo_out.assign(b->gain_out.begin(), b->gain_out.end());
  int max_chunks = 0;
  for (int s = 0; s < B; ++s) {
    RagStream& q = rs[s];
    q = RagStream{};
    const int n = n_samples[s];
    q.io_off = (long long)s * r.cell; q.n = n;
    q.active = n > 0 ? 1 : 0;
    q.hop0 = r.hops_s[s]; q.t0 = r.t48_s[s];
    if (!q.active) { seg[s] = GainSeg{1.0, 1.0, 1.0}; seg[B + s] = GainSeg{1.0, 1.0, 1.0}; continue; }
    if (!rag_plan(b, s, n, q, seg[s], seg[B + s])) {
      rw.clk = rw.clk_undo;
      std::copy(rw.gain_undo_in.begin(), rw.gain_undo_in.end(), b->gain_in.begin());
      std::copy(rw.gain_undo_out.begin(), rw.gain_undo_out.end(), b->gain_out.begin());
      return -2;
    }
    max_chunks = std::max(max_chunks, q.n_chunks);
  }
  for (int s = 0; s < B; ++s) {   // (the plan stands: the streams' sample and hop counts move)
    if (!rs[s].active) continue;
    r.t48_s[s] += rs[s].din.n_out;
    for (int c = 0; c < rs[s].n_chunks; ++c) r.hops_s[s] += rs[s].fires[c];
  }
  // one launch for every stream's input half and FIFO pieces; piece c belongs to the c-th step this call feeds (if any stream fires in
  // it), which takes the next resident slot
  WraprCallArgs a{};
  a.in = r.d_in + (size_t)(call % r.n_slots) * B * r.cell; a.channels = r.channels; a.st = b->d_wrap; a.gain = seg; a.taps_all = rw.d_taps;
  a.rs = rs; a.inner = b->d_wrap_inner; a.stride = kInnerStride; a.in16 = r.d_in16; a.B = B; a.slot_map = r.d_map; a.map_ring = r.map_ring;
  bool fire_at[kMaxChunks] = {};
  {
    int slot = b->io_host;
    for (int ci = 0; ci < max_chunks; ++ci) {
      for (int s = 0; s < B && !fire_at[ci]; ++s) fire_at[ci] = rs[s].active && ci < rs[s].n_chunks && rs[s].fires[ci];
      a.slot[ci] = slot;
      if (fire_at[ci]) slot = (slot + 1) % r.io_slots;
    }
  }
  hipLaunchKernelGGL(wrapr_call_kernel, dim3(B), dim3(256), 0, st, a);
  BeatriceBatch::SilentRule& sr = b->silent;
  int ticks = 0;
  for (int ci = 0; ci < max_chunks; ++ci) {
    if (!fire_at[ci]) continue;
    bool any_out = false;
    for (int s = 0; s < B; ++s) {
      const bool fires = rs[s].active && ci < rs[s].n_chunks && rs[s].fires[ci];
      sr.next[s] = fires ? 0 : 1;
      any_out = any_out || !fires;
    }
    sr.any_next = any_out;   // (tick_run: the flagged streams sit this step out, and clears the flags)
    if (!tick_run(b, true)) return -2;
    std::fill(sr.next.begin(), sr.next.end(), 0);
    sr.any_next = false;
    ++ticks;
  }
  if (ticks == 0 && !tick_run(b, false)) return -2;   // the pipeline advances with every call
  r.jobs.push_back(BeatriceBatch::ResidentBlocks::Job{call, 0, Dir{}});
  r.calls = call + 1;
  bool ok = true;
  while (ok && !r.jobs.empty() && r.jobs.front().call + r.delay <= call) {
    ok = rb_post(b, r.jobs.front());
    r.jobs.pop_front();
  }
  b->inflight = true;
  return ok ? 0 : -2;
}
int BeatriceBatch_ProcessBlocksRaggedDevice(BeatriceBatch* b, const int* n_samples) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (!b->rb.on || !b->rb.ragged || !n_samples) return -1;
  return rbr_step(b, n_samples);
}
// d_in / d_out: [n_slots][B][channels * max_samples]: stream s's block of a call, planar [channels][n_samples[s]], at the start of its cell
int BeatriceBatch_BindResidentBlocksRagged(BeatriceBatch* b, const float* d_in, float* d_out, int channels, int max_samples, int n_slots) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::ResidentBlocks& r = b->rb;
  if (r.on) { const int rc = rb_unbind(b); if (rc) return rc; }
  if (!d_in && !d_out) return 0;
  const int stages = b->tk.plan.count();
  if (!b->rw.ready || !d_in || !d_out || channels < 1 || channels > 2 || max_samples < 1 || n_slots < stages + 1 || b->H != 1 ||
      b->io_slots > 0 || b->pipelined || b->tk.on || b->hs.on || b->r48.on || b->silent.on)
    return -1;
  const int B = b->B;
  // binding restarts every stream's resampler pair and FIFO (their in-order form keeps processed samples in the FIFO, this one does not)
  std::vector<double> rates(B);
  for (int s = 0; s < B; ++s) rates[s] = b->rw.classes[b->rw.cls[s]].rate;
  if (BeatriceBatch_ConfigureWrapperRates(b, rates.data()) != 0) return -2;
  // steps a call can feed = the most hops one stream can fire in it; a hop's resident output is read until `delay` calls after the call
  // in which the stream's NEXT hop fired
  int hops_per_call = 1;
  for (int s = 0; s < B; ++s) {
    const int lim = std::max(1, (int)std::floor((wrapn::kMaxSamples - 8) * std::min(1.0, rates[s] / 48000.0)));
    const int m_max = (int)std::ceil(std::min(max_samples, lim) * 48000.0 / rates[s]) + 2;
    hops_per_call = std::max(hops_per_call, (m_max + wrapn::kBlock - 1) / wrapn::kBlock + 1);
  }
  r.delay = stages - 1;
  r.ring = r.delay + 3;
  r.H = 1;
  r.io_slots = std::max(stages + 1, (r.delay + 2) * hops_per_call + 2);
  if (r.io_slots > stepc::kImmediateMaxSlot + 1) { r = BeatriceBatch::ResidentBlocks{}; return -1; }
  r.map_ring = r.io_slots;
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_in16), sizeof(float) * r.io_slots * B * B_IN_HOP), "rb in16") &&
            hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_out24), sizeof(float) * r.io_slots * B * B_OUT_HOP), "rb out24") &&
            hip_ok(hipMemset(r.d_in16, 0, sizeof(float) * r.io_slots * B * B_IN_HOP), "rb zero") &&
            hip_ok(hipMemset(r.d_out24, 0, sizeof(float) * r.io_slots * B * B_OUT_HOP), "rb zero") &&
            hip_ok(hipHostMalloc(reinterpret_cast<void**>(&r.h_gains), sizeof(wrapn::GainSeg) * r.ring * 2 * B, hipHostMallocDefault), "rb gains host") &&
            hip_ok(hipHostMalloc(reinterpret_cast<void**>(&r.h_rs), sizeof(wrapn::RagStream) * r.ring * B, hipHostMallocDefault), "rb records host") &&
            hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_map), sizeof(int) * B * r.map_ring), "rb slot map") &&
            hip_ok(hipMemset(r.d_map, 0, sizeof(int) * B * r.map_ring), "rb slot map zero");
  if (ok) {
    r.gain_ev = new hipEvent_t[r.ring]();
    for (int i = 0; i < r.ring && ok; ++i) ok = hip_ok(hipEventCreateWithFlags(&r.gain_ev[i], hipEventDisableTiming), "rb event");
    r.ev_recorded.assign(r.ring, 0);
  }
  ok = ok && hip_ok(hipDeviceSynchronize(), "rb sync") && BeatriceBatch_BindResidentIO(b, r.d_in16, r.d_out24, r.io_slots) == 0 && tick_enable(b, true) == 0;
  if (!ok) {
    (void)tick_enable(b, false);
    (void)BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0);
    rb_release(b);
    return -2;
  }
  b->silent.next.assign(B, 0);   // the ragged steps' flags (tick_run): set per tick by rbr_step, not by the caller
  b->silent.any_next = false;
  b->silent.on = true;
  r.d_in = d_in; r.d_out = d_out; r.channels = channels; r.max_samples = max_samples; r.cell = channels * max_samples; r.n = 0;
  r.n_slots = n_slots; r.ragged = true;
  r.t48_s.assign(B, 0);
  r.hops_s.assign(B, 0);
  r.on = true;
  return 0;
}
int BeatriceBatch_ResidentBlocksDelay(const BeatriceBatch* b) { return b && b->ok && b->rb.on ? b->rb.delay : -1; }
int BeatriceBatch_ResidentBlocksDelayFor(const BeatriceBatch* b, int n) { return b && b->ok && b->wrap.ready && n >= 1 ? rb_delay(b, n) : -1; }
// End of the material: every call made so far gets its output block as if silence had followed (several hops per step: the step still filling
// is completed with hops of silence -- no caller slot is read or written for them), then the binding starts over as a new one does.
int BeatriceBatch_FlushResidentBlocks(BeatriceBatch* b) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::ResidentBlocks& r = b->rb;
  if (!r.on) return -1;
  if (r.dead) return -2;
  if (!sync_all(b)) return -2;           // (one hop per step and clocks per stream: everything is out now)
  if (r.jobs.empty()) return 0;
  if (r.ragged || r.H == 1) return -2;   // (cannot be: nothing stays owed there)
  if (!r.d_zero) {
    const size_t bytes = sizeof(float) * b->B * r.channels * r.n;
    if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_zero), bytes), "rb zeros") || !hip_ok(hipMemset(r.d_zero, 0, bytes), "rb zeros")) return -2;
  }
  for (int guard = 0; !r.jobs.empty() && guard < 64 * r.H; ++guard) {
    const long long fed = r.hops_fed();
    if (!hip_ok(hipStreamSynchronize(b->stream), "flush sync") || !rb_step(b, true)) { r.dead = true; return -2; }   // (the made-up calls share one gain ring entry: one at a time)
    if (r.hops_fed() > fed && !sync_all(b)) return -2;   // a step went in: drained, the output halves of everything fed have run
  }
  if (!r.jobs.empty()) { r.dead = true; return -2; }
  // the binding starts over: resampler pair and FIFO as after BeatriceBatch_BindResidentBlocks (the gains keep their state), call 0 reads slot 0
  r.on = false;
  const int rc = BeatriceBatch_ConfigureWrapper(b, b->wrap.rate);
  r.on = true;
  if (rc != 0) { r.dead = true; return -2; }
  r.calls = 0; r.t48 = 0; r.hops_fired = 0; r.hops_done = 0;
  std::fill(r.ev_recorded.begin(), r.ev_recorded.end(), 0);
  b->io_host = 0;   // (hop k of the binding rides in resident slot (k / H) mod io_slots: wrap_post_kernel)
  return 0;
}
int BeatriceBatch_ResidentBlocksOwed(const BeatriceBatch* b) { return b && b->ok && b->rb.on ? (int)b->rb.jobs.size() : -1; }
int BeatriceBatch_ProcessBlocks(BeatriceBatch* b, const float* in, float* out, int channels, int n) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (!b->wrap.ready || channels < 1 || channels > 2 || !in || !out || n < 1 || n > wrap_max_chunk(b) || b->H != 1 || b->io_slots > 0 || b->pipelined || b->tk.on) return -1;
  const size_t cnt = (size_t)b->B * channels * n;
  float* h_in = b->h_wrap_io;
  float* h_out = b->h_wrap_io + (size_t)b->B * 2 * wrapn::kMaxSamples;
  float* d_in = b->d_wrap_io;
  float* d_out = b->d_wrap_io + (size_t)b->B * 2 * wrapn::kMaxSamples;
  std::memcpy(h_in, in, sizeof(float) * cnt);
  bool ok = hip_ok(hipMemcpyAsync(d_in, h_in, sizeof(float) * cnt, hipMemcpyHostToDevice, b->stream), "wrap in");
  ok = ok && wrap_chunk(b, d_in, d_out, channels, n);
  ok = ok && hip_ok(hipMemcpyAsync(h_out, d_out, sizeof(float) * cnt, hipMemcpyDeviceToHost, b->stream), "wrap out");
  ok = hip_ok(hipStreamSynchronize(b->stream), "wrap sync") && ok;
  b->inflight = false;
  if (ok) std::memcpy(out, h_out, sizeof(float) * cnt);
  else std::memset(out, 0, sizeof(float) * cnt);
  return ok ? 0 : -2;
}

