// batch_host_stream.hip.h -- host streaming: the tick pipeline with pinned HOST buffers on either side (BeatriceBatch_StreamFrames)
// (Part of batch.hip's translation unit: included there, after struct BeatriceBatch and the helpers above it; not a stand-alone header.)
#pragma once

// ---- host streaming: the tick pipeline with HOST buffers on either side -----------------------------------------------------
// The resident I/O slots of the ticks are the batch's PINNED HOST mirrors: the stages that read a hop (f1, fft, pitch head)
// and the one that writes samples (the tail) go over PCIe themselves -- 160 + 240 KB per tick at 256 streams, spread over
// hundreds of workgroups that have plenty to overlap it with -- so a call is: memcpy the hop into its slot, launch the
// tick, record an event, and hand back the step whose tick finished at least two ticks ago (the host then never waits
// for the device's current work, and two ticks stay queued).  3.07-3.16 M frames/s from and to host memory at 256 streams
// against 3.2-3.55 M with resident device buffers (before / after the last changes of the tick bodies).  BEATRICE_HIP_HS_COPIES=1 (A/B): device slots with an upload and a
// download stream beside the ticks instead -- 2.36 M: copy commands and cross-stream waits cost more than PCIe loads.
static void host_stream_fetch(BeatriceBatch* b) {  // enqueue the download of every step the ticks run so far have completed
  BeatriceBatch::HostStream& h = b->hs;
  const long long last_tick = b->tk.tick - 1;
  const size_t n_out = (size_t)b->B * b->H * B_OUT_HOP;
  for (auto& p : h.pending) {
    if (p.fetched || p.done_tick > last_tick) continue;
    if (h.mapped) { p.fetched = true; continue; }  // nothing to download: the last stage wrote host memory
    // (the event recorded behind the tick just launched: it is at or after the tick that completed this step, also when
    //  ticks were run by a drain in between, which records none)
    (void)hipStreamWaitEvent(h.s_out, h.ev_tick[h.rec[0] % h.ev_tick.size()], 0);
    (void)hipMemcpyAsync(h.h_out + p.slot * n_out, h.d_out + p.slot * n_out, sizeof(float) * n_out, hipMemcpyDeviceToHost, h.s_out);
    (void)hipEventRecord(h.ev_out[p.slot], h.s_out);
    p.fetched = true;
  }
}
static bool host_stream_tick(BeatriceBatch* b, bool feeding) {
  BeatriceBatch::HostStream& h = b->hs;
  if (!tick_run(b, feeding)) return false;   // (may run a whole drain first: a stage that comes or goes)
  const long long t = b->tk.tick - 1;        // the tick just launched
  (void)hipEventRecord(h.ev_tick[t % h.ev_tick.size()], b->stream);
  h.tick_of_ev[t % h.ev_tick.size()] = t;
  h.rec[1] = h.rec[0]; h.rec[0] = t;
  host_stream_fetch(b);
  return true;
}
// the samples of pending step f are in the pinned output mirror
static bool host_stream_wait(BeatriceBatch* b, const BeatriceBatch::HostStream::Pending& f) {
  BeatriceBatch::HostStream& h = b->hs;
  if (!h.mapped) return hip_ok(hipEventSynchronize(h.ev_out[f.slot]), "hs download");
  const size_t n = h.ev_tick.size();
  for (long long t = f.done_tick; t <= h.rec[0]; ++t)   // the first event recorded at or behind the tick that completed it
    if (h.tick_of_ev[t % n] == t) return hip_ok(hipEventSynchronize(h.ev_tick[t % n]), "hs tick done");
  return false;
}
}  // extern "C"
namespace {
void host_stream_free(BeatriceBatch* b) {
  BeatriceBatch::HostStream& h = b->hs;
  for (hipEvent_t e : h.ev_in) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h.ev_out) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h.ev_tick) if (e) (void)hipEventDestroy(e);
  h.ev_in.clear(); h.ev_out.clear(); h.ev_tick.clear();
  if (h.s_in) (void)hipStreamDestroy(h.s_in);
  if (h.s_out) (void)hipStreamDestroy(h.s_out);
  if (h.d_in) (void)hipFree(h.d_in);
  if (h.d_out) (void)hipFree(h.d_out);
  if (h.h_in) (void)hipHostFree(h.h_in);
  if (h.h_out) (void)hipHostFree(h.h_out);
  h = BeatriceBatch::HostStream{};
}
}  // namespace
extern "C" {
int BeatriceBatch_EnableHostStreaming(BeatriceBatch* b, int enable) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::HostStream& h = b->hs;
  if ((enable != 0) == h.on) return 0;
  if (!enable) {
    if (!sync_all(b)) return -2;
    (void)hipStreamSynchronize(h.s_in); (void)hipStreamSynchronize(h.s_out);
    const int rc = tick_enable(b, false);
    if (rc) return rc;
    const int rb = BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0);
    host_stream_free(b);
    return rb;
  }
  if (b->H > tick::kMaxHops || b->io_slots > 0 || b->tk.on || b->pipelined || b->silent.on) return -1;   // (the in-order silent-block rule: switch it off first)  // one or two hops per step (buffers are [B][H x 160] -> [B][H x 240]); no other binding or pipelining
  h.n_slots = b->tk.plan.count() + 8;
  const size_t n_in = (size_t)b->B * b->H * B_IN_HOP, n_out = (size_t)b->B * b->H * B_OUT_HOP;
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&h.d_in), sizeof(float) * n_in * h.n_slots), "hs d_in") &&
            hip_ok(hipMalloc(reinterpret_cast<void**>(&h.d_out), sizeof(float) * n_out * h.n_slots), "hs d_out") &&
            hip_ok(hipHostMalloc(reinterpret_cast<void**>(&h.h_in), sizeof(float) * n_in * h.n_slots, hipHostMallocDefault), "hs h_in") &&
            hip_ok(hipHostMalloc(reinterpret_cast<void**>(&h.h_out), sizeof(float) * n_out * h.n_slots, hipHostMallocDefault), "hs h_out") &&
            hip_ok(hipMemset(h.d_in, 0, sizeof(float) * n_in * h.n_slots), "hs zero") &&
            hip_ok(hipStreamCreateWithFlags(&h.s_in, hipStreamNonBlocking), "hs s_in") &&
            hip_ok(hipStreamCreateWithFlags(&h.s_out, hipStreamNonBlocking), "hs s_out");
  h.ev_in.assign(h.n_slots, nullptr); h.ev_out.assign(h.n_slots, nullptr); h.ev_tick.assign(tick::kRing, nullptr);
  for (auto* v : {&h.ev_in, &h.ev_out, &h.ev_tick})
    for (hipEvent_t& e : *v) ok = ok && hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hs event");
  h.tick_of_ev.assign(tick::kRing, -1);
  h.mapped = bhip::meas_env("BEATRICE_HIP_HS_COPIES") == nullptr;   // A/B switch: copies on two more streams instead
  if (ok && h.mapped) std::memset(h.h_in, 0, sizeof(float) * n_in * h.n_slots);
  ok = ok && BeatriceBatch_BindResidentIO(b, h.mapped ? h.h_in : h.d_in, h.mapped ? h.h_out : h.d_out, h.n_slots) == 0 && tick_enable(b, true) == 0;
  if (!ok) { (void)tick_enable(b, false); (void)BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0); host_stream_free(b); return -2; }
  h.pending.clear();
  h.fed = 0;
  h.rec[0] = h.rec[1] = -1;
  h.on = true;
  return 0;
}
int BeatriceBatch_HostStreamDelay(const BeatriceBatch* b) { return b ? b->tk.plan.count() + 1 : 0; }
// in: [B][160] host; out: [B][240] host.  Returns 1 when `out` received the samples of the step fed
// BeatriceBatch_HostStreamDelay() calls ago, 0 while the pipeline is still filling (out untouched), < 0 on error.
int BeatriceBatch_StreamFrames(BeatriceBatch* b, const float* in, float* out) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::HostStream& h = b->hs;
  if (!h.on || !in || !out) return -1;
  const size_t n_in = (size_t)b->B * b->H * B_IN_HOP, n_out = (size_t)b->B * b->H * B_OUT_HOP;
  const int slot = b->io_host;  // the slot the tick about to be fed reads and, pipeline depth later, writes
  if (h.mapped) {
    // the slot's last readers (stage 9 of the step fed n_slots calls ago) are done: every call since the pipeline filled
    // has waited for a tick later than theirs before handing back its output
    std::memcpy(h.h_in + slot * n_in, in, sizeof(float) * n_in);
    if (!host_stream_tick(b, true)) return -2;
    h.pending.push_back({h.fed, slot, b->tk.last_feed_tick + b->tk.plan.count() - 1, false});
    h.fed += 1;
    const BeatriceBatch::HostStream::Pending& f = h.pending.front();
    if (!f.fetched || f.done_tick > b->tk.last_feed_tick - 2) return 0;   // keep two ticks queued on the device while the host waits
    if (!host_stream_wait(b, f)) return -2;
    std::memcpy(out, h.h_out + f.slot * n_out, sizeof(float) * n_out);
    h.pending.pop_front();
    return 1;
  }
  if (!hip_ok(hipEventSynchronize(h.ev_in[slot]), "hs in reuse")) return -2;  // the upload that last used this pinned slot (long done)
  std::memcpy(h.h_in + slot * n_in, in, sizeof(float) * n_in);
  // every reader of the device slot's old contents is done once the tick before the previous one is (the slot ring is
  // longer than the deepest reader's stage by more than that)
  if (h.rec[1] >= 0) (void)hipStreamWaitEvent(h.s_in, h.ev_tick[h.rec[1] % h.ev_tick.size()], 0);
  bool ok = hip_ok(hipMemcpyAsync(h.d_in + slot * n_in, h.h_in + slot * n_in, sizeof(float) * n_in, hipMemcpyHostToDevice, h.s_in), "hs upload");
  (void)hipEventRecord(h.ev_in[slot], h.s_in);
  (void)hipStreamWaitEvent(b->stream, h.ev_in[slot], 0);
  (void)hipStreamWaitEvent(b->stream, h.ev_out[slot], 0);  // the output slot this step will overwrite has been downloaded
  ok = ok && host_stream_tick(b, true);
  if (!ok) return -2;
  h.pending.push_back({h.fed, slot, b->tk.last_feed_tick + b->tk.plan.count() - 1, false});  // leaves the last stage that many ticks on
  h.fed += 1;
  const BeatriceBatch::HostStream::Pending& f = h.pending.front();
  if (!f.fetched || f.done_tick > b->tk.last_feed_tick - 2) return 0;   // hand back only what was enqueued for download two ticks ago
  if (!hip_ok(hipEventSynchronize(h.ev_out[f.slot]), "hs download")) return -2;
  std::memcpy(out, h.h_out + f.slot * n_out, sizeof(float) * n_out);
  h.pending.pop_front();
  return 1;
}
// After the last StreamFrames: hands back the next step still inside the pipeline (running ticks without input as
// needed); returns 1 with `out` filled, 0 when nothing is pending.
int BeatriceBatch_StreamFlush(BeatriceBatch* b, float* out) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::HostStream& h = b->hs;
  if (!h.on || !out) return -1;
  if (h.pending.empty()) return 0;
  const size_t n_out = (size_t)b->B * b->H * B_OUT_HOP;
  while (!h.pending.front().fetched)
    if (!host_stream_tick(b, false)) return -2;
  const BeatriceBatch::HostStream::Pending f = h.pending.front();
  if (!host_stream_wait(b, f)) return -2;
  std::memcpy(out, h.h_out + f.slot * n_out, sizeof(float) * n_out);
  h.pending.pop_front();
  return 1;
}

