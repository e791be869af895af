// abi.hip -- the 1-stream C-ABI of include/beatrice_abi.h on top of the HIP modules.
//
// Each entry point cites the reference declaration it replaces (reference
// lib/beatricelib/beatrice.h) and the reference call site that defines its contract (reference
// src/common/processor_core_2.cc).  Conventions of the boundary:
//   * only the file readers can fail (Beatrice_ErrorCode); every other entry returns void and, on
//     an internal HIP failure, leaves zeros in its outputs -- never throws, never blocks unboundedly;
//   * per-hop calls are synchronous: outputs are valid on return (the host consumes them on the
//     audio thread right away, processor_core_2.cc:184-255);
//   * per-hop calls do not allocate: device state, pinned staging and the HIP stream are created in
//     Create*Context1, which the host calls from non-real-time threads (processor_core_2.cc:259-266).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_set>
#include <vector>

#include "abi_objects.h"

using namespace bhip;

namespace bhip {

// ---- model files (MODEL_SPEC section 5); error mapping = reference beatrice.h:30-37 -----------
static const uint32_t kMagic = 0x43525442u, kVersion = 1u;

Beatrice_ErrorCode parse_model_bytes(const unsigned char* bytes, size_t size, uint32_t kind, long expect_floats,
                                     std::vector<float>* out) {
  if (size < 16) return Beatrice_kFileTooSmall;
  uint32_t hdr[4];
  std::memcpy(hdr, bytes, 16);
  if (hdr[0] != kMagic || hdr[1] != kind || hdr[2] != kVersion) return Beatrice_kInvalidFileSize;
  const size_t payload = size - 16;
  if (expect_floats >= 0) {
    if (payload < (size_t)expect_floats * 4) return Beatrice_kFileTooSmall;
    if (payload > (size_t)expect_floats * 4) return Beatrice_kFileTooLarge;
  }
  if (payload % 4 != 0 || (size_t)hdr[3] * 4 != payload) return Beatrice_kInvalidFileSize;
  out->resize(payload / 4);
  std::memcpy(out->data(), bytes + 16, payload);
  return Beatrice_kSuccess;
}

Beatrice_ErrorCode read_model_file(const char* path, uint32_t kind, long expect_floats, std::vector<float>* out) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) return Beatrice_kFileOpenError;
  const std::streamoff size = f.tellg();
  if (size < 0) return Beatrice_kFileOpenError;
  std::vector<unsigned char> bytes((size_t)size);
  f.seekg(0);
  if (size > 0 && !f.read(reinterpret_cast<char*>(bytes.data()), size)) return Beatrice_kFileOpenError;
  return parse_model_bytes(bytes.data(), bytes.size(), kind, expect_floats, out);
}

template <class Obj>
static Beatrice_ErrorCode install(Obj* m, std::vector<float>& host) {
  const DeviceScope dev_(m->device);
  m->loaded = false;
  decltype(m->w)::pack_host(host.data());  // GEMM tensors -> MFMA-fragment order
  if (!m->blob.upload(host.data(), host.size())) return Beatrice_kFileOpenError;  // device failure
  m->w.bind(m->blob.d);
  m->loaded = true;
  return Beatrice_kSuccess;
}

namespace { thread_local int t_target_device = -1; }
int target_device() {
  if (t_target_device >= 0) return t_target_device;
  int cur = 0;
  return hipGetDevice(&cur) == hipSuccess ? cur : -1;   // (-1: no usable GPU; the object is created unhealthy)
}
void set_target_device(int d) { t_target_device = d; }

bool make_stream(hipStream_t* s) { BHIP_TRY(hipStreamCreateWithFlags(s, hipStreamNonBlocking)); return true; }

// Completion wait of the synchronous per-hop calls.  They run on the host's audio thread, which has nothing else
// to do until the ~100 us of device work are done: polling the stream returns a few microseconds after the last
// copy lands, a blocking hipStreamSynchronize only after the runtime's wake-up path (352 -> 328 us per hop of
// three calls, profiles/r01_notes.md).  Falls back to the blocking wait if the work is unusually long.
bool wait_stream(hipStream_t s) {
  for (int spins = 0; spins < 200000; ++spins) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return true;
    if (e != hipErrorNotReady) return hip_ok(e, "stream query");
    __builtin_ia32_pause();
  }
  return hip_ok(hipStreamSynchronize(s), "sync");
}

// BEATRICE_HIP_HOP_IMMEDIATE=1: the step counter of a call travels in the kernels' arguments (stepc::immediate) instead of
// the mailbox in device memory; the launches are then plain ones (arguments change every call).
static bool hop_immediate() {
  static const bool on = bhip::meas_env("BEATRICE_HIP_HOP_IMMEDIATE") != nullptr;
  return on;
}
// Captures `enqueue` (copies + kernels on `s`) into g for this parameter blob / variant; nothing runs.  On any failure the
// capture is ended, the graph dropped and the pair remembered as "eager", so that the stream is never left in capture
// state and later hops fall back to plain launches instead of returning zeros.
template <class F>
static void capture_hop(HopGraph& g, const void* blob, int variant, hipStream_t s, F& enqueue) {
  g.drop();
  g.blob = blob;
  g.variant = variant;
  bool ok = hip_ok(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal), "begin capture");
  if (ok) {
    enqueue();
    ok = hip_ok(hipStreamEndCapture(s, &g.graph), "end capture") && g.graph != nullptr;   // (also leaves capture mode after a failed node)
    ok = ok && hip_ok(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0), "instantiate");
  }
  if (!ok) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
    g.exec = nullptr; g.graph = nullptr;
    g.eager = true;
    (void)hipGetLastError();  // the failure is handled: do not let it fail the eager launches that follow
  }
}
// Runs `enqueue` through the context's captured graph; captures it first when there is none for this parameter blob /
// variant.  The key is the DEVICE BLOB the captured kernels read (not the model object's address): Read*Parameters on the
// same object frees and re-allocates the blob, and a graph holding the old pointers must not be replayed.
// Round 5: PLAIN LAUNCHES are the default.  With each module's convolutions in one team launch a call is 4-6 kernels + 2-3 copies,
// and replaying them as a hipGraph measures ~10 us per hop SLOWER than enqueuing them (p50 285 vs 275 us over three A/B rounds,
// profiles/r05_notes.md section 10; with 49 kernels per hop, rounds 1-3, the graph won).  BEATRICE_HIP_HOP_GRAPH=1: the graphs.
static bool hop_graphs() {
  static const bool on = std::getenv("BEATRICE_HIP_HOP_GRAPH") != nullptr;
  return on;
}
template <class F>
static bool run_hop(HopGraph& g, const void* blob, int variant, hipStream_t s, F enqueue) {
  if (!hop_graphs() || hop_immediate()) { enqueue(); return hip_ok(hipGetLastError(), "hop launch"); }
  if (g.blob != blob || g.variant != variant || (!g.exec && !g.eager)) capture_hop(g, blob, variant, s, enqueue);
  if (!g.exec) { enqueue(); return hip_ok(hipGetLastError(), "hop launch"); }
  BHIP_TRY(hipGraphLaunch(g.exec, s));
  return true;
}

// ---- Completion of a per-hop call without a call into the runtime -----------------------------------------------------------------------
// Every call's LAST kernel writes the call's results into the pinned block the host reads and then, behind a system-scope fence, the call's
// sequence word (sent down with the input copy) -- the waveform generator's tail and the pitch estimator's head do it themselves, the phone
// vector goes through publish_kernel below, in the place of the copy command.  The host polls that word in its own memory.  Round 1's poll of
// hipStreamQuery returned a few microseconds after the last copy command had landed AND had the runtime retire the call's commands on the
// critical path (~1.5 us each: 10 us for a finished stream of seven); that bookkeeping now happens inside the next call's enqueues, beside
// device work.  A word that does not arrive (a failed launch, a device fault) ends in the stream wait every call had before.
static __global__ __launch_bounds__(256) void publish_kernel(const float* __restrict__ src, float* __restrict__ dst, const int n, const int* __restrict__ seq, int* flag) {
  const int t = threadIdx.x;
  if (t < n) { dst[t] = src[t]; __threadfence_system(); }
  __syncthreads();
  if (t == 0) __hip_atomic_store(flag, *seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static bool flag_wait(const int* flag_word, const int seq, hipStream_t s) {
  const volatile int* flag = flag_word;
  for (long spins = 0; spins < 300000; ++spins) {
    if (*flag == seq) return true;
    __builtin_ia32_pause();
  }
  return wait_stream(s) && *flag == seq;
}

// The bookkeeping left behind, done where it costs nothing: a call that has enqueued its work and has 40-130 us of device time to wait for first asks the runtime
// about the OTHER contexts' streams this thread completed calls on (hipStreamQuery on a finished stream retires its commands).  Left entirely to the enqueues the
// runtime catches up in bursts -- every 32-64 hops a few hops 30-45 us longer (p99 233 -> 250 us; profiles/r06_notes.md section 10).
namespace {
std::shared_mutex g_pair_mu;   // the registries below and the pairing pointers of the pre-execution (shared: a call looking at them; exclusive: create / destroy / pair)
std::unordered_set<hipStream_t> g_live_streams;
thread_local hipStream_t t_dirty[8];
thread_local int t_n_dirty = 0;
}
static void mark_dirty(hipStream_t s) {
  for (int i = 0; i < t_n_dirty; ++i) if (t_dirty[i] == s) return;
  if (t_n_dirty < 8) t_dirty[t_n_dirty++] = s;
}
static void housekeep(hipStream_t own) {
  if (t_n_dirty == 0) return;
  std::shared_lock<std::shared_mutex> g(g_pair_mu, std::try_to_lock);   // (an audio thread never waits for a create / destroy / reload elsewhere: next call then)
  if (!g.owns_lock()) return;
  int kept = 0;
  for (int i = 0; i < t_n_dirty; ++i) {
    const hipStream_t s = t_dirty[i];
    if (s != own) {
      if (!g_live_streams.count(s)) continue;                  // (its context is gone)
      if (hipStreamQuery(s) != hipErrorNotReady) continue;     // retired (or failed: its own calls will see that)
    }
    t_dirty[kept++] = s;
  }
  t_n_dirty = kept;
  (void)hipGetLastError();   // (hipErrorNotReady is not this call's failure)
}
static void stream_born(hipStream_t s) { if (s) { std::unique_lock<std::shared_mutex> g(g_pair_mu); g_live_streams.insert(s); } }
static void stream_gone(hipStream_t s) { if (s) { std::unique_lock<std::shared_mutex> g(g_pair_mu); g_live_streams.erase(s); } }

// ---- The pitch call runs BESIDE the phone call (pre-execution) ----------------------------------------------------------------------
// The reference's hop is ExtractPhone1(x) -> EstimatePitch1(x) -> GenerateWaveform1 (processor_core_2.cc:184,188,253): two independent
// modules over the SAME 160 samples, one after the other because a CPU has nothing to gain from anything else.  Here the two are device
// work on two streams, and the per-hop latency is the sum of what could overlap.  So, once a pitch context has been seen to be called
// right after a phone context with the same samples (same thread), the phone call also enqueues the pitch context's hop for ITS input on
// the pitch context's own stream, and returns when the phone features are there, as always.  EstimatePitch1 then finds its hop done or
// under way: if it arrives with the same 160 samples, bin range, estimator and parameters, it waits for that hop and hands out its
// results; if anything differs, the pre-executed hop is dropped and the call runs as it always did.
// Dropping is exact: a hop writes slot `hop` of the context's rings (rewritten by the real hop), the previous bin (put back from
// committed_prev_q) and this hop's granule tags of the team launch (the real hop then runs the per-layer launches, redo_plain: the same
// values, tests/test_gpu_realtime_contract.py).  The hop counter only moves when a hop counts.  A context whose pre-executions keep being
// dropped stops getting them (spec_off).  BEATRICE_HIP_NO_SPECULATION=1 turns the whole mechanism off (include/beatrice_batch.h).
namespace {
std::unordered_set<const void*> g_live_phone, g_live_pitch_models;   // (behind g_pair_mu)
thread_local Beatrice20rc0_PhoneContext1* t_last_phone = nullptr;   // the phone context this thread called last (a candidate partner)
}
static bool speculation_on() {
  static const bool on = std::getenv("BEATRICE_HIP_NO_SPECULATION") == nullptr;
  return on && !hop_graphs() && !hop_immediate();
}
// the hop's input and mailbox (counter | lowest bin | highest bin) into the pinned block; the counter is NOT advanced here
static void pitch_stage(Beatrice20rc0_PitchContext1* ctx, const float* input) {
  float* h_in = ctx->h_io;
  std::memcpy(h_in, input, sizeof(float) * B_IN_HOP);
  int* mb = reinterpret_cast<int*>(h_in + B_IN_HOP);
  mb[0] = ctx->hop_count; mb[1] = ctx->min_q; mb[2] = ctx->max_q;
  mb[3] = ++ctx->seq;   // (comes back as the last word the head kernel writes into the pinned result block: pitch_wait)
}
// input copy, the module's kernels, result copy on the context's stream; plain: the per-layer launches instead of the team launch
static void pitch_enqueue(const Beatrice20rc0_PitchEstimator* m, Beatrice20rc0_PitchContext1* ctx, const bool plain) {
  float* h_in = ctx->h_io;
  (void)hipMemcpyAsync(ctx->st.d_in, h_in, sizeof(float) * (B_IN_HOP + kMailboxWords), hipMemcpyHostToDevice, ctx->stream);
  const bool team_was_off = ctx->st.team_off;
  if (plain) ctx->st.team_off = true;   // (for this one hop; the caller holds the context)
  pitch_forward(m->w, ctx->st, ctx->stream);
  ctx->st.team_off = team_was_off;
  // (no result copy: the head kernel writes the four features, the raw bin and then the call's sequence word into the pinned block itself, PitchState::h_result)
}
static bool pitch_wait(Beatrice20rc0_PitchContext1* ctx) {   // (flag_wait above: the head kernel's sequence word)
  return flag_wait(reinterpret_cast<const int*>(ctx->h_io + B_IN_HOP + kMailboxWords) + 5, ctx->seq, ctx->stream);
}
// a pre-executed hop that does not count (the stream is idle): the previous bin back, the hop's tags are used up
static void spec_drop(Beatrice20rc0_PitchContext1* ctx) {
  int* h_prev = reinterpret_cast<int*>(ctx->h_io + B_IN_HOP + kMailboxWords + 6);
  *h_prev = ctx->committed_prev_q;
  (void)hipMemcpyAsync(ctx->st.d_prev_q, h_prev, sizeof(int), hipMemcpyHostToDevice, ctx->stream);
  ctx->redo_plain = true;
  ++ctx->spec_misses;
  if (++ctx->spec_strikes >= 8) ctx->spec_off = true;
}
// called by ExtractPhone1 with its own work enqueued and not yet waited for: the partner's hop for the same samples
static void spec_launch(Beatrice20rc0_PhoneContext1* phone) {
  t_last_phone = phone;
  if (!phone->paired_pitch) return;
  std::shared_lock<std::shared_mutex> g(g_pair_mu, std::try_to_lock);   // (held exclusively while an estimator is reloaded or a context comes or goes: no pre-execution for this hop)
  if (!g.owns_lock()) return;
  Beatrice20rc0_PitchContext1* q = phone->paired_pitch;
  if (!q || !q->ok || q->device != phone->device || !q->spec_mu.try_lock()) return;
  std::lock_guard<std::mutex> own(q->spec_mu, std::adopt_lock);
  if (q->spec_pending) {   // the last one was never asked for
    (void)(q->spec_launch_ok ? pitch_wait(q) : wait_stream(q->stream));
    q->spec_pending = false;
    if (team_timed_out(q->st)) { team_recover(q->st, q->stream); (void)hipMemsetAsync(q->st.d_prev_q, 0, sizeof(int), q->stream); q->committed_prev_q = 0; q->redo_plain = false; ++q->spec_misses; }
    else spec_drop(q);
  }
  const Beatrice20rc0_PitchEstimator* m = q->spec_model;
  if (q->spec_off || !m || !g_live_pitch_models.count(m) || !m->loaded || m->device != q->device || m->generation != q->spec_generation) return;
  pitch_stage(q, phone->h_io);
  q->spec_min_q = q->min_q; q->spec_max_q = q->max_q;
  pitch_enqueue(m, q, q->redo_plain);
  q->spec_launch_ok = hipGetLastError() == hipSuccess;   // (a hop whose launches failed is never claimed)
  q->spec_pending = true;
}
// EstimatePitch1 without a partner yet: was this thread's last phone call given the same samples?
static void spec_learn(Beatrice20rc0_PitchContext1* ctx, const float* input) {
  Beatrice20rc0_PhoneContext1* p = t_last_phone;
  if (!p || ctx->paired_phone || ctx->spec_off) return;
  std::unique_lock<std::shared_mutex> g(g_pair_mu, std::try_to_lock);   // (busy: the next hop will do)
  if (!g.owns_lock() || !g_live_phone.count(p) || !p->ok || p->device != ctx->device || p->paired_pitch || std::memcmp(p->h_io, input, sizeof(float) * B_IN_HOP) != 0) return;
  p->paired_pitch = ctx;
  ctx->paired_phone = p;
}

// the same for callers in other translation units (legacy.hip)
bool run_hop_graph(HopGraph& g, const void* blob, int variant, hipStream_t s, void (*enqueue)(void*), void* ctx) {
  return run_hop(g, blob, variant, s, [&] { enqueue(ctx); });
}

}  // namespace bhip

extern "C" {

// ================================ phone extractor ==============================================
// ref beatrice.h:230-232
Beatrice20rc0_PhoneExtractor* Beatrice20rc0_CreatePhoneExtractor(void) { return new Beatrice20rc0_PhoneExtractor(); }
void Beatrice20rc0_DestroyPhoneExtractor(Beatrice20rc0_PhoneExtractor* m) {
  if (!m) return;
  const DeviceScope dev_(m->device);
  m->blob.release();
  delete m;
}
// ref beatrice.h:235-238; caller processor_core_2.cc:302-307
Beatrice_ErrorCode Beatrice20rc0_ReadPhoneExtractorParameters(Beatrice20rc0_PhoneExtractor* m, const char* path) {
  std::vector<float> host;
  const Beatrice_ErrorCode e = read_model_file(path, KIND_PHONE, (long)PhoneWeights::n_floats(), &host);
  return e ? e : install(m, host);
}
// ref beatrice.h:233-234; callers processor_core_2.h:36, processor_core_2.cc:263 (non-real-time threads: everything a
// hop or a setter may need later -- state, pinned staging, the codebook pool -- is allocated here)
Beatrice20rc0_PhoneContext1* Beatrice20rc0_CreatePhoneContext1(void) {
  auto* c = new Beatrice20rc0_PhoneContext1();
  const DeviceScope dev_(c->device);
  constexpr size_t kCbFloats = (size_t)B_CODEBOOK * B_PHONE_CH, kSlotFloats = kCbFloats + B_CODEBOOK;
  c->ok = make_stream(&c->stream) && c->st.create(1, 1, nullptr) &&
          hip_ok(hipHostMalloc(reinterpret_cast<void**>(&c->h_io), sizeof(float) * (B_IN_HOP + kMailboxWords + B_PHONE_CH + 1), hipHostMallocDefault),
                 "hipHostMalloc") &&
          hip_ok(hipMalloc(reinterpret_cast<void**>(&c->d_pool), sizeof(float) * kSlotFloats * kCodebookPool), "cb pool") &&
          hip_ok(hipMalloc(reinterpret_cast<void**>(&c->d_cb_stage), sizeof(float) * 2 * kCbFloats), "cb stage") &&
          hip_ok(hipHostMalloc(reinterpret_cast<void**>(&c->h_cb_stage), sizeof(float) * 2 * kCbFloats, hipHostMallocDefault), "cb stage host") &&
          hip_ok(hipEventCreateWithFlags(&c->stage_done[0], hipEventDisableTiming), "ev") &&
          hip_ok(hipEventCreateWithFlags(&c->stage_done[1], hipEventDisableTiming), "ev");
  if (c->ok) {
    std::memset(c->h_io, 0, sizeof(float) * (B_IN_HOP + kMailboxWords + B_PHONE_CH + 1));
    for (int i = 0; i < kCodebookPool; ++i) {
      c->pool[i].d_cbT = c->d_pool + (size_t)i * kSlotFloats;
      c->pool[i].d_cnorm = c->pool[i].d_cbT + kCbFloats;
    }
    // the kernels read the step counter and the k-NN selectors from the mailbox behind the audio; they arrive with the input copy
    c->st.hop = c->st.hop_in = c->st.hop_mailbox;
    c->own_sel[0] = c->st.d_vqk; c->own_sel[1] = (void*)c->st.d_cbT; c->own_sel[2] = (void*)c->st.d_cnorm;
    c->st.d_vqk = c->st.hop_mailbox + 1;
    c->st.d_cbT = reinterpret_cast<const float**>(c->st.hop_mailbox + 2);
    c->st.d_cnorm = reinterpret_cast<const float**>(c->st.hop_mailbox + 4);
  }
  c->st.advance_hop = false;
  c->st.skip_vq = true;  // k = 0 until SetVQNumNeighbors says otherwise
  { std::unique_lock<std::shared_mutex> g(g_pair_mu); g_live_phone.insert(c); }
  stream_born(c->stream);
  return c;
}
void Beatrice20rc0_DestroyPhoneContext1(Beatrice20rc0_PhoneContext1* c) {
  if (!c) return;
  {
    std::unique_lock<std::shared_mutex> g(g_pair_mu);
    g_live_phone.erase(c);
    g_live_streams.erase(c->stream);
    if (c->paired_pitch) { c->paired_pitch->paired_phone = nullptr; c->paired_pitch = nullptr; }
  }
  const DeviceScope dev_(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (HopGraph& g : c->hop_graph) g.drop();
  if (c->own_sel[0]) {
    c->st.d_vqk = static_cast<int*>(c->own_sel[0]);
    c->st.d_cbT = static_cast<const float**>(c->own_sel[1]);
    c->st.d_cnorm = static_cast<const float**>(c->own_sel[2]);
  }
  c->st.destroy();
  if (c->d_pool) (void)hipFree(c->d_pool);
  if (c->d_cb_stage) (void)hipFree(c->d_cb_stage);
  if (c->h_cb_stage) (void)hipHostFree(c->h_cb_stage);
  for (hipEvent_t e : c->stage_done) if (e) (void)hipEventDestroy(e);
  if (c->h_io) (void)hipHostFree(c->h_io);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
// ref beatrice.h:239-242; caller processor_core_2.cc:585-590 (clamped there to 0..8).  Host-side only: k travels to
// the device with the next hop's input.
void Beatrice20rc0_SetVQNumNeighbors(Beatrice20rc0_PhoneContext1* ctx, int k) {
  if (!ctx || !ctx->ok) return;
  ctx->vq_k = k < 0 ? 0 : (k > B_CODEBOOK ? B_CODEBOOK : k);
  ctx->st.skip_vq = ctx->vq_k == 0;  // no k-NN launch while the codebook is unused
}
// Fingerprint of a caller-owned codebook: 96 words sampled over the table.  Cheap enough for a per-hop call, and a
// table rewritten in place by a model reload differs somewhere in the sample with overwhelming probability.
static uint64_t codebook_print(const float* cb) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(cb);
  constexpr size_t n = (size_t)B_CODEBOOK * B_PHONE_CH;
  uint64_t h = 1469598103934665603ull;
  auto mix = [&h](uint32_t v) { h = (h ^ v) * 1099511628211ull; };
  for (size_t i = 0; i < 16; ++i) { mix(w[i]); mix(w[n - 1 - i]); }
  for (size_t i = 0; i < 64; ++i) mix(w[(i * 1021 + 389) % n]);
  return h;
}
// ref beatrice.h:318-322.  The host passes a pointer into its own table and, in morph mode, calls this every hop on
// the audio thread (processor_core_2.cc:118-121): O(1) when the table is in the context's pool, otherwise ONE
// stream-ordered upload (pinned staging -> device -> transpose + norms) into the least recently used slot -- no
// allocation and no wait either way (the next ExtractPhone1 runs behind it on the context's stream).
void Beatrice20rc0_SetCodebook(Beatrice20rc0_PhoneContext1* ctx, const float* codebook) {
  if (!ctx || !ctx->ok || !codebook) return;
  const DeviceScope dev_(ctx->device);
  const uint64_t print = codebook_print(codebook);
  CodebookEntry* hit = nullptr;
  CodebookEntry* lru = &ctx->pool[0];
  for (CodebookEntry& e : ctx->pool) {
    if (e.host == codebook && e.print == print) { hit = &e; break; }
    if (e.host == codebook) { lru = &e; break; }  // same storage, new contents: refresh this slot
    if (e.last_use < lru->last_use) lru = &e;
  }
  if (!hit) {
    constexpr size_t n = (size_t)B_CODEBOOK * B_PHONE_CH;
    const int sb = ctx->stage_next;
    ctx->stage_next ^= 1;
    if (ctx->stage_busy[sb]) { (void)hipEventSynchronize(ctx->stage_done[sb]); ctx->stage_busy[sb] = false; }  // two uploads ago: long done
    std::memcpy(ctx->h_cb_stage + sb * n, codebook, n * sizeof(float));
    float* d_raw = ctx->d_cb_stage + sb * n;
    if (!hip_ok(hipMemcpyAsync(d_raw, ctx->h_cb_stage + sb * n, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream), "cb upload")) return;
    codebook_prepare(d_raw, 1, lru->d_cbT, lru->d_cnorm, ctx->stream);
    ctx->stage_busy[sb] = hip_ok(hipEventRecord(ctx->stage_done[sb], ctx->stream), "cb ev");
    lru->host = codebook;
    lru->print = print;
    hit = lru;
  }
  hit->last_use = ++ctx->use_clock;
  ctx->sel_cbT = hit->d_cbT;
  ctx->sel_cnorm = hit->d_cnorm;
}
// SetCodebook recognises a table by its address and a 96-word fingerprint (a per-hop call cannot hash 256 KB).  A caller that
// rewrites a table IN PLACE in words the fingerprint does not sample (a model reload into the same storage samples all of
// them with overwhelming probability; a deliberate edit may not) says so here, off the audio thread: entries made from
// `codebook` (NULL: every entry) are forgotten, and the next SetCodebook of that address uploads the table again.
extern "C" void BeatriceHip_InvalidateCodebook(Beatrice20rc0_PhoneContext1* ctx, const float* codebook) {
  if (!ctx || !ctx->ok) return;
  for (CodebookEntry& e : ctx->pool)
    if (!codebook || e.host == codebook) { e.host = nullptr; e.print = 0; e.last_use = 0; }
}
// ref beatrice.h:243-247; caller processor_core_2.cc:183-185
void Beatrice20rc0_ExtractPhone1(const Beatrice20rc0_PhoneExtractor* m, const float* input, float* output,
                                 Beatrice20rc0_PhoneContext1* ctx) {
  std::memset(output, 0, sizeof(float) * B_PHONE_CH);
  if (!m || !m->loaded || !ctx || !ctx->ok || m->device != ctx->device) return;   // (model and context must live on one GPU)
  const DeviceScope dev_(ctx->device);
  float* h_in = ctx->h_io;
  float* h_out = ctx->h_io + B_IN_HOP + kMailboxWords;
  std::memcpy(h_in, input, sizeof(float) * B_IN_HOP);
  {  // mailbox: counter | k | codebook pointers
    int* mb = reinterpret_cast<int*>(h_in + B_IN_HOP);
    mb[0] = ctx->hop_count;
    mb[1] = ctx->sel_cbT ? ctx->vq_k : 0;
    std::memcpy(mb + 2, &ctx->sel_cbT, sizeof(float*));
    std::memcpy(mb + 4, &ctx->sel_cnorm, sizeof(float*));
    mb[6] = ++ctx->seq;   // (comes back behind the phone vector: flag_wait)
  }
  if (hop_immediate()) ctx->st.hop = ctx->st.hop_in = const_cast<int*>(stepc::immediate(ctx->hop_count));
  ctx->hop_count = hop_next(ctx->hop_count);
  auto enqueue = [&] {
    (void)hipMemcpyAsync(ctx->st.d_in, h_in, sizeof(float) * (B_IN_HOP + kMailboxWords), hipMemcpyHostToDevice, ctx->stream);
    phone_forward(m->w, ctx->st, ctx->stream);
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->st.d_phone, h_out, B_PHONE_CH, ctx->st.hop_mailbox + 6, reinterpret_cast<int*>(h_out + B_PHONE_CH));
  };
  const int variant = ctx->st.skip_vq ? 0 : 1;
  // Both variants (k-NN launch present / absent) are captured at the first hop with a given parameter blob, so that a
  // later SetVQNumNeighbors toggle on the audio thread finds its graph ready instead of spending 1-2 ms on a capture.
  HopGraph& other = ctx->hop_graph[variant ^ 1];
  if (other.blob != m->blob.d && !hop_immediate() && hop_graphs()) {
    const bool keep = ctx->st.skip_vq;
    ctx->st.skip_vq = !keep;
    capture_hop(other, m->blob.d, variant ^ 1, ctx->stream, enqueue);
    ctx->st.skip_vq = keep;
  }
  // the partner pitch context's hop for the same samples, beside this one ("pre-execution" above): enqueued right behind this call's team launch, whose
  // ~45 us cover the host time of those launches and of this call's remaining ones
  if (speculation_on()) { ctx->st.after_convs = [](void* c) { spec_launch(static_cast<Beatrice20rc0_PhoneContext1*>(c)); }; ctx->st.after_convs_arg = ctx; }
  bool ok = run_hop(ctx->hop_graph[variant], m->blob.d, variant, ctx->stream, enqueue);
  ctx->st.after_convs = nullptr;
  housekeep(ctx->stream);
  ok = (ok ? flag_wait(reinterpret_cast<const int*>(h_out + B_PHONE_CH), ctx->seq, ctx->stream) : wait_stream(ctx->stream)) && ok;
  mark_dirty(ctx->stream);
  if (team_timed_out(ctx->st)) {   // a team launch gave a wait up: zeros for this call (as for any internal failure), the per-layer launches from the next one on (engine.h team_recover)
    ok = false;
    team_recover(ctx->st, ctx->stream);
    for (HopGraph& g : ctx->hop_graph) g.drop();   // (they hold the team launch)
  }
  if (ok) std::memcpy(output, h_out, sizeof(float) * B_PHONE_CH);
}

// ================================ pitch estimator ==============================================
// ref beatrice.h:249-251
Beatrice20rc0_PitchEstimator* Beatrice20rc0_CreatePitchEstimator(void) {
  auto* m = new Beatrice20rc0_PitchEstimator();
  std::unique_lock<std::shared_mutex> g(g_pair_mu);
  g_live_pitch_models.insert(m);
  return m;
}
void Beatrice20rc0_DestroyPitchEstimator(Beatrice20rc0_PitchEstimator* m) {
  if (!m) return;
  { std::unique_lock<std::shared_mutex> g(g_pair_mu); g_live_pitch_models.erase(m); }   // (no pre-execution starts with it from here on; hipFree below waits for those under way)
  const DeviceScope dev_(m->device);
  m->blob.release();
  delete m;
}
// ref beatrice.h:254-257; caller processor_core_2.cc:308-313
Beatrice_ErrorCode Beatrice20rc0_ReadPitchEstimatorParameters(Beatrice20rc0_PitchEstimator* m, const char* path) {
  std::vector<float> host;
  const Beatrice_ErrorCode e = read_model_file(path, KIND_PITCH, (long)PitchWeights::n_floats(), &host);
  if (e) return e;
  std::unique_lock<std::shared_mutex> g(g_pair_mu);   // (a pre-executed hop is claimed only by the parameters it ran with)
  ++m->generation;
  return install(m, host);
}
// ref beatrice.h:252-253
Beatrice20rc0_PitchContext1* Beatrice20rc0_CreatePitchContext1(void) {
  auto* c = new Beatrice20rc0_PitchContext1();
  const DeviceScope dev_(c->device);
  c->ok = make_stream(&c->stream) && c->st.create(1, 1, nullptr, false) &&
          hip_ok(hipHostMalloc(reinterpret_cast<void**>(&c->h_io), sizeof(float) * (B_IN_HOP + kMailboxWords + 8), hipHostMallocDefault), "hipHostMalloc");
  if (c->ok) {  // step counter and bin range arrive with the input copy (mailbox behind the audio)
    c->st.hop = c->st.hop_in = c->st.hop_mailbox;
    c->own_sel[0] = c->st.d_min_q; c->own_sel[1] = c->st.d_max_q;
    c->st.d_min_q = c->st.hop_mailbox + 1;
    c->st.d_max_q = c->st.hop_mailbox + 2;
    std::memset(c->h_io, 0, sizeof(float) * (B_IN_HOP + kMailboxWords + 8));
    c->st.h_result = c->h_io + B_IN_HOP + kMailboxWords;   // the head kernel writes the results and the call's sequence word here itself (pitch_wait)
  }
  c->st.advance_hop = false;
  stream_born(c->stream);
  return c;
}
void Beatrice20rc0_DestroyPitchContext1(Beatrice20rc0_PitchContext1* c) {
  if (!c) return;
  {
    std::unique_lock<std::shared_mutex> g(g_pair_mu);   // (no phone call is enqueuing here while this is held)
    if (c->paired_phone) { c->paired_phone->paired_pitch = nullptr; c->paired_phone = nullptr; }
    g_live_streams.erase(c->stream);
  }
  const DeviceScope dev_(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->hop_graph.drop();
  if (c->own_sel[0]) { c->st.d_min_q = static_cast<int*>(c->own_sel[0]); c->st.d_max_q = static_cast<int*>(c->own_sel[1]); }
  c->st.destroy();
  if (c->h_io) (void)hipHostFree(c->h_io);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
static int clamp_bin(int q) { return q < 1 ? 1 : (q > B_PITCH_BINS - 1 ? B_PITCH_BINS - 1 : q); }
// ref beatrice.h:258-265; callers processor_core_2.cc:561-583
void Beatrice20rc0_SetMinQuantizedPitch(Beatrice20rc0_PitchContext1* ctx, int q) {
  if (!ctx || !ctx->ok) return;
  ctx->min_q = clamp_bin(q);  // host-side only: travels with the next hop's input
}
void Beatrice20rc0_SetMaxQuantizedPitch(Beatrice20rc0_PitchContext1* ctx, int q) {
  if (!ctx || !ctx->ok) return;
  ctx->max_q = clamp_bin(q);
}
// ref beatrice.h:266-271; caller processor_core_2.cc:186-189
void Beatrice20rc0_EstimatePitch1(const Beatrice20rc0_PitchEstimator* m, const float* input, int* out_q, float* out_feat,
                                  Beatrice20rc0_PitchContext1* ctx) {
  *out_q = 1;
  std::memset(out_feat, 0, sizeof(float) * 4);
  if (!m || !m->loaded || !ctx || !ctx->ok || m->device != ctx->device) return;
  const DeviceScope dev_(ctx->device);
  std::lock_guard<std::mutex> own(ctx->spec_mu);
  float* h_in = ctx->h_io;
  float* h_feat = ctx->h_io + B_IN_HOP + kMailboxWords;
  int* h_q = reinterpret_cast<int*>(ctx->h_io + B_IN_HOP + kMailboxWords + 4);
  bool ok = false, have = false;
  if (ctx->spec_pending) {   // the phone call before this one enqueued this hop for ITS samples (pre-execution, above)
    ctx->spec_pending = false;
    ok = ctx->spec_launch_ok ? pitch_wait(ctx) : wait_stream(ctx->stream);
    have = ctx->spec_launch_ok && ok && m == ctx->spec_model && m->generation == ctx->spec_generation && ctx->min_q == ctx->spec_min_q && ctx->max_q == ctx->spec_max_q &&
           std::memcmp(input, h_in, sizeof(float) * B_IN_HOP) == 0;
    if (have) {
      ctx->redo_plain = false;
      if ((++ctx->spec_hits & 255) == 0 && ctx->spec_strikes > 0) --ctx->spec_strikes;
    } else if (team_timed_out(ctx->st)) {   // (the hop that does not count gave a wait up: the context restarts as after any time-out, then runs this hop)
      team_recover(ctx->st, ctx->stream);
      (void)hipMemsetAsync(ctx->st.d_prev_q, 0, sizeof(int), ctx->stream);
      ctx->committed_prev_q = 0;
      ctx->redo_plain = false;
      ++ctx->spec_misses;
    } else {
      spec_drop(ctx);
    }
  }
  if (!have) {
    pitch_stage(ctx, input);
    if (hop_immediate()) ctx->st.hop = ctx->st.hop_in = const_cast<int*>(stepc::immediate(ctx->hop_count));
    const bool plain = ctx->redo_plain;
    ok = run_hop(ctx->hop_graph, m->blob.d, 0, ctx->stream, [&] { pitch_enqueue(m, ctx, plain); });
    housekeep(ctx->stream);
    ok = (ok ? pitch_wait(ctx) : wait_stream(ctx->stream)) && ok;
    ctx->redo_plain = false;
    ctx->spec_model = m;
    ctx->spec_generation = m->generation;
    if (speculation_on() && !ctx->paired_phone) spec_learn(ctx, input);
  }
  ctx->hop_count = hop_next(ctx->hop_count);
  mark_dirty(ctx->stream);
  if (team_timed_out(ctx->st)) {
    ok = false;
    team_recover(ctx->st, ctx->stream);
    (void)hipMemsetAsync(ctx->st.d_prev_q, 0, sizeof(int), ctx->stream);   // (the one piece of the stream's state outside the rings)
    ctx->hop_graph.drop();
    ctx->committed_prev_q = 0;
  } else if (ok) {
    ctx->committed_prev_q = *h_q;
  }
  if (ok) { *out_q = *h_q; std::memcpy(out_feat, h_feat, sizeof(float) * 4); }
}
// Counters of the pre-execution for tests and tools: hops claimed / dropped so far; returns 1 while the context has a partner and gets pre-executions, else 0.
int BeatriceHip_PitchSpeculation(Beatrice20rc0_PitchContext1* ctx, long long* claimed, long long* dropped) {
  if (!ctx || !ctx->ok) return -1;
  std::lock_guard<std::mutex> own(ctx->spec_mu);
  if (claimed) *claimed = ctx->spec_hits;
  if (dropped) *dropped = ctx->spec_misses;
  return ctx->paired_phone != nullptr && !ctx->spec_off && speculation_on() ? 1 : 0;
}

// ================================ waveform generator ===========================================
// ref beatrice.h:292-294
Beatrice20rc0_WaveformGenerator* Beatrice20rc0_CreateWaveformGenerator(void) { return new Beatrice20rc0_WaveformGenerator(); }
void Beatrice20rc0_DestroyWaveformGenerator(Beatrice20rc0_WaveformGenerator* m) {
  if (!m) return;
  const DeviceScope dev_(m->device);
  m->blob.release();
  delete m;
}
// ref beatrice.h:297-300; caller processor_core_2.cc:314-319
Beatrice_ErrorCode Beatrice20rc0_ReadWaveformGeneratorParameters(Beatrice20rc0_WaveformGenerator* m, const char* path) {
  std::vector<float> host;
  const Beatrice_ErrorCode e = read_model_file(path, KIND_WAVE, (long)WaveWeights::n_floats(), &host);
  return e ? e : install(m, host);
}
// ref beatrice.h:295-296.  Inputs of one hop (phone 128 f32 | feat 4 f32 | bin 1 i32) are one
// contiguous device block so GenerateWaveform1 needs a single host-to-device copy.
Beatrice20rc0_WaveformContext1* Beatrice20rc0_CreateWaveformContext1(void) {
  auto* c = new Beatrice20rc0_WaveformContext1();
  const DeviceScope dev_(c->device);
  const size_t in_floats = B_PHONE_CH + 4 + 1 + 1 + 1;  // ... | step counter | sequence word
  c->ok = make_stream(&c->stream) &&
          hip_ok(hipMalloc(reinterpret_cast<void**>(&c->d_inputs), sizeof(float) * in_floats), "inputs") &&
          hip_ok(hipMemset(c->d_inputs, 0, sizeof(float) * in_floats), "inputs0") &&
          c->st.create(1, 1, 1, 1, 1, c->d_inputs, reinterpret_cast<int*>(c->d_inputs + B_PHONE_CH + 4), c->d_inputs + B_PHONE_CH) &&
          hip_ok(hipHostMalloc(reinterpret_cast<void**>(&c->h_io), sizeof(float) * (in_floats + B_OUT_HOP + 1), hipHostMallocDefault), "hipHostMalloc");
  c->st.hop = reinterpret_cast<int*>(c->d_inputs + B_PHONE_CH + 4 + 1);
  c->st.advance_hop = false;
  if (c->ok) {   // the tail kernel writes the hop's 240 samples straight into the pinned block, then the call's sequence word behind them (flag_wait)
    std::memset(c->h_io, 0, sizeof(float) * (in_floats + B_OUT_HOP + 1));
    c->dev_d_out = c->st.d_out;
    c->st.d_out = c->h_io + in_floats;
    c->st.h_flag = reinterpret_cast<int*>(c->h_io + in_floats + B_OUT_HOP);
    c->st.d_seq = reinterpret_cast<const int*>(c->d_inputs + B_PHONE_CH + 4 + 2);
    c->out_mapped = true;
  }
  stream_born(c->stream);
  return c;
}
void Beatrice20rc0_DestroyWaveformContext1(Beatrice20rc0_WaveformContext1* c) {
  if (!c) return;
  stream_gone(c->stream);
  const DeviceScope dev_(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->hop_graph.drop();
  if (c->out_mapped) { c->st.d_out = c->dev_d_out; c->out_mapped = false; }
  c->st.destroy();
  if (c->d_inputs) (void)hipFree(c->d_inputs);
  if (c->h_io) (void)hipHostFree(c->h_io);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
// ref beatrice.h:301-307; caller processor_core_2.cc:253-255
void Beatrice20rc0_GenerateWaveform1(const Beatrice20rc0_WaveformGenerator* m, const float* phone, const int* q,
                                     const float* feat, float* output, Beatrice20rc0_WaveformContext1* ctx) {
  std::memset(output, 0, sizeof(float) * B_OUT_HOP);
  if (!m || !m->loaded || !ctx || !ctx->ok || m->device != ctx->device) return;
  const DeviceScope dev_(ctx->device);
  const size_t in_floats = B_PHONE_CH + 4 + 1 + 1 + 1;
  float* h_in = ctx->h_io;
  float* h_out = ctx->h_io + in_floats;
  std::memcpy(h_in, phone, sizeof(float) * B_PHONE_CH);
  std::memcpy(h_in + B_PHONE_CH, feat, sizeof(float) * 4);
  std::memcpy(h_in + B_PHONE_CH + 4, q, sizeof(int));
  std::memcpy(h_in + B_PHONE_CH + 5, &ctx->hop_count, sizeof(int));
  ++ctx->seq;
  std::memcpy(h_in + B_PHONE_CH + 6, &ctx->seq, sizeof(int));
  if (hop_immediate()) ctx->st.hop = const_cast<int*>(stepc::immediate(ctx->hop_count));
  ctx->hop_count = hop_next(ctx->hop_count);
  bool ok = run_hop(ctx->hop_graph, m->blob.d, 0, ctx->stream, [&] {
    (void)hipMemcpyAsync(ctx->d_inputs, h_in, sizeof(float) * in_floats, hipMemcpyHostToDevice, ctx->stream);
    wave_forward(m->w, ctx->st, ctx->stream);   // (its last kernel writes h_out and the sequence word)
  });
  housekeep(ctx->stream);
  ok = (ok ? flag_wait(ctx->st.h_flag, ctx->seq, ctx->stream) : wait_stream(ctx->stream)) && ok;
  mark_dirty(ctx->stream);
  if (team_timed_out(ctx->st)) {
    ok = false;
    team_recover(ctx->st, ctx->stream);
    ctx->hop_graph.drop();
  }
  if (ok) std::memcpy(output, h_out, sizeof(float) * B_OUT_HOP);
}

// Test hook: makes the context's NEXT GenerateWaveform1 behave as if its team launch had timed out (engine.h team_recover) -- that
// call returns zeros, the context's rings restart from silence and the following calls run the per-layer launches.  -1: the
// context has no team launch.
int BeatriceHip_InjectTeamTimeout(Beatrice20rc0_WaveformContext1* ctx) {
  if (!ctx || !ctx->ok || !ctx->st.d_team_dead || ctx->st.team_off) return -1;
  *ctx->st.d_team_dead = 1;
  return 0;
}
// the same for the other two modules' contexts (their recovery paths: ExtractPhone1 / EstimatePitch1 above)
int BeatriceHip_InjectTeamTimeoutPhone(Beatrice20rc0_PhoneContext1* ctx) {
  if (!ctx || !ctx->ok || !ctx->st.d_team_dead || ctx->st.team_off) return -1;
  *ctx->st.d_team_dead = 1;
  return 0;
}
int BeatriceHip_InjectTeamTimeoutPitch(Beatrice20rc0_PitchContext1* ctx) {
  if (!ctx || !ctx->ok || !ctx->st.d_team_dead || ctx->st.team_off) return -1;
  *ctx->st.d_team_dead = 1;
  return 0;
}

// ================================ embedding setter =============================================
// ref beatrice.h:309-311
Beatrice20rc0_EmbeddingSetter* Beatrice20rc0_CreateEmbeddingSetter(void) { return new Beatrice20rc0_EmbeddingSetter(); }
void Beatrice20rc0_DestroyEmbeddingSetter(Beatrice20rc0_EmbeddingSetter* m) {
  if (!m) return;
  const DeviceScope dev_(m->device);
  m->blob.release();
  delete m;
}
// ref beatrice.h:314-317; caller processor_core_2.cc:320-325
Beatrice_ErrorCode Beatrice20rc0_ReadEmbeddingSetterParameters(Beatrice20rc0_EmbeddingSetter* m, const char* path) {
  std::vector<float> host;
  const Beatrice_ErrorCode e = read_model_file(path, KIND_EMBED, (long)EmbedWeights::n_floats(), &host);
  return e ? e : install(m, host);
}
// ref beatrice.h:312-313.  Pinned staging for every setter is allocated here, so that the setters themselves (the
// host calls SetAdditive / Register / SetKeyValue on its audio thread in morph mode, processor_core_2.cc:124-172)
// neither allocate nor wait: uploads and projections are enqueued on HIP streams, ordered by events.
Beatrice20rc0_EmbeddingContext* Beatrice20rc0_CreateEmbeddingContext(void) {
  auto* c = new Beatrice20rc0_EmbeddingContext();
  const DeviceScope dev_(c->device);
  const size_t n = (size_t)B_KV_LEN * B_KV_CH + 2 * 2 * B_HID + 2 * B_HID;
  const size_t nh = (size_t)B_KV_LEN * B_KV_CH + 4 * B_HID;
  c->ok = make_stream(&c->stream) && hip_ok(hipMalloc(reinterpret_cast<void**>(&c->d_block), sizeof(float) * n), "embed ctx") &&
          hip_ok(hipMemset(c->d_block, 0, sizeof(float) * n), "embed ctx0") &&
          hip_ok(hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), sizeof(float) * nh, hipHostMallocDefault), "embed stage") &&
          hip_ok(hipEventCreateWithFlags(&c->kv_uploaded, hipEventDisableTiming), "ev") &&
          hip_ok(hipEventCreateWithFlags(&c->kv_projected, hipEventDisableTiming), "ev");
  for (int i = 0; i < 4 && c->ok; ++i) c->ok = hip_ok(hipEventCreateWithFlags(&c->vec_sent[i], hipEventDisableTiming), "ev");
  (void)hipDeviceSynchronize();  // NULL-stream memset vs the context's non-blocking stream
  c->d_kv_raw = c->d_block;
  c->d_tmp = c->d_block + (size_t)B_KV_LEN * B_KV_CH;  // [4][256] upload slots: additive x2, formant x2
  c->d_add = c->d_tmp + 4 * B_HID;
  c->d_frm = c->d_add + B_HID;
  return c;
}
void Beatrice20rc0_DestroyEmbeddingContext(Beatrice20rc0_EmbeddingContext* c) {
  if (!c) return;
  const DeviceScope dev_(c->device);
  (void)hipDeviceSynchronize();  // setter work may sit on waveform contexts' streams
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->d_block) (void)hipFree(c->d_block);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  for (hipEvent_t e : c->vec_sent) if (e) (void)hipEventDestroy(e);
  if (c->kv_uploaded) (void)hipEventDestroy(c->kv_uploaded);
  if (c->kv_projected) (void)hipEventDestroy(c->kv_projected);
  delete c;
}
// kind 0 = additive, 1 = formant.  The work goes to the waveform context's stream when there is one, so it is ordered
// before that context's next GenerateWaveform1 without a wait here.
static void set_vector(const Beatrice20rc0_EmbeddingSetter* m, int kind, const float* w, const float* b, const float* embedding,
                       Beatrice20rc0_EmbeddingContext* ec, float* d_ctx_vec, Beatrice20rc0_WaveformContext1* wc, float* d_wave_row) {
  if (!m || !m->loaded || !ec || !ec->ok || !embedding || m->device != ec->device || (wc && wc->ok && wc->device != ec->device)) return;
  const DeviceScope dev_(ec->device);
  const int slot = kind * 2 + (ec->vec_next[kind] ^= 1);
  if (ec->vec_busy[slot]) { (void)hipEventSynchronize(ec->vec_sent[slot]); ec->vec_busy[slot] = false; }  // two calls ago
  float* h = ec->h_stage + (size_t)B_KV_LEN * B_KV_CH + (size_t)slot * B_HID;
  float* d = ec->d_tmp + (size_t)slot * B_HID;
  std::memcpy(h, embedding, sizeof(float) * B_HID);
  hipStream_t st = (wc && wc->ok) ? wc->stream : ec->stream;
  if (!hip_ok(hipMemcpyAsync(d, h, sizeof(float) * B_HID, hipMemcpyHostToDevice, st), "emb up")) return;
  embed_project_rows(w, b, d, d_ctx_vec, 1, st);
  if (d_wave_row) (void)hip_ok(hipMemcpyAsync(d_wave_row, d_ctx_vec, sizeof(float) * B_HID, hipMemcpyDeviceToDevice, st), "emb d2d");
  // The slot (pinned staging AND the device slot `d`) is free again only when the projection that reads `d` is done: the
  // event is recorded behind it, and the call that re-uses the slot waits for it above -- whatever stream that call is on
  // (one embedding context may serve several waveform contexts, or none).
  ec->vec_busy[slot] = hip_ok(hipEventRecord(ec->vec_sent[slot], st), "emb ev");
}
// ref beatrice.h:323-327; callers processor_core_2.cc:137-141, 451-455
void Beatrice20rc0_SetAdditiveSpeakerEmbedding(const Beatrice20rc0_EmbeddingSetter* m, const float* embedding,
                                               Beatrice20rc0_EmbeddingContext* ec, Beatrice20rc0_WaveformContext1* wc) {
  if (!m || !m->loaded) return;
  set_vector(m, 0, m->w.add_w, m->w.add_b, embedding, ec, ec ? ec->d_add : nullptr, wc, (wc && wc->ok) ? wc->st.d_add_tab : nullptr);
}
// ref beatrice.h:328-332; caller processor_core_2.cc:475-479
void Beatrice20rc0_SetFormantShiftEmbedding(const Beatrice20rc0_EmbeddingSetter* m, const float* embedding,
                                            Beatrice20rc0_EmbeddingContext* ec, Beatrice20rc0_WaveformContext1* wc) {
  if (!m || !m->loaded) return;
  set_vector(m, 1, m->w.frm_w, m->w.frm_b, embedding, ec, ec ? ec->d_frm : nullptr, wc, (wc && wc->ok) ? wc->st.d_frm_tab : nullptr);
}
// ref beatrice.h:333-338; callers processor_core_2.cc:165-170, 456-462.  Copies before returning: the host rewrites
// its morph slot in place right after registering (processor_core_2.cc:158-170).  The copy is host-side (pinned
// staging); the upload is enqueued behind the projections that still read the previous registration.
void Beatrice20rc0_RegisterKeyValueSpeakerEmbedding(const Beatrice20rc0_EmbeddingSetter* m, const float* kv,
                                                    Beatrice20rc0_EmbeddingContext* ec) {
  (void)m;
  if (!ec || !ec->ok || !kv) return;
  const DeviceScope dev_(ec->device);
  if (ec->kv_busy) { (void)hipEventSynchronize(ec->kv_uploaded); ec->kv_busy = false; }  // previous upload still reads the staging
  std::memcpy(ec->h_stage, kv, sizeof(float) * B_KV_LEN * B_KV_CH);
  if (ec->kv_proj_pending) (void)hip_ok(hipStreamWaitEvent(ec->stream, ec->kv_projected, 0), "kv order");
  if (!hip_ok(hipMemcpyAsync(ec->d_kv_raw, ec->h_stage, sizeof(float) * B_KV_LEN * B_KV_CH, hipMemcpyHostToDevice, ec->stream), "kv up")) return;
  ec->kv_busy = hip_ok(hipEventRecord(ec->kv_uploaded, ec->stream), "kv ev");
}
// ref beatrice.h:339-343; caller processor_core_2.h:161-169 (one block per hop after a change).  Projection on the
// waveform context's stream, behind the upload: done before that context's next GenerateWaveform1, no wait here.
void Beatrice20rc0_SetKeyValueSpeakerEmbedding(const Beatrice20rc0_EmbeddingSetter* m, int block,
                                               Beatrice20rc0_EmbeddingContext* ec, Beatrice20rc0_WaveformContext1* wc) {
  if (!m || !m->loaded || !ec || !ec->ok || !wc || !wc->ok || block < 0 || block >= B_NBLOCKS) return;
  if (m->device != ec->device || wc->device != ec->device) return;
  const DeviceScope dev_(ec->device);
  if (ec->kv_busy) (void)hip_ok(hipStreamWaitEvent(wc->stream, ec->kv_uploaded, 0), "kv wait");
  // kv_projected is ONE event: when the previous projection ran on another waveform context's stream, chain behind it
  // first, so that the event recorded below covers every projection that still reads the registration
  if (ec->kv_proj_pending && ec->kv_proj_stream != wc->stream) (void)hip_ok(hipStreamWaitEvent(wc->stream, ec->kv_projected, 0), "kv chain");
  embed_project_kv(m->w, block, ec->d_kv_raw, 1, wc->st.d_kt[block], wc->st.d_v[block], wc->stream);
  ec->kv_proj_pending = hip_ok(hipEventRecord(ec->kv_projected, wc->stream), "kv proj ev");
  ec->kv_proj_stream = wc->stream;
}

// ================================ speaker file =================================================
static Beatrice_ErrorCode open_speakers(const char* path, std::vector<float>* host, int* n) {
  const Beatrice_ErrorCode e = read_model_file(path, KIND_SPEAKERS, -1, host);
  if (e) return e;
  const long per = (long)B_CODEBOOK * B_PHONE_CH + B_HID + (long)B_KV_LEN * B_KV_CH;
  const long body = (long)host->size() - 9L * B_HID;
  if (body < per) return Beatrice_kFileTooSmall;
  if (body % per != 0) return Beatrice_kInvalidFileSize;
  *n = (int)(body / per);
  return Beatrice_kSuccess;
}
// ref beatrice.h:273-275; caller processor_core_2.cc:328-333
Beatrice_ErrorCode Beatrice20rc0_ReadNSpeakers(const char* path, int* output) {
  std::vector<float> host;
  int n = 0;
  const Beatrice_ErrorCode e = open_speakers(path, &host, &n);
  if (e) return e;
  *output = n;
  return Beatrice_kSuccess;
}
// ref beatrice.h:276-290; caller processor_core_2.cc:344-351
Beatrice_ErrorCode Beatrice20rc0_ReadSpeakerEmbeddings(const char* path, float* codebook, float* additive, float* formant,
                                                       float* kv) {
  std::vector<float> host;
  int n = 0;
  const Beatrice_ErrorCode e = open_speakers(path, &host, &n);
  if (e) return e;
  const float* p = host.data();
  std::memcpy(formant, p, sizeof(float) * 9 * B_HID);
  p += 9 * B_HID;
  for (int s = 0; s < n; ++s) {
    std::memcpy(codebook + (size_t)s * B_CODEBOOK * B_PHONE_CH, p, sizeof(float) * B_CODEBOOK * B_PHONE_CH);
    p += B_CODEBOOK * B_PHONE_CH;
    std::memcpy(additive + (size_t)s * B_HID, p, sizeof(float) * B_HID);
    p += B_HID;
    std::memcpy(kv + (size_t)s * B_KV_LEN * B_KV_CH, p, sizeof(float) * B_KV_LEN * B_KV_CH);
    p += B_KV_LEN * B_KV_CH;
  }
  return Beatrice_kSuccess;
}

}  // extern "C"
