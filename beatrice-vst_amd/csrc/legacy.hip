// legacy.hip -- the 2.0.0-alpha.2 (Beatrice20a2_*) and 2.0.0-beta.1 (Beatrice20b1_*) generations of the C-ABI
// (reference lib/beatricelib/beatrice.h:39-203; callers: reference src/common/processor_core_0.cc, processor_core_1.cc).
//
// Their networks are MODEL_SPEC section 6: the rc.0 modules with a 256-wide phone vector and no codebook step, 384 pitch
// classes, and a waveform generator whose conditioning is ONE additive vector the host hands over with every hop (speaker
// embedding + formant-shift embedding, processor_core_1.cc:121-142) -- no embedding setter, no key/value attention.  They run
// on the SAME kernels as rc.0 (the modules of phone.hip / pitch.hip / wave.hip take the widths at run time), through the
// same one-graph-per-call scheme as abi.hip.  Both generations share one implementation; the entry points are stamped per
// prefix.  Conventions of the boundary as in abi.hip: only the readers fail, per-hop calls emit zeros on any failure.
#include <cstring>
#include <vector>

#include "abi_objects.h"

using namespace bhip;

namespace bhip {
// (defined in abi.hip)
bool wait_stream(hipStream_t s);
bool run_hop_graph(HopGraph& g, const void* blob, int variant, hipStream_t s, void (*enqueue)(void*), void* ctx);
}  // namespace bhip

namespace {

enum : uint32_t { KIND_L_PHONE = 11, KIND_L_PITCH = 12, KIND_L_WAVE = 13, KIND_L_ROWS = 15 };
constexpr int kLPhoneCh = BEATRICE_20B1_PHONE_CHANNELS, kLBins = BEATRICE_20B1_PITCH_BINS;

// model objects: device blob + weight pointers, immutable after Read*Parameters
struct LPhoneModel { int device = target_device(); DeviceBlob blob; PhoneWeightsLegacy w{}; bool loaded = false; };
struct LPitchModel { int device = target_device(); DeviceBlob blob; PitchWeightsLegacy w{}; bool loaded = false; };
struct LWaveModel { int device = target_device(); DeviceBlob blob; WaveWeightsLegacy w{}; bool loaded = false; };

template <class Obj, class W>
Beatrice_ErrorCode read_into(Obj* m, const char* path, uint32_t kind) {
  if (!m) return Beatrice_kFileOpenError;
  std::vector<float> host;
  const Beatrice_ErrorCode e = read_model_file(path, kind, (long)W::n_floats(), &host);
  if (e) return e;
  const DeviceScope dev_(m->device);
  m->loaded = false;
  W::pack_host(host.data());
  if (!m->blob.upload(host.data(), host.size())) return Beatrice_kFileOpenError;  // device failure
  m->w.bind(m->blob.d);
  m->loaded = true;
  return Beatrice_kSuccess;
}
template <class Obj>
void destroy_model(Obj* m) {   // (Obj = the generation's own type: the object is deleted as what it was created as)
  if (!m) return;
  const DeviceScope dev_(m->device);
  m->blob.release();
  delete m;
}

// per-stream contexts
struct LPhoneCtx {
  int device = target_device();
  PhoneState st;
  hipStream_t stream = nullptr;
  float* h_io = nullptr;   // pinned: 160 in | mailbox (step counter) | 256 out
  int hop_count = 0;
  HopGraph graph;
  bool ok = false;
};
struct LPitchCtx {
  int device = target_device();
  PitchState st;
  hipStream_t stream = nullptr;
  float* h_io = nullptr;   // pinned: 160 in | mailbox (step counter, bin range) | 4 feat | 1 bin
  int hop_count = 0, min_q = 1, max_q = kLBins - 1;
  void* own_sel[2] = {nullptr, nullptr};
  HopGraph graph;
  bool ok = false;
};
struct LWaveCtx {
  int device = target_device();
  WaveState st;
  hipStream_t stream = nullptr;
  float* d_inputs = nullptr;   // device: 256 phone | 4 feat | 1 bin | step counter | 256 speaker vector
  float* h_io = nullptr;       // pinned: the same | 240 out
  int hop_count = 0;
  HopGraph graph;
  bool ok = false;
};
constexpr size_t kWaveInFloats = kLPhoneCh + 4 + 1 + 1 + B_HID;

template <class T>
T* create_phone_ctx() {
  auto* c = new T();
  const DeviceScope dev_(c->device);
  c->ok = make_stream(&c->stream) && c->st.create(1, 1, nullptr, 1, false, kLPhoneCh) &&
          hip_ok(hipHostMalloc(reinterpret_cast<void**>(&c->h_io), sizeof(float) * (B_IN_HOP + kMailboxWords + kLPhoneCh), hipHostMallocDefault), "hipHostMalloc");
  if (c->ok) c->st.hop = c->st.hop_in = c->st.hop_mailbox;   // the step counter arrives with the input copy
  c->st.advance_hop = false;
  c->st.skip_vq = true;   // these generations have no codebook: phone.out writes the module output
  return c;
}
template <class T>
void destroy_phone_ctx(T* c) {
  if (!c) return;
  const DeviceScope dev_(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->graph.drop();
  c->st.destroy();
  if (c->h_io) (void)hipHostFree(c->h_io);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
template <class T>
T* create_pitch_ctx() {
  auto* c = new T();
  const DeviceScope dev_(c->device);
  c->ok = make_stream(&c->stream) && c->st.create(1, 1, nullptr, false, false, kLBins) &&
          hip_ok(hipHostMalloc(reinterpret_cast<void**>(&c->h_io), sizeof(float) * (B_IN_HOP + kMailboxWords + 8), hipHostMallocDefault), "hipHostMalloc");
  if (c->ok) {
    c->st.hop = c->st.hop_in = c->st.hop_mailbox;
    c->own_sel[0] = c->st.d_min_q; c->own_sel[1] = c->st.d_max_q;
    c->st.d_min_q = c->st.hop_mailbox + 1;
    c->st.d_max_q = c->st.hop_mailbox + 2;
  }
  c->st.advance_hop = false;
  return c;
}
template <class T>
void destroy_pitch_ctx(T* c) {
  if (!c) return;
  const DeviceScope dev_(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->graph.drop();
  if (c->own_sel[0]) { c->st.d_min_q = static_cast<int*>(c->own_sel[0]); c->st.d_max_q = static_cast<int*>(c->own_sel[1]); }
  c->st.destroy();
  if (c->h_io) (void)hipHostFree(c->h_io);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
template <class T>
T* create_wave_ctx() {
  auto* c = new T();
  const DeviceScope dev_(c->device);
  c->ok = make_stream(&c->stream) &&
          hip_ok(hipMalloc(reinterpret_cast<void**>(&c->d_inputs), sizeof(float) * kWaveInFloats), "inputs") &&
          hip_ok(hipMemset(c->d_inputs, 0, sizeof(float) * kWaveInFloats), "inputs0") &&
          c->st.create(1, 1, 1, 1, 1, c->d_inputs, reinterpret_cast<int*>(c->d_inputs + kLPhoneCh + 4), c->d_inputs + kLPhoneCh, 1, false, /*legacy=*/true) &&
          hip_ok(hipHostMalloc(reinterpret_cast<void**>(&c->h_io), sizeof(float) * (kWaveInFloats + B_OUT_HOP), hipHostMallocDefault), "hipHostMalloc");
  if (c->ok) {
    c->st.hop = reinterpret_cast<int*>(c->d_inputs + kLPhoneCh + 4 + 1);
    // the conditioning "table" is the vector that travels with the hop's inputs (row 0: add_idx is zero-filled)
    (void)hipFree(c->st.d_add_tab);
    c->st.d_add_tab = c->d_inputs + kLPhoneCh + 4 + 2;
  }
  c->st.advance_hop = false;
  return c;
}
template <class T>
void destroy_wave_ctx(T* c) {
  if (!c) return;
  const DeviceScope dev_(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->graph.drop();
  if (c->ok) c->st.d_add_tab = nullptr;   // (points into d_inputs: not the state's to free)
  c->st.destroy();
  if (c->d_inputs) (void)hipFree(c->d_inputs);
  if (c->h_io) (void)hipHostFree(c->h_io);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int clamp_bin(int q) { return q < 1 ? 1 : (q > kLBins - 1 ? kLBins - 1 : q); }

template <class F>
bool run_graph(HopGraph& g, const void* blob, hipStream_t s, F enqueue) {
  return run_hop_graph(g, blob, 0, s, [](void* p) { (*static_cast<F*>(p))(); }, &enqueue);
}

// ref beatrice.h:64-68 / 133-137; caller processor_core_1.cc:51-53
void extract_phone(const LPhoneModel* m, const float* input, float* output, LPhoneCtx* ctx) {
  std::memset(output, 0, sizeof(float) * kLPhoneCh);
  if (!m || !m->loaded || !ctx || !ctx->ok || m->device != ctx->device) return;
  const DeviceScope dev_(ctx->device);
  float* h_in = ctx->h_io;
  float* h_out = ctx->h_io + B_IN_HOP + kMailboxWords;
  std::memcpy(h_in, input, sizeof(float) * B_IN_HOP);
  reinterpret_cast<int*>(h_in + B_IN_HOP)[0] = ctx->hop_count;
  ctx->hop_count = hop_next(ctx->hop_count);
  bool ok = run_graph(ctx->graph, m->blob.d, ctx->stream, [&] {
    (void)hipMemcpyAsync(ctx->st.d_in, h_in, sizeof(float) * (B_IN_HOP + kMailboxWords), hipMemcpyHostToDevice, ctx->stream);
    phone_forward(m->w, ctx->st, ctx->stream);
    (void)hipMemcpyAsync(h_out, ctx->st.d_phone, sizeof(float) * kLPhoneCh, hipMemcpyDeviceToHost, ctx->stream);
  });
  ok = wait_stream(ctx->stream) && ok;
  if (team_timed_out(ctx->st)) { ok = false; team_recover(ctx->st, ctx->stream); ctx->graph.drop(); }   // (engine.h: zeros for this call, the per-layer launches from the next one on)
  if (ok) std::memcpy(output, h_out, sizeof(float) * kLPhoneCh);
}
// ref beatrice.h:88-93 / 157-162; caller processor_core_1.cc:54-57
void estimate_pitch(const LPitchModel* m, const float* input, int* out_q, float* out_feat, LPitchCtx* ctx) {
  *out_q = 1;
  std::memset(out_feat, 0, sizeof(float) * 4);
  if (!m || !m->loaded || !ctx || !ctx->ok || m->device != ctx->device) return;
  const DeviceScope dev_(ctx->device);
  float* h_in = ctx->h_io;
  float* h_feat = ctx->h_io + B_IN_HOP + kMailboxWords;
  int* h_q = reinterpret_cast<int*>(ctx->h_io + B_IN_HOP + kMailboxWords + 4);
  std::memcpy(h_in, input, sizeof(float) * B_IN_HOP);
  int* mb = reinterpret_cast<int*>(h_in + B_IN_HOP);
  mb[0] = ctx->hop_count; mb[1] = ctx->min_q; mb[2] = ctx->max_q;
  ctx->hop_count = hop_next(ctx->hop_count);
  bool ok = run_graph(ctx->graph, m->blob.d, ctx->stream, [&] {
    (void)hipMemcpyAsync(ctx->st.d_in, h_in, sizeof(float) * (B_IN_HOP + kMailboxWords), hipMemcpyHostToDevice, ctx->stream);
    pitch_forward(m->w, ctx->st, ctx->stream);
    if (ctx->st.q_raw_in_feat) {
      (void)hipMemcpyAsync(h_feat, ctx->st.d_feat, sizeof(float) * 5, hipMemcpyDeviceToHost, ctx->stream);
    } else {
      (void)hipMemcpyAsync(h_feat, ctx->st.d_feat, sizeof(float) * 4, hipMemcpyDeviceToHost, ctx->stream);
      (void)hipMemcpyAsync(h_q, ctx->st.d_q_raw, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    }
  });
  ok = wait_stream(ctx->stream) && ok;
  if (team_timed_out(ctx->st)) { ok = false; team_recover(ctx->st, ctx->stream); (void)hipMemsetAsync(ctx->st.d_prev_q, 0, sizeof(int), ctx->stream); ctx->graph.drop(); }
  if (ok) { *out_q = *h_q; std::memcpy(out_feat, h_feat, sizeof(float) * 4); }
}
// ref beatrice.h:112-120 / 181-189; caller processor_core_1.cc:139-142
void generate_waveform(const LWaveModel* m, const float* phone, const int* q, const float* feat, const float* speaker, float* output,
                       LWaveCtx* ctx) {
  std::memset(output, 0, sizeof(float) * B_OUT_HOP);
  if (!m || !m->loaded || !ctx || !ctx->ok || m->device != ctx->device) return;
  const DeviceScope dev_(ctx->device);
  float* h_in = ctx->h_io;
  float* h_out = ctx->h_io + kWaveInFloats;
  std::memcpy(h_in, phone, sizeof(float) * kLPhoneCh);
  std::memcpy(h_in + kLPhoneCh, feat, sizeof(float) * 4);
  std::memcpy(h_in + kLPhoneCh + 4, q, sizeof(int));
  std::memcpy(h_in + kLPhoneCh + 5, &ctx->hop_count, sizeof(int));
  std::memcpy(h_in + kLPhoneCh + 6, speaker, sizeof(float) * B_HID);
  ctx->hop_count = hop_next(ctx->hop_count);
  bool ok = run_graph(ctx->graph, m->blob.d, ctx->stream, [&] {
    (void)hipMemcpyAsync(ctx->d_inputs, h_in, sizeof(float) * kWaveInFloats, hipMemcpyHostToDevice, ctx->stream);
    wave_forward(m->w, ctx->st, ctx->stream);
    (void)hipMemcpyAsync(h_out, ctx->st.d_out, sizeof(float) * B_OUT_HOP, hipMemcpyDeviceToHost, ctx->stream);
  });
  ok = wait_stream(ctx->stream) && ok;
  if (team_timed_out(ctx->st)) { ok = false; team_recover(ctx->st, ctx->stream); ctx->graph.drop(); }
  if (ok) std::memcpy(output, h_out, sizeof(float) * B_OUT_HOP);
}

// embedding rows [n][256]: speaker_embeddings.bin and formant_shift_embeddings.bin go through the same reader
// (processor_core_1.cc:189-216)
Beatrice_ErrorCode open_rows(const char* path, std::vector<float>* host, int* rows) {
  const Beatrice_ErrorCode e = read_model_file(path, KIND_L_ROWS, -1, host);
  if (e) return e;
  if (host->size() < (size_t)B_HID) return Beatrice_kFileTooSmall;
  if (host->size() % B_HID != 0) return Beatrice_kInvalidFileSize;
  *rows = (int)(host->size() / B_HID);
  return Beatrice_kSuccess;
}
Beatrice_ErrorCode read_n_rows(const char* path, int* output) {
  std::vector<float> host;
  int rows = 0;
  const Beatrice_ErrorCode e = open_rows(path, &host, &rows);
  if (e) return e;
  *output = rows;
  return Beatrice_kSuccess;
}
Beatrice_ErrorCode read_rows(const char* path, float* output) {
  std::vector<float> host;
  int rows = 0;
  const Beatrice_ErrorCode e = open_rows(path, &host, &rows);
  if (e) return e;
  std::memcpy(output, host.data(), sizeof(float) * host.size());
  return Beatrice_kSuccess;
}

}  // namespace

// the opaque objects of one generation are the shared ones under its own type names
#define BEATRICE_LEGACY_GENERATION(G)                                                                                                  \
  struct G##_PhoneExtractor : LPhoneModel {}; struct G##_PhoneContext1 : LPhoneCtx {};                                                  \
  struct G##_PitchEstimator : LPitchModel {}; struct G##_PitchContext1 : LPitchCtx {};                                                  \
  struct G##_WaveformGenerator : LWaveModel {}; struct G##_WaveformContext1 : LWaveCtx {};                                              \
  extern "C" {                                                                                                                          \
  G##_PhoneExtractor* G##_CreatePhoneExtractor(void) { return new G##_PhoneExtractor(); }                                               \
  void G##_DestroyPhoneExtractor(G##_PhoneExtractor* m) { destroy_model(m); }                                \
  G##_PhoneContext1* G##_CreatePhoneContext1(void) { return create_phone_ctx<G##_PhoneContext1>(); }                      \
  void G##_DestroyPhoneContext1(G##_PhoneContext1* c) { destroy_phone_ctx(c); }                                                         \
  Beatrice_ErrorCode G##_ReadPhoneExtractorParameters(G##_PhoneExtractor* m, const char* p) { return read_into<G##_PhoneExtractor, PhoneWeightsLegacy>(m, p, KIND_L_PHONE); } \
  void G##_ExtractPhone1(const G##_PhoneExtractor* m, const float* in, float* out, G##_PhoneContext1* c) { extract_phone(m, in, out, c); } \
  G##_PitchEstimator* G##_CreatePitchEstimator(void) { return new G##_PitchEstimator(); }                                               \
  void G##_DestroyPitchEstimator(G##_PitchEstimator* m) { destroy_model(m); }                                \
  G##_PitchContext1* G##_CreatePitchContext1(void) { return create_pitch_ctx<G##_PitchContext1>(); }                      \
  void G##_DestroyPitchContext1(G##_PitchContext1* c) { destroy_pitch_ctx(c); }                                                         \
  Beatrice_ErrorCode G##_ReadPitchEstimatorParameters(G##_PitchEstimator* m, const char* p) { return read_into<G##_PitchEstimator, PitchWeightsLegacy>(m, p, KIND_L_PITCH); } \
  void G##_SetMinQuantizedPitch(G##_PitchContext1* c, int q) { if (c && c->ok) c->min_q = clamp_bin(q); }                               \
  void G##_SetMaxQuantizedPitch(G##_PitchContext1* c, int q) { if (c && c->ok) c->max_q = clamp_bin(q); }                               \
  void G##_EstimatePitch1(const G##_PitchEstimator* m, const float* in, int* q, float* f, G##_PitchContext1* c) { estimate_pitch(m, in, q, f, c); } \
  Beatrice_ErrorCode G##_ReadNSpeakers(const char* p, int* o) { return read_n_rows(p, o); }                                             \
  Beatrice_ErrorCode G##_ReadSpeakerEmbeddings(const char* p, float* o) { return read_rows(p, o); }                                     \
  G##_WaveformGenerator* G##_CreateWaveformGenerator(void) { return new G##_WaveformGenerator(); }                                      \
  void G##_DestroyWaveformGenerator(G##_WaveformGenerator* m) { destroy_model(m); }                           \
  G##_WaveformContext1* G##_CreateWaveformContext1(void) { return create_wave_ctx<G##_WaveformContext1>(); }              \
  void G##_DestroyWaveformContext1(G##_WaveformContext1* c) { destroy_wave_ctx(c); }                                                    \
  Beatrice_ErrorCode G##_ReadWaveformGeneratorParameters(G##_WaveformGenerator* m, const char* p) { return read_into<G##_WaveformGenerator, WaveWeightsLegacy>(m, p, KIND_L_WAVE); } \
  void G##_GenerateWaveform1(const G##_WaveformGenerator* m, const float* ph, const int* q, const float* f, const float* s, float* out, \
                             G##_WaveformContext1* c) { generate_waveform(m, ph, q, f, s, out, c); }                                    \
  }

BEATRICE_LEGACY_GENERATION(Beatrice20a2)
BEATRICE_LEGACY_GENERATION(Beatrice20b1)
