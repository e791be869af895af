// processor_core_legacy.cc -- ProcessorCoreLegacy (processor_core.h): the host core of the 2.0.0-alpha.2 and
// 2.0.0-beta.1 generations on the Beatrice20a2_* / Beatrice20b1_* entry points of include/beatrice_abi.h.
// Behavioural reference: src/common/processor_core_1.cc (processor_core_0.cc is the same text on the 20a2 prefix).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "processor_core.h"

namespace beatrice_amd {

// one generation's entry points behind untyped handles, so that one class serves both prefixes
struct LegacyLib {
  void* (*create[6])();  // PhoneExtractor, PitchEstimator, WaveformGenerator, PhoneContext1, PitchContext1, WaveformContext1
  void (*destroy[6])(void*);
  Beatrice_ErrorCode (*read_model[3])(void*, const char*);
  Beatrice_ErrorCode (*read_n_speakers)(const char*, int*);
  Beatrice_ErrorCode (*read_rows)(const char*, float*);
  void (*extract_phone)(const void*, const float*, float*, void*);
  void (*estimate_pitch)(const void*, const float*, int*, float*, void*);
  void (*set_min)(void*, int);
  void (*set_max)(void*, int);
  void (*generate)(const void*, const float*, const int*, const float*, const float*, float*, void*);
};

namespace {
#define BEATRICE_LEGACY_LIB(G)                                                                                                  \
  const LegacyLib k##G = {                                                                                                      \
      {[]() -> void* { return G##_CreatePhoneExtractor(); }, []() -> void* { return G##_CreatePitchEstimator(); },              \
       []() -> void* { return G##_CreateWaveformGenerator(); }, []() -> void* { return G##_CreatePhoneContext1(); },            \
       []() -> void* { return G##_CreatePitchContext1(); }, []() -> void* { return G##_CreateWaveformContext1(); }},            \
      {[](void* p) { G##_DestroyPhoneExtractor(static_cast<G##_PhoneExtractor*>(p)); },                                         \
       [](void* p) { G##_DestroyPitchEstimator(static_cast<G##_PitchEstimator*>(p)); },                                         \
       [](void* p) { G##_DestroyWaveformGenerator(static_cast<G##_WaveformGenerator*>(p)); },                                   \
       [](void* p) { G##_DestroyPhoneContext1(static_cast<G##_PhoneContext1*>(p)); },                                           \
       [](void* p) { G##_DestroyPitchContext1(static_cast<G##_PitchContext1*>(p)); },                                           \
       [](void* p) { G##_DestroyWaveformContext1(static_cast<G##_WaveformContext1*>(p)); }},                                    \
      {[](void* m, const char* f) { return G##_ReadPhoneExtractorParameters(static_cast<G##_PhoneExtractor*>(m), f); },         \
       [](void* m, const char* f) { return G##_ReadPitchEstimatorParameters(static_cast<G##_PitchEstimator*>(m), f); },         \
       [](void* m, const char* f) { return G##_ReadWaveformGeneratorParameters(static_cast<G##_WaveformGenerator*>(m), f); }},  \
      [](const char* f, int* n) { return G##_ReadNSpeakers(f, n); },                                                            \
      [](const char* f, float* o) { return G##_ReadSpeakerEmbeddings(f, o); },                                                  \
      [](const void* m, const float* in, float* out, void* c) {                                                                 \
        G##_ExtractPhone1(static_cast<const G##_PhoneExtractor*>(m), in, out, static_cast<G##_PhoneContext1*>(c));              \
      },                                                                                                                        \
      [](const void* m, const float* in, int* q, float* f, void* c) {                                                           \
        G##_EstimatePitch1(static_cast<const G##_PitchEstimator*>(m), in, q, f, static_cast<G##_PitchContext1*>(c));            \
      },                                                                                                                        \
      [](void* c, int v) { G##_SetMinQuantizedPitch(static_cast<G##_PitchContext1*>(c), v); },                                  \
      [](void* c, int v) { G##_SetMaxQuantizedPitch(static_cast<G##_PitchContext1*>(c), v); },                                  \
      [](const void* m, const float* ph, const int* q, const float* f, const float* spk, float* out, void* c) {                 \
        G##_GenerateWaveform1(static_cast<const G##_WaveformGenerator*>(m), ph, q, f, spk, out,                                 \
                              static_cast<G##_WaveformContext1*>(c));                                                           \
      }};
BEATRICE_LEGACY_LIB(Beatrice20a2)
BEATRICE_LEGACY_LIB(Beatrice20b1)
#undef BEATRICE_LEGACY_LIB
static_assert(BEATRICE_20A2_PITCH_BINS == BEATRICE_20B1_PITCH_BINS && BEATRICE_20A2_PHONE_CHANNELS == BEATRICE_20B1_PHONE_CHANNELS);
constexpr int kHidden = BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS;
enum { kPhoneExtractor, kPitchEstimator, kWaveformGenerator, kPhoneContext, kPitchContext, kWaveformContext };
}  // namespace

ProcessorCoreLegacy::ProcessorCoreLegacy(double sample_rate, int version)
    : StreamingCore(sample_rate, BEATRICE_20B1_PITCH_BINS), version_(version), lib_(version == 0 ? kBeatrice20a2 : kBeatrice20b1),
      phone_extractor_(lib_.create[kPhoneExtractor]()), pitch_estimator_(lib_.create[kPitchEstimator]()),
      waveform_generator_(lib_.create[kWaveformGenerator]()), phone_context_(lib_.create[kPhoneContext]()),
      pitch_context_(lib_.create[kPitchContext]()), waveform_context_(lib_.create[kWaveformContext]()) {}

ProcessorCoreLegacy::~ProcessorCoreLegacy() {
  lib_.destroy[kPhoneExtractor](phone_extractor_);
  lib_.destroy[kPitchEstimator](pitch_estimator_);
  lib_.destroy[kWaveformGenerator](waveform_generator_);
  lib_.destroy[kPhoneContext](phone_context_);
  lib_.destroy[kPitchContext](pitch_context_);
  lib_.destroy[kWaveformContext](waveform_context_);
}

// reference processor_core_1.cc:35-40: the speaker id is checked when audio arrives, not when it is set
ErrorCode ProcessorCoreLegacy::Preflight() const {
  return target_speaker_ < 0 || target_speaker_ > n_speakers_ ? ErrorCode::kSpeakerIDOutOfRange : ErrorCode::kSuccess;
}

// one model hop (reference processor_core_1.cc:52-143)
void ProcessorCoreLegacy::Hop(const float* in160, float* out240) {
  alignas(64) float phone[BEATRICE_20B1_PHONE_CHANNELS];
  lib_.extract_phone(phone_extractor_, in160, phone, phone_context_);
  int q = 0;
  float feature[4];
  lib_.estimate_pitch(pitch_estimator_, in160, &q, feature, pitch_context_);
  q = TransformPitch(q);
  RecordPitch(q);
  const size_t target = static_cast<size_t>(target_speaker_);
  // morphing: one solver update per hop; the slot follows it until it has converged (:121-130)
  if (target_speaker_ == n_speakers_ && !mean_.Update()) mean_.Result(speaker_embeddings_.data() + target * kHidden);
  alignas(64) float speaker[kHidden];
  const float* row = speaker_embeddings_.data() + target * kHidden;
  const float* shift = formant_shift_embeddings_.data() + static_cast<size_t>(std::round(formant_shift_ * 2 + 4)) * kHidden;
  for (int i = 0; i < kHidden; ++i) speaker[i] = row[i] + shift[i];
  lib_.generate(waveform_generator_, phone, &q, feature, speaker, out240, waveform_context_);
}

ErrorCode ProcessorCoreLegacy::ResetContext() {  // reference processor_core_1.cc:145-163
  lib_.destroy[kPhoneContext](phone_context_);
  lib_.destroy[kPitchContext](pitch_context_);
  lib_.destroy[kWaveformContext](waveform_context_);
  phone_context_ = lib_.create[kPhoneContext]();
  pitch_context_ = lib_.create[kPitchContext]();
  waveform_context_ = lib_.create[kWaveformContext]();
  ErrorCode error = SetMinSourcePitch(min_source_pitch_);
  if (const ErrorCode e = SetMaxSourcePitch(max_source_pitch_); error == ErrorCode::kSuccess) error = e;
  return error;
}

ErrorCode ProcessorCoreLegacy::LoadModel(const std::filesystem::path& model_file) {  // reference processor_core_1.cc:165-221
  model_file_.clear();
  const auto dir = model_file.parent_path();
  auto path = [&](const char* name) { return (dir / name).u8string(); };
  auto cstr = [](const auto& s) { return reinterpret_cast<const char*>(s.c_str()); };
#define BEATRICE_TRY_READ(call) \
  if (const auto err = (call)) return static_cast<ErrorCode>(err);
  BEATRICE_TRY_READ(lib_.read_model[kPhoneExtractor](phone_extractor_, cstr(path("phone_extractor.bin"))))
  BEATRICE_TRY_READ(lib_.read_model[kPitchEstimator](pitch_estimator_, cstr(path("pitch_estimator.bin"))))
  BEATRICE_TRY_READ(lib_.read_model[kWaveformGenerator](waveform_generator_, cstr(path("waveform_generator.bin"))))
  const auto spk = path("speaker_embeddings.bin");
  BEATRICE_TRY_READ(lib_.read_n_speakers(cstr(spk), &n_speakers_))
  if (n_speakers_ < 1) return ErrorCode::kInvalidFileSize;   // (a table without rows: nothing to convert to, and no row to morph from)
  speaker_embeddings_.resize((static_cast<size_t>(n_speakers_) + 1) * kHidden, 0.0f);  // + morph slot
  BEATRICE_TRY_READ(lib_.read_rows(cstr(spk), speaker_embeddings_.data()))
  mean_.Initialize(n_speakers_, kHidden, speaker_embeddings_.data());
  formant_shift_embeddings_.resize(9 * kHidden);
  BEATRICE_TRY_READ(lib_.read_rows(cstr(path("formant_shift_embeddings.bin")), formant_shift_embeddings_.data()))
#undef BEATRICE_TRY_READ
  model_file_ = model_file;
  return ApplySpeakerMorphingWeights();
}

ErrorCode ProcessorCoreLegacy::SetTargetSpeaker(int id) {  // reference processor_core_1.cc:233-240
  if (id < 0) return ErrorCode::kSpeakerIDOutOfRange;
  target_speaker_ = id;
  return ErrorCode::kSuccess;
}
ErrorCode ProcessorCoreLegacy::SetFormantShift(double v) { formant_shift_ = std::clamp(v, -2.0, 2.0); return ErrorCode::kSuccess; }
ErrorCode ProcessorCoreLegacy::SetMinSourcePitch(double v) {  // reference processor_core_1.cc:309-319
  min_source_pitch_ = std::clamp(v, 0.0, 128.0);
  lib_.set_min(pitch_context_, NoteToBin(min_source_pitch_));
  return ErrorCode::kSuccess;
}
ErrorCode ProcessorCoreLegacy::SetMaxSourcePitch(double v) {
  max_source_pitch_ = std::clamp(v, 0.0, 128.0);
  lib_.set_max(pitch_context_, NoteToBin(max_source_pitch_));
  return ErrorCode::kSuccess;
}

ErrorCode ProcessorCoreLegacy::SetSpeakerMorphingWeights(const std::array<float, kMaxNSpeakers>& weights) {
  if (weights == morph_weights_) return ErrorCode::kSuccess;  // reference processor_core_1.cc:260-267
  morph_weights_ = weights;
  return ApplySpeakerMorphingWeights();
}

// reference processor_core_1.cc:269-281: every speaker with a prepared weight > 0 takes part (no "8 largest" here); the
// slot receives the solver's starting point at once and is refined hop by hop
ErrorCode ProcessorCoreLegacy::ApplySpeakerMorphingWeights() {
  if (!IsLoaded()) return ErrorCode::kSuccess;
  const auto w = PrepareVoiceMorphWeights(morph_weights_, n_speakers_);
  mean_.SetWeights(n_speakers_, w.data());
  mean_.Result(speaker_embeddings_.data() + static_cast<size_t>(n_speakers_) * kHidden);
  return ErrorCode::kSuccess;
}

}  // namespace beatrice_amd
