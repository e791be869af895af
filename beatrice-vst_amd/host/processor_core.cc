// processor_core.cc -- see processor_core.h.  Behavioural references are cited per function; the
// arithmetic order of the DSP stages is kept identical to the reference so that outputs match it
// bit for bit (tests/test_host_layer.py compares against vectors minted from the reference headers).
#include "processor_core.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace beatrice_amd {

namespace {
constexpr int kTapsPerOutput = 32;  // filter length in low-rate samples (reference resample.h:415 passes 32)
constexpr int kBlock = 480;         // 10 ms at 48 kHz
constexpr double kPi = 3.14159265358979323846;
constexpr double kBinsPerSemitone = BEATRICE_PITCH_BINS_PER_OCTAVE / 12.0;

double DbToAmplitude(double db) { return std::pow(10.0, db * 0.05); }

// smallest-denominator search on the Stern-Brocot tree, parts < 1000 (reference resample.h:25-46)
void SimpleFraction(double ratio, int* numer, int* denom) {
  int a = 0, b = 1, c = 1, d = 0;
  for (;;) {
    const int mn = a + c, md = b + d;
    const bool big = mn >= 1000 || md >= 1000;
    if (ratio * md < mn) {
      if (big) { *numer = a; *denom = b; return; }
      c = mn; d = md;
    } else {
      if (big) { *numer = c; *denom = d; return; }
      a = mn; b = md;
    }
  }
}
}  // namespace

// ---- gain (reference gain.h:41-71) -------------------------------------------------------------
void GainRamp::Apply(const float* in, float* out, int n) {
  const double goal = DbToAmplitude(target_db_);
  double amp = DbToAmplitude(now_db_);
  const double per_sample_db = 2.0 / (rate_ * 0.001);
  int i = 0;
  if (amp < goal) {
    const double up = DbToAmplitude(per_sample_db);
    while (i < n && amp < goal) { amp = std::min(amp * up, goal); out[i] = static_cast<float>(in[i] * amp); ++i; }
  } else if (amp > goal) {
    const double down = DbToAmplitude(-per_sample_db);
    while (i < n && amp > goal) { amp = std::max(amp * down, goal); out[i] = static_cast<float>(in[i] * amp); ++i; }
  }
  for (; i < n; ++i) out[i] = static_cast<float>(in[i] * amp);
  now_db_ = 20.0 * std::log10(amp);
}

// ---- resampler pair (reference resample.h:130-270) ---------------------------------------------
void RateBridge::Configure(double outer_rate, double inner_rate, double cutoff_in, double cutoff_out) {
  ready_ = false;
  if (outer_rate <= 0.0 || inner_rate <= 0.0) return;
  high_is_outer_ = outer_rate >= inner_rate;
  const double high = high_is_outer_ ? outer_rate : inner_rate, low = high_is_outer_ ? inner_rate : outer_rate;
  const double cut_down = high_is_outer_ ? cutoff_in : cutoff_out, cut_up = high_is_outer_ ? cutoff_out : cutoff_in;
  SimpleFraction(high / low, &hi_, &lo_);
  if (hi_ == 0 || lo_ == 0) return;
  const int n = kTapsPerOutput * hi_ + 1, mid = n / 2;
  taps_down_.resize(n);
  taps_up_.resize(n);
  auto sinc = [](double x) { return std::abs(x) < 1e-8 ? 1.0 : std::sin(x * kPi) / (x * kPi); };
  for (int i = 0; i < n; ++i) {
    const double x = static_cast<double>(i - mid) / static_cast<double>(hi_);
    const double hann = 0.5 - 0.5 * std::cos(kPi * 2.0 / static_cast<double>(n - 1) * static_cast<double>(i));
    taps_down_[i] = static_cast<float>(cut_down * sinc(x * cut_down) * hann);
    taps_up_[i] = static_cast<float>(cut_up * sinc(x * cut_up) * hann);
  }
  phase_down_ = phase_up_ = hi_ - 1;
  hist_high_.Reset(kTapsPerOutput * hi_ / lo_ + 1);
  hist_low_.Reset(kTapsPerOutput + 1);
  ready_ = true;
}

void RateBridge::Decimate(const std::vector<float>& in, std::vector<float>& out) {
  const float scale = static_cast<float>(lo_) / static_cast<float>(hi_);
  out.clear();
  const int last = static_cast<int>(taps_down_.size()) - 1;
  for (const float x : in) {
    hist_high_.Push(x);
    phase_down_ += lo_;
    if (phase_down_ < hi_) continue;
    phase_down_ -= hi_;
    float acc = 0.0f;
    int back = 1;
    for (int tap = lo_ - phase_down_; tap < last; tap += lo_) acc += hist_high_.Back(back++) * taps_down_[tap];
    out.push_back(acc * scale);
  }
}

void RateBridge::Interpolate(const std::vector<float>& in, std::vector<float>& out) {
  const int n_in = static_cast<int>(in.size());
  const int n_out = high_is_outer_ ? (n_in * hi_ + phase_down_ - phase_up_) / lo_ : ((n_in + 1) * hi_ - phase_up_ - 1) / lo_;
  out.resize(n_out);
  const int last = static_cast<int>(taps_up_.size()) - 1;
  int next = 0;
  for (int o = 0; o < n_out; ++o) {
    phase_up_ += lo_;
    if (phase_up_ >= hi_) { phase_up_ -= hi_; hist_low_.Push(in[next++]); }
    float acc = 0.0f;
    int back = 1;
    for (int tap = phase_up_; tap < last; tap += hi_) acc += hist_low_.Back(back++) * taps_up_[tap];
    out[o] = acc;
  }
}

void RateBridge::ToInner(const std::vector<float>& in, std::vector<float>& out) {
  if (!ready_) { out.clear(); return; }
  if (high_is_outer_) Decimate(in, out); else Interpolate(in, out);
}
void RateBridge::ToOuter(const std::vector<float>& in, std::vector<float>& out) {
  if (!ready_) { out.clear(); return; }
  if (high_is_outer_) Interpolate(in, out); else Decimate(in, out);
}

// ---- processor ----------------------------------------------------------------------------------
static void ConfigureBridge(RateBridge& b, double sr) {
  // cutoffs: reference resample.h:412-417
  b.Configure(sr, 48000.0, 0.99 * 16000.0 / std::clamp(sr, 16000.0, 48000.0), 0.99 * 24000.0 / std::clamp(sr, 24000.0, 48000.0));
}

StreamingCore::StreamingCore(double sample_rate, int pitch_bins)
    : sample_rate_(sample_rate), pitch_bins_(pitch_bins), gain_in_(sample_rate), gain_out_(sample_rate), fifo_(kBlock, 0.0f) {
  ConfigureBridge(bridge_, sample_rate);
  ReserveBlocks(kDefaultMaxBlock);
}

unsigned long long StreamingCore::BufferFingerprint() const {
  unsigned long long h = 1469598103934665603ull;
  for (const std::vector<float>* v : {&io_, &work_, &scratch_}) {
    h = (h ^ reinterpret_cast<unsigned long long>(v->data())) * 1099511628211ull;
    h = (h ^ static_cast<unsigned long long>(v->capacity())) * 1099511628211ull;
  }
  // (every other container the per-hop path writes: the pitch-trace ring)
  h = (h ^ reinterpret_cast<unsigned long long>(pitch_trace_.data())) * 1099511628211ull;
  h = (h ^ static_cast<unsigned long long>(pitch_trace_.capacity())) * 1099511628211ull;
  return h;
}
void StreamingCore::EnablePitchTrace(int capacity) {
  pitch_trace_.assign(static_cast<size_t>(capacity > 0 ? capacity : 0), 0);
  pitch_trace_.shrink_to_fit();
  trace_count_ = 0;
}
std::vector<int> StreamingCore::TakePitchTrace() {
  std::vector<int> t;
  const unsigned long long cap = pitch_trace_.size();
  if (cap == 0) return t;
  const unsigned long long n = trace_count_ < cap ? trace_count_ : cap;
  t.reserve(static_cast<size_t>(n));
  for (unsigned long long i = trace_count_ - n; i < trace_count_; ++i) t.push_back(pitch_trace_[static_cast<size_t>(i % cap)]);
  trace_count_ = 0;
  return t;
}
void StreamingCore::ReserveBlocks(int max_block) {
  if (max_block < 1) return;
  reserved_block_ = std::max(reserved_block_, max_block);
  // inner (48 kHz) samples a host block can yield: n * 48000 / rate, + the clocks' slack
  const double ratio = sample_rate_ > 0.0 ? 48000.0 / sample_rate_ : 1.0;
  const size_t inner = static_cast<size_t>(std::ceil(reserved_block_ * std::max(1.0, ratio))) + 8;
  io_.reserve(inner);
  work_.reserve(inner);
  scratch_.reserve(inner);
}

ProcessorCore2::ProcessorCore2(double sample_rate)
    : StreamingCore(sample_rate, BEATRICE_20RC0_PITCH_BINS),
      phone_extractor_(Beatrice20rc0_CreatePhoneExtractor()), pitch_estimator_(Beatrice20rc0_CreatePitchEstimator()),
      waveform_generator_(Beatrice20rc0_CreateWaveformGenerator()), embedding_setter_(Beatrice20rc0_CreateEmbeddingSetter()),
      phone_context_(Beatrice20rc0_CreatePhoneContext1()), pitch_context_(Beatrice20rc0_CreatePitchContext1()),
      waveform_context_(Beatrice20rc0_CreateWaveformContext1()), embedding_context_(Beatrice20rc0_CreateEmbeddingContext()) {}

ProcessorCore2::~ProcessorCore2() {
  Beatrice20rc0_DestroyPhoneExtractor(phone_extractor_);
  Beatrice20rc0_DestroyPitchEstimator(pitch_estimator_);
  Beatrice20rc0_DestroyWaveformGenerator(waveform_generator_);
  Beatrice20rc0_DestroyEmbeddingSetter(embedding_setter_);
  Beatrice20rc0_DestroyPhoneContext1(phone_context_);
  Beatrice20rc0_DestroyPitchContext1(pitch_context_);
  Beatrice20rc0_DestroyWaveformContext1(waveform_context_);
  Beatrice20rc0_DestroyEmbeddingContext(embedding_context_);
}

// guards and chain: reference processor_core_2.cc:24-48
ErrorCode StreamingCore::Process(const float* input, float* output, int n_samples) {
  auto silence = [&](ErrorCode e) { std::memset(output, 0, sizeof(float) * n_samples); return e; };
  if (!IsLoaded()) return silence(ErrorCode::kModelNotLoaded);
  if (!bridge_.IsReady()) return silence(ErrorCode::kResamplerNotReady);
  if (!gain_in_.IsReady() || !gain_out_.IsReady()) return silence(ErrorCode::kGainNotReady);
  if (const ErrorCode e = Preflight(); e != ErrorCode::kSuccess) return silence(e);
  if (pitch_correction_type_ < 0 || pitch_correction_type_ > 1) return silence(ErrorCode::kInvalidPitchCorrectionType);
  io_.resize(n_samples);
  gain_in_.Apply(input, io_.data(), n_samples);
  bridge_.ToInner(io_, work_);                               // host rate -> 48 kHz
  scratch_.resize(work_.size());
  Reblock(work_.data(), scratch_.data(), static_cast<int>(work_.size()));
  io_.assign(scratch_.begin(), scratch_.end());
  bridge_.ToOuter(io_, work_);                               // 48 kHz -> host rate
  std::memcpy(output, work_.data(), sizeof(float) * n_samples);
  gain_out_.Apply(output, output, n_samples);
  return ErrorCode::kSuccess;
}

// exact-480 FIFO, emits the previous block's result (reference resample.h:343-363)
void StreamingCore::Reblock(const float* in, float* out, int n) {
  int done = 0;
  while (done < n) {
    const int take = std::min(kBlock - fifo_fill_, n - done);
    std::memcpy(out + done, fifo_.data() + fifo_fill_, sizeof(float) * take);
    std::memcpy(fifo_.data() + fifo_fill_, in + done, sizeof(float) * take);
    fifo_fill_ += take;
    done += take;
    if (fifo_fill_ == kBlock) {
      fifo_fill_ = 0;
      float processed[kBlock];
      Block480(fifo_.data(), processed);
      std::memcpy(fifo_.data(), processed, sizeof(processed));
    }
  }
}

// 480 @48 kHz -> every third sample -> model hop -> zero-stuffed 480 (reference resample.h:380-394)
void StreamingCore::Block480(const float* in480, float* out480) {
  alignas(64) float in160[BEATRICE_IN_HOP_LENGTH];
  alignas(64) float out240[BEATRICE_OUT_HOP_LENGTH];
  for (int i = 0; i < BEATRICE_IN_HOP_LENGTH; ++i) in160[i] = in480[3 * i + 2];
  Hop(in160, out240);
  std::memset(out480, 0, sizeof(float) * kBlock);
  for (int i = 0; i < BEATRICE_OUT_HOP_LENGTH; ++i) out480[2 * i] = out240[i];
}

// one model hop (reference processor_core_2.cc:179-255, without the morph branch)
void ProcessorCore2::Hop(const float* in160, float* out240) {
  if (target_speaker_ == n_speakers_) MorphStep();
  InstallNextKeyValueBlock();  // at most one block per hop, :179-181
  alignas(64) float phone[BEATRICE_20RC0_PHONE_CHANNELS];
  Beatrice20rc0_ExtractPhone1(phone_extractor_, in160, phone, phone_context_);
  int q = 0;
  float feature[4];
  Beatrice20rc0_EstimatePitch1(pitch_estimator_, in160, &q, feature, pitch_context_);
  q = TransformPitch(q);
  RecordPitch(q);
  Beatrice20rc0_GenerateWaveform1(waveform_generator_, phone, &q, feature, out240, waveform_context_);
}

// pitch shift, intonation, correction, clamp (reference processor_core_2.cc:190-252)
int StreamingCore::TransformPitch(int q) const {
  double t = average_source_pitch_ + (static_cast<double>(q) - average_source_pitch_) * intonation_intensity_ +
             kBinsPerSemitone * pitch_shift_;
  if (pitch_correction_ != 0.0) {
    if (pitch_correction_type_ == 0) {
      const double anchor = (std::floor(t / kBinsPerSemitone) + 0.5) * kBinsPerSemitone;
      const double d = (t - anchor) * (2.0 / kBinsPerSemitone);
      t = std::abs(d) < 1e-4 ? anchor : anchor + d * std::pow(std::abs(d), -pitch_correction_) * (kBinsPerSemitone / 2.0);
    } else {
      const double anchor = std::round(t / kBinsPerSemitone) * kBinsPerSemitone;
      const double d = (t - anchor) * (2.0 / kBinsPerSemitone);
      if (pitch_correction_ > 1 - 1e-4) t = anchor;
      else if (d >= 0.0) t = anchor + std::pow(d, 1.0 / (1.0 - pitch_correction_)) * (kBinsPerSemitone / 2.0);
      else t = anchor - std::pow(-d, 1.0 / (1.0 - pitch_correction_)) * (kBinsPerSemitone / 2.0);
    }
  }
  return std::clamp(static_cast<int>(std::round(t)), 1, pitch_bins_ - 1);
}

bool ProcessorCore2::InstallNextKeyValueBlock() {  // reference processor_core_2.h:161-169
  if (kv_blocks_set_ >= BEATRICE_20RC0_N_BLOCKS) return false;
  Beatrice20rc0_SetKeyValueSpeakerEmbedding(embedding_setter_, kv_blocks_set_++, embedding_context_, waveform_context_);
  return true;
}

// reference processor_core_2.cc:293-419
ErrorCode ProcessorCore2::LoadModel(const std::filesystem::path& model_file) {
  model_file_.clear();
  ready_to_set_speaker_ = false;
  // The reference builds a NEW core for every load (processor_proxy.h:55-70), so a loaded model always starts on
  // fresh contexts: no audio history of the previous model, no device copy of its codebooks.
  RecreateContexts();
  const auto dir = model_file.parent_path();
  auto path = [&](const char* name) { return (dir / name).u8string(); };
#define BEATRICE_TRY_READ(call) \
  if (const auto err = (call)) return static_cast<ErrorCode>(err);
  BEATRICE_TRY_READ(Beatrice20rc0_ReadPhoneExtractorParameters(phone_extractor_, reinterpret_cast<const char*>(path("phone_extractor.bin").c_str())))
  BEATRICE_TRY_READ(Beatrice20rc0_ReadPitchEstimatorParameters(pitch_estimator_, reinterpret_cast<const char*>(path("pitch_estimator.bin").c_str())))
  BEATRICE_TRY_READ(Beatrice20rc0_ReadWaveformGeneratorParameters(waveform_generator_, reinterpret_cast<const char*>(path("waveform_generator.bin").c_str())))
  BEATRICE_TRY_READ(Beatrice20rc0_ReadEmbeddingSetterParameters(embedding_setter_, reinterpret_cast<const char*>(path("embedding_setter.bin").c_str())))
  const auto spk = path("speaker_embeddings.bin");
  BEATRICE_TRY_READ(Beatrice20rc0_ReadNSpeakers(reinterpret_cast<const char*>(spk.c_str()), &n_speakers_))
  if (n_speakers_ < 1) return ErrorCode::kInvalidFileSize;   // (a table without speakers: the morph lottery would draw from an empty range)
  const size_t slots = static_cast<size_t>(n_speakers_) + 1;  // + morph slot, zero-filled
  codebooks_.assign(slots * BEATRICE_20RC0_CODEBOOK_SIZE * BEATRICE_20RC0_PHONE_CHANNELS, 0.0f);
  additive_.assign(slots * BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS, 0.0f);
  formant_.assign(9 * BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS, 0.0f);
  key_value_.assign(slots * BEATRICE_20RC0_KV_LENGTH * BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS, 0.0f);
  BEATRICE_TRY_READ(Beatrice20rc0_ReadSpeakerEmbeddings(reinterpret_cast<const char*>(spk.c_str()), codebooks_.data(), additive_.data(),
                                                        formant_.data(), key_value_.data()))
#undef BEATRICE_TRY_READ
  {  // spherical-mean solvers over the speakers (reference processor_core_2.cc:384-406)
    const int lim = std::min(n_speakers_, kSphAvgMaxNSpeakers);
    mean_additive_.Initialize(n_speakers_, BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS, additive_.data(), lim);
    mean_kv_.assign(BEATRICE_20RC0_KV_LENGTH, SphericalMean());
    std::vector<float> token(static_cast<size_t>(n_speakers_) * BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS);
    for (int i = 0; i < BEATRICE_20RC0_KV_LENGTH; ++i) {
      for (int j = 0; j < n_speakers_; ++j)
        std::copy_n(key_value_.data() + (static_cast<size_t>(j) * BEATRICE_20RC0_KV_LENGTH + i) * BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS,
                    BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS, token.data() + static_cast<size_t>(j) * BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS);
      mean_kv_[i].Initialize(n_speakers_, BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS, token.data(), lim);
    }
    morph_counter_ = std::numeric_limits<int>::max();
  }
  ready_to_set_speaker_ = true;
  if (const auto err = SetTargetSpeaker(0); err != ErrorCode::kSuccess) return err;
  while (InstallNextKeyValueBlock()) {}
  model_file_ = model_file;
  ApplySpeakerMorphingWeights();
  // the reference's proxy re-syncs every parameter after a load (processor_proxy.h:95)
  SetFormantShift(formant_shift_);
  SetMinSourcePitch(min_source_pitch_);
  SetMaxSourcePitch(max_source_pitch_);
  SetVQNumNeighbors(vq_num_neighbors_);
  return ErrorCode::kSuccess;
}

// reference processor_core_2.cc:258-291
void ProcessorCore2::RecreateContexts() {
  Beatrice20rc0_DestroyPhoneContext1(phone_context_);
  Beatrice20rc0_DestroyPitchContext1(pitch_context_);
  Beatrice20rc0_DestroyWaveformContext1(waveform_context_);
  Beatrice20rc0_DestroyEmbeddingContext(embedding_context_);
  phone_context_ = Beatrice20rc0_CreatePhoneContext1();
  pitch_context_ = Beatrice20rc0_CreatePitchContext1();
  waveform_context_ = Beatrice20rc0_CreateWaveformContext1();
  embedding_context_ = Beatrice20rc0_CreateEmbeddingContext();
}

ErrorCode ProcessorCore2::ResetContext() {
  RecreateContexts();
  ErrorCode error = SetTargetSpeaker(target_speaker_);
  while (InstallNextKeyValueBlock()) {}
  for (const ErrorCode e : {SetFormantShift(formant_shift_), SetMinSourcePitch(min_source_pitch_),
                            SetMaxSourcePitch(max_source_pitch_), SetVQNumNeighbors(vq_num_neighbors_)})
    if (error == ErrorCode::kSuccess) error = e;
  return error;
}

ErrorCode StreamingCore::SetSampleRate(double sr) {  // reference processor_core_2.cc:421-429
  if (sr == sample_rate_) return ErrorCode::kSuccess;
  sample_rate_ = sr;
  ConfigureBridge(bridge_, sr);
  ReserveBlocks(std::max(reserved_block_, kDefaultMaxBlock));
  std::fill(fifo_.begin(), fifo_.end(), 0.0f);
  fifo_fill_ = 0;
  gain_in_.SetSampleRate(sr);
  gain_out_.SetSampleRate(sr);
  return ErrorCode::kSuccess;
}

ErrorCode ProcessorCore2::SetTargetSpeaker(int id) {  // reference processor_core_2.cc:431-466
  if (!ready_to_set_speaker_) return ErrorCode::kModelNotLoaded;
  if (id < 0 || id > n_speakers_) return ErrorCode::kSpeakerIDOutOfRange;
  const size_t s = static_cast<size_t>(id);
  Beatrice20rc0_SetCodebook(phone_context_, codebooks_.data() + s * BEATRICE_20RC0_CODEBOOK_SIZE * BEATRICE_20RC0_PHONE_CHANNELS);
  Beatrice20rc0_SetAdditiveSpeakerEmbedding(embedding_setter_, additive_.data() + s * BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS,
                                            embedding_context_, waveform_context_);
  Beatrice20rc0_RegisterKeyValueSpeakerEmbedding(
      embedding_setter_, key_value_.data() + s * BEATRICE_20RC0_KV_LENGTH * BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS, embedding_context_);
  target_speaker_ = id;
  kv_blocks_set_ = 0;
  return ErrorCode::kSuccess;
}

ErrorCode ProcessorCore2::SetFormantShift(double v) {  // reference processor_core_2.cc:468-481
  formant_shift_ = std::clamp(v, -2.0, 2.0);
  if (formant_.empty()) return ErrorCode::kSuccess;
  const int index = static_cast<int>(std::round(formant_shift_ * 2.0 + 4.0));
  Beatrice20rc0_SetFormantShiftEmbedding(embedding_setter_, formant_.data() + static_cast<size_t>(index) * BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS,
                                         embedding_context_, waveform_context_);
  return ErrorCode::kSuccess;
}
ErrorCode StreamingCore::SetPitchShift(double v) { pitch_shift_ = std::clamp(v, -24.0, 24.0); return ErrorCode::kSuccess; }
ErrorCode StreamingCore::SetInputGain(double db) { gain_in_.SetTargetGain(db); return ErrorCode::kSuccess; }
ErrorCode StreamingCore::SetOutputGain(double db) { gain_out_.SetTargetGain(db); return ErrorCode::kSuccess; }
ErrorCode StreamingCore::SetAverageSourcePitch(double v) { average_source_pitch_ = std::clamp(v, 0.0, 128.0); return ErrorCode::kSuccess; }
ErrorCode StreamingCore::SetIntonationIntensity(double v) { intonation_intensity_ = v; return ErrorCode::kSuccess; }
ErrorCode StreamingCore::SetPitchCorrection(double v) { pitch_correction_ = std::clamp(v, 0.0, 1.0); return ErrorCode::kSuccess; }
ErrorCode StreamingCore::SetPitchCorrectionType(int type) {
  if (type < 0 || type > 1) return ErrorCode::kInvalidPitchCorrectionType;
  pitch_correction_type_ = type;
  return ErrorCode::kSuccess;
}
int StreamingCore::NoteToBin(double note) const {  // reference processor_core_2.cc:561-583
  const int q = static_cast<int>(std::round((note - 33.0) * kBinsPerSemitone));
  return std::clamp(q, 1, pitch_bins_ - 1);
}
ErrorCode ProcessorCore2::SetMinSourcePitch(double v) {
  min_source_pitch_ = std::clamp(v, 0.0, 128.0);
  Beatrice20rc0_SetMinQuantizedPitch(pitch_context_, NoteToBin(min_source_pitch_));
  return ErrorCode::kSuccess;
}
ErrorCode ProcessorCore2::SetMaxSourcePitch(double v) {
  max_source_pitch_ = std::clamp(v, 0.0, 128.0);
  Beatrice20rc0_SetMaxQuantizedPitch(pitch_context_, NoteToBin(max_source_pitch_));
  return ErrorCode::kSuccess;
}
ErrorCode ProcessorCore2::SetVQNumNeighbors(int k) {  // reference processor_core_2.cc:585-590
  vq_num_neighbors_ = std::clamp(k, 0, 8);
  Beatrice20rc0_SetVQNumNeighbors(phone_context_, vq_num_neighbors_);
  return ErrorCode::kSuccess;
}

// ---- morphing ------------------------------------------------------------------------------------
std::array<float, kMaxNSpeakers> PrepareVoiceMorphWeights(std::array<float, kMaxNSpeakers> w, int speaker_count) {
  if (speaker_count <= 0) return {};
  const int count = std::min(speaker_count, kMaxNSpeakers);
  for (int i = count; i < kMaxNSpeakers; ++i) w[count - 1] += w[i];
  std::fill(w.begin() + count, w.end(), 0.0f);
  for (int i = 0; i < count; ++i) if (w[i] < 0.01f) w[i] = 0.0f;
  return w;
}

ErrorCode ProcessorCore2::SetSpeakerMorphingWeights(const std::array<float, kMaxNSpeakers>& weights) {
  if (weights == morph_weights_) return ErrorCode::kSuccess;  // reference processor_core_2.cc:498-505
  morph_weights_ = weights;
  return ApplySpeakerMorphingWeights();
}

// weight preparation: overflow speakers folded into the last one, < 0.01 dropped (reference
// voice_morph_state.h:87-104), then the 8 largest kept (processor_core_2.cc:507-532)
ErrorCode ProcessorCore2::ApplySpeakerMorphingWeights() {
  if (!ready_to_set_speaker_) return ErrorCode::kSuccess;
  const std::array<float, kMaxNSpeakers> w = PrepareVoiceMorphWeights(morph_weights_, n_speakers_);
  std::iota(morph_order_.data(), morph_order_.data() + n_speakers_, 0);
  std::sort(morph_order_.data(), morph_order_.data() + n_speakers_, [&w](const int a, const int b) -> bool { return w[a] > w[b]; });
  morph_pruned_.fill(0.0f);
  const int keep = std::min(n_speakers_, kSphAvgMaxNSpeakers);
  for (int i = 0; i < keep; ++i) morph_pruned_[morph_order_[i]] = w[morph_order_[i]];
  morph_counter_ = 0;  // the averages are recomputed over the next hops, not here
  return ErrorCode::kSuccess;
}

// the morph branch of one hop (reference processor_core_2.cc:51-177, lottery variant :94-121)
void ProcessorCore2::MorphStep() {
  const size_t cb = static_cast<size_t>(BEATRICE_20RC0_CODEBOOK_SIZE) * BEATRICE_20RC0_PHONE_CHANNELS;
  const size_t hid = BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS;
  const size_t kvc = BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS, kvl = BEATRICE_20RC0_KV_LENGTH;
  {  // codebook: one real speaker per hop, drawn with the morph weights as odds
    const int n_weights = std::min(n_speakers_, kSphAvgMaxNSpeakers);
    float sum = 0.0f;
    for (int i = 0; i < n_weights; ++i) sum += morph_pruned_[morph_order_[i]];
    int idx = morph_order_[0];
    if (sum <= std::numeric_limits<float>::epsilon()) {
      idx = std::uniform_int_distribution<int>(0, n_speakers_ - 1)(lottery_);
    } else {
      float r = std::uniform_real_distribution<float>(0.0f, sum)(lottery_);
      for (int i = 0; i < n_weights; ++i) {
        const int speaker = morph_order_[i];
        r -= morph_pruned_[speaker];
        if (r < 0.0f) { idx = speaker; break; }
      }
    }
    Beatrice20rc0_SetCodebook(phone_context_, codebooks_.data() + static_cast<size_t>(idx) * cb);
  }
  if (morph_counter_ == 0) {  // additive embedding: in one go, on the hop after a weight change
    mean_additive_.SetWeights(n_speakers_, morph_pruned_.data(), morph_order_.data());
    for (int j = 0; j < kSphAvgMaxNUpdates; ++j) if (mean_additive_.Update()) break;
    float* slot = additive_.data() + static_cast<size_t>(n_speakers_) * hid;
    mean_additive_.Result(slot);
    Beatrice20rc0_SetAdditiveSpeakerEmbedding(embedding_setter_, slot, embedding_context_, waveform_context_);
  }
  if (morph_counter_ < kSphAvgMaxNState) {  // key/value tokens: a quarter of them per hop
    const int first = static_cast<int>(kvl) * morph_counter_ / kSphAvgMaxNState;
    const int last = static_cast<int>(kvl) * (morph_counter_ + 1) / kSphAvgMaxNState;
    for (int i = first; i < last; ++i) {
      mean_kv_[i].SetWeights(n_speakers_, morph_pruned_.data(), morph_order_.data());
      for (int j = 0; j < kSphAvgMaxNUpdates; ++j) if (mean_kv_[i].Update()) break;
      mean_kv_[i].Result(key_value_.data() + (static_cast<size_t>(n_speakers_) * kvl + i) * kvc);
    }
  } else if (morph_counter_ == kSphAvgMaxNState) {
    Beatrice20rc0_RegisterKeyValueSpeakerEmbedding(embedding_setter_, key_value_.data() + static_cast<size_t>(n_speakers_) * kvl * kvc,
                                                   embedding_context_);
    kv_blocks_set_ = 0;
  }
  if (morph_counter_ <= kSphAvgMaxNState) ++morph_counter_;
}

}  // namespace beatrice_amd

// ---- plain-C view of the classes for FFI callers and the tests ----------------------------------
using beatrice_amd::ProcessorCoreBase;
namespace beatrice_amd {
std::unique_ptr<ProcessorCoreBase> MakeProcessorCore(int version, double sample_rate) {  // reference processor_proxy.h:57-70
  switch (version) {
    case 0: case 1: return std::make_unique<ProcessorCoreLegacy>(sample_rate, version);
    case 2: return std::make_unique<ProcessorCore2>(sample_rate);
    default: return nullptr;
  }
}
}  // namespace beatrice_amd
static ProcessorCoreBase* core(void* p) { return static_cast<ProcessorCoreBase*>(p); }
extern "C" {
void* BeatriceHost_Create(double sample_rate) { return beatrice_amd::MakeProcessorCore(2, sample_rate).release(); }
// version: 0 = 2.0.0-alpha.2, 1 = 2.0.0-beta.1, 2 = 2.0.0-rc.0; null for anything else
void* BeatriceHost_CreateVersion(double sample_rate, int version) { return beatrice_amd::MakeProcessorCore(version, sample_rate).release(); }
void BeatriceHost_Destroy(void* p) { delete core(p); }
int BeatriceHost_GetVersion(void* p) { return core(p)->GetVersion(); }
int BeatriceHost_LoadModel(void* p, const char* toml_path) { return static_cast<int>(core(p)->LoadModel(toml_path)); }
int BeatriceHost_Process(void* p, const float* in, float* out, int n) { return static_cast<int>(core(p)->Process(in, out, n)); }
int BeatriceHost_ResetContext(void* p) { return static_cast<int>(core(p)->ResetContext()); }
int BeatriceHost_SetSampleRate(void* p, double v) { return static_cast<int>(core(p)->SetSampleRate(v)); }
int BeatriceHost_ReserveBlocks(void* p, int max_block) { if (max_block < 1) return -1; core(p)->ReserveBlocks(max_block); return 0; }
unsigned long long BeatriceHost_BufferFingerprint(void* p) { return core(p)->BufferFingerprint(); }
int BeatriceHost_SetTargetSpeaker(void* p, int v) { return static_cast<int>(core(p)->SetTargetSpeaker(v)); }
int BeatriceHost_SetFormantShift(void* p, double v) { return static_cast<int>(core(p)->SetFormantShift(v)); }
int BeatriceHost_SetPitchShift(void* p, double v) { return static_cast<int>(core(p)->SetPitchShift(v)); }
int BeatriceHost_SetInputGain(void* p, double v) { return static_cast<int>(core(p)->SetInputGain(v)); }
int BeatriceHost_SetOutputGain(void* p, double v) { return static_cast<int>(core(p)->SetOutputGain(v)); }
int BeatriceHost_SetAverageSourcePitch(void* p, double v) { return static_cast<int>(core(p)->SetAverageSourcePitch(v)); }
int BeatriceHost_SetIntonationIntensity(void* p, double v) { return static_cast<int>(core(p)->SetIntonationIntensity(v)); }
int BeatriceHost_SetPitchCorrection(void* p, double v) { return static_cast<int>(core(p)->SetPitchCorrection(v)); }
int BeatriceHost_SetPitchCorrectionType(void* p, int v) { return static_cast<int>(core(p)->SetPitchCorrectionType(v)); }
int BeatriceHost_SetMinSourcePitch(void* p, double v) { return static_cast<int>(core(p)->SetMinSourcePitch(v)); }
int BeatriceHost_SetMaxSourcePitch(void* p, double v) { return static_cast<int>(core(p)->SetMaxSourcePitch(v)); }
int BeatriceHost_SetVQNumNeighbors(void* p, int v) { return static_cast<int>(core(p)->SetVQNumNeighbors(v)); }
int BeatriceHost_SetSpeakerMorphingWeights(void* p, const float* weights, int n) {
  std::array<float, beatrice_amd::kMaxNSpeakers> w{};
  for (int i = 0; i < n && i < beatrice_amd::kMaxNSpeakers; ++i) w[i] = weights[i];
  return static_cast<int>(core(p)->SetSpeakerMorphingWeights(w));
}
void BeatriceHost_SetMorphSeed(void* p, unsigned seed) { core(p)->SetMorphSeed(seed); }
int BeatriceHost_NumSpeakers(void* p) { return core(p)->n_speakers(); }
void BeatriceHost_EnablePitchTrace(void* p, int capacity) { core(p)->EnablePitchTrace(capacity); }
int BeatriceHost_TakePitchTrace(void* p, int* out, int cap) {
  const auto t = core(p)->TakePitchTrace();
  const int held = static_cast<int>(t.size());
  const int n = held < cap ? held : (cap > 0 ? cap : 0);
  for (int i = 0; i < n; ++i) out[i] = t[held - n + i];   // the NEWEST n entries, oldest of them first (beatrice_host.h)
  return held;
}
}

// weighted spherical mean as the morph branch computes it (Initialize -> SetWeights -> <= max_updates
// Update -> Result); returns the number of updates performed
extern "C" int BeatriceHost_SphericalMean(int dim, int n_points, const float* points, const float* weights, const int* order,
                                          int limit, int max_updates, float* out) {
  beatrice_amd::SphericalMean m;
  m.Initialize(n_points, dim, points, limit);
  m.SetWeights(n_points, weights, order);
  int it = 0;
  for (; it < max_updates; ++it) if (m.Update()) break;
  m.Result(out);
  return it;
}
