// processor_core.h -- C++ host layer above the beatrice C-ABI: this project's counterpart of the
// reference's ProcessorCore2 (reference src/common/processor_core_2.{h,cc}), for callers that want
// "any sample rate, any block size in; same length out" per stream.
//
// Same member names, argument meaning and error behaviour as the reference's ProcessorCoreBase /
// ProcessorCore2 interface (reference src/common/processor_core.h:22-92), so code written against
// the reference class ports by changing the type name.  What it does per Process() call is the
// reference's chain (processor_core_2.cc:24-48, 50-256):
//   input gain -> host rate to 48 kHz -> 480-sample FIFO -> every 3rd sample (160 @16 kHz)
//   -> [one K/V block if pending] ExtractPhone1, EstimatePitch1, pitch transform, GenerateWaveform1
//   -> zero-stuff 240 -> 480 -> 48 kHz to host rate -> output gain.
// The three library calls go to whatever implements include/beatrice_abi.h at link time: the HIP
// library in the product build (libbeatrice_host.so).  Speaker morphing (target speaker == n_speakers:
// SetSpeakerMorphingWeights, reference processor_core_2.cc:51-177, 498-532) runs on the host as in the
// reference: spherical means of the additive and key/value embeddings spread over four hops, a weighted
// lottery for the codebook; the only deviation is that the lottery seed can be fixed (SetMorphSeed).
#pragma once
#include <array>
#include <cstdint>
#include <filesystem>
#include <limits>
#include <random>
#include <vector>

#include "beatrice_abi.h"
#include "spherical_mean.h"

namespace beatrice_amd {

// numeric values are the reference's (reference src/common/error.h:11-25)
enum class ErrorCode : std::uint8_t {
  kSuccess = 0,
  kFileOpenError = Beatrice_kFileOpenError,
  kFileTooSmall = Beatrice_kFileTooSmall,
  kFileTooLarge = Beatrice_kFileTooLarge,
  kInvalidFileSize = Beatrice_kInvalidFileSize,
  kTOMLSyntaxError,
  kInvalidModelConfig,
  kSpeakerIDOutOfRange,
  kInvalidPitchCorrectionType,
  kModelNotLoaded,
  kResamplerNotReady,
  kGainNotReady,
  kUnknownError,
};

// dB-ramped gain, 2 dB/ms slew (behaviour of reference src/common/gain.h:41-71)
class GainRamp {
 public:
  explicit GainRamp(double sample_rate, double gain_db = 0.0) : rate_(sample_rate), target_db_(gain_db), now_db_(gain_db) {}
  void SetTargetGain(double db) { target_db_ = db; }
  void SetSampleRate(double sr) { rate_ = sr; }
  bool IsReady() const { return rate_ > 1e-5; }
  void Apply(const float* in, float* out, int n);

 private:
  double rate_, target_db_, now_db_;
};

// Rational polyphase resampler pair sharing one set of fractional clocks (behaviour of reference
// src/common/resample.h:76-271): outer = host rate, inner = 48 kHz.
class RateBridge {
 public:
  void Configure(double outer_rate, double inner_rate, double cutoff_in, double cutoff_out);
  bool IsReady() const { return ready_; }
  void ToInner(const std::vector<float>& in, std::vector<float>& out);
  void ToOuter(const std::vector<float>& in, std::vector<float>& out);
  int ratio_high() const { return hi_; }
  int ratio_low() const { return lo_; }

 private:
  struct History {  // the newest `size` samples, zeros before the stream starts
    std::vector<float> ring;
    int head = 0;
    void Reset(int size) { ring.assign(size, 0.0f); head = 0; }
    void Push(float v) { ring[head] = v; head = head + 1 == (int)ring.size() ? 0 : head + 1; }
    float Back(int k) const { int i = head - k; return ring[i < 0 ? i + (int)ring.size() : i]; }
  };
  void Decimate(const std::vector<float>& in, std::vector<float>& out);
  void Interpolate(const std::vector<float>& in, std::vector<float>& out);
  bool ready_ = false, high_is_outer_ = true;
  int hi_ = 1, lo_ = 1, phase_down_ = 0, phase_up_ = 0;
  std::vector<float> taps_down_, taps_up_;
  History hist_high_, hist_low_;
};

class ProcessorCore2 {
 public:
  explicit ProcessorCore2(double sample_rate);
  ~ProcessorCore2();
  ProcessorCore2(const ProcessorCore2&) = delete;
  ProcessorCore2& operator=(const ProcessorCore2&) = delete;

  int GetVersion() const { return 2; }
  ErrorCode Process(const float* input, float* output, int n_samples);
  ErrorCode ResetContext();
  // `model_file` is the package's .toml path; as in the reference only its directory is used
  ErrorCode LoadModel(const std::filesystem::path& model_file);
  ErrorCode SetSampleRate(double sample_rate);
  ErrorCode SetTargetSpeaker(int target_speaker);
  ErrorCode SetFormantShift(double formant_shift);
  ErrorCode SetPitchShift(double pitch_shift);
  ErrorCode SetInputGain(double db);
  ErrorCode SetOutputGain(double db);
  ErrorCode SetAverageSourcePitch(double average_pitch);
  ErrorCode SetIntonationIntensity(double intonation_intensity);
  ErrorCode SetPitchCorrection(double pitch_correction);
  ErrorCode SetPitchCorrectionType(int pitch_correction_type);
  ErrorCode SetMinSourcePitch(double min_source_pitch);
  ErrorCode SetMaxSourcePitch(double max_source_pitch);
  ErrorCode SetVQNumNeighbors(int vq_num_neighbors);
  static constexpr int kMaxNSpeakers = 256;      // reference src/common/model_config.h:17
  static constexpr int kSphAvgMaxNSpeakers = 8;  // reference processor_core_2.h:26
  ErrorCode SetSpeakerMorphingWeights(const std::array<float, kMaxNSpeakers>& weights);
  void SetMorphSeed(std::uint32_t seed) { lottery_.seed(seed); }  // the reference seeds from std::random_device
  int n_speakers() const { return n_speakers_; }
  // test hook: bins handed to GenerateWaveform1 since the last call (pitch transform output)
  std::vector<int> TakePitchTrace() { std::vector<int> t; t.swap(pitch_trace_); return t; }

 private:
  bool IsLoaded() const { return !model_file_.empty(); }
  void Hop(const float* in160, float* out240);
  void Block480(const float* in480, float* out480);
  void Reblock(const float* in, float* out, int n);
  bool InstallNextKeyValueBlock();
  void RecreateContexts();
  int TransformPitch(int q) const;
  ErrorCode ApplySpeakerMorphingWeights();
  void MorphStep();

  std::filesystem::path model_file_;
  double sample_rate_;
  int target_speaker_ = 0, n_speakers_ = 0, pitch_correction_type_ = 0, vq_num_neighbors_ = 0, kv_blocks_set_ = 0;
  double formant_shift_ = 0.0, pitch_shift_ = 0.0, average_source_pitch_ = 52.0, intonation_intensity_ = 1.0;
  double pitch_correction_ = 0.0, min_source_pitch_ = 33.125, max_source_pitch_ = 80.875;
  bool ready_to_set_speaker_ = false;

  RateBridge bridge_;
  GainRamp gain_in_, gain_out_;
  std::vector<float> fifo_;  // 480-sample block adapter
  int fifo_fill_ = 0;
  std::vector<float> io_, work_, scratch_;
  std::vector<int> pitch_trace_;

  Beatrice20rc0_PhoneExtractor* phone_extractor_;
  Beatrice20rc0_PitchEstimator* pitch_estimator_;
  Beatrice20rc0_WaveformGenerator* waveform_generator_;
  Beatrice20rc0_EmbeddingSetter* embedding_setter_;
  Beatrice20rc0_PhoneContext1* phone_context_;
  Beatrice20rc0_PitchContext1* pitch_context_;
  Beatrice20rc0_WaveformContext1* waveform_context_;
  Beatrice20rc0_EmbeddingContext* embedding_context_;
  // caller-owned tables, (n_speakers + 1) slots like the reference (last = morph result)
  std::vector<float> codebooks_, additive_, formant_, key_value_;
  // morphing (reference processor_core_2.h:137-152)
  static constexpr int kSphAvgMaxNUpdates = 4, kSphAvgMaxNState = 4;
  std::array<float, kMaxNSpeakers> morph_weights_{}, morph_pruned_{};
  std::array<int, kMaxNSpeakers> morph_order_{};
  int morph_counter_ = std::numeric_limits<int>::max();
  std::mt19937 lottery_{std::random_device{}()};
  SphericalMean mean_additive_;
  std::vector<SphericalMean> mean_kv_;
};

}  // namespace beatrice_amd
