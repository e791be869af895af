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
#include <memory>
#include <random>
#include <vector>

#include "beatrice_abi.h"
#include "model_config.h"  // kMaxNSpeakers
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

// What the proxy drives (reference ProcessorCoreBase, src/common/processor_core.h:22-92): every setter a core does
// not have succeeds and does nothing.
class ProcessorCoreBase {
 public:
  ProcessorCoreBase() = default;
  ProcessorCoreBase(const ProcessorCoreBase&) = delete;
  ProcessorCoreBase& operator=(const ProcessorCoreBase&) = delete;
  virtual ~ProcessorCoreBase() = default;
  virtual int GetVersion() const = 0;
  virtual ErrorCode Process(const float* input, float* output, int n_samples) = 0;
  virtual ErrorCode ResetContext() { return ErrorCode::kSuccess; }
  // `model_file` is the package's .toml path; as in the reference only its directory is used
  virtual ErrorCode LoadModel(const std::filesystem::path& /*model_file*/) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetSampleRate(double) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetTargetSpeaker(int) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetFormantShift(double) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetPitchShift(double) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetInputGain(double) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetOutputGain(double) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetAverageSourcePitch(double) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetIntonationIntensity(double) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetPitchCorrection(double) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetPitchCorrectionType(int) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetMinSourcePitch(double) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetMaxSourcePitch(double) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetVQNumNeighbors(int) { return ErrorCode::kSuccess; }
  virtual ErrorCode SetSpeakerMorphingWeights(const std::array<float, kMaxNSpeakers>&) { return ErrorCode::kSuccess; }
  // not in the reference: the lottery seed of rc.0 morphing, and two test hooks
  virtual void SetMorphSeed(std::uint32_t) {}
  virtual int n_speakers() const { return 0; }
  virtual std::vector<int> TakePitchTrace() { return {}; }
  virtual void EnablePitchTrace(int /*capacity*/) {}
  // (ours, not the reference's: see StreamingCore::ReserveBlocks)
  virtual void ReserveBlocks(int /*max_block*/) {}
  virtual unsigned long long BufferFingerprint() const { return 0; }
};

// The part every generation's core shares (reference processor_core_{0,1,2}.cc hold three copies of it): the guards of
// Process(), input gain -> host rate to 48 kHz -> 480-sample FIFO -> every 3rd sample -> Hop() of the generation ->
// zero-stuffed 480 -> host rate -> output gain, and the pitch parameters with their transform.
class StreamingCore : public ProcessorCoreBase {
 public:
  explicit StreamingCore(double sample_rate, int pitch_bins);
  ErrorCode Process(const float* input, float* output, int n_samples) final;
  ErrorCode SetSampleRate(double sample_rate) override;
  ErrorCode SetPitchShift(double pitch_shift) override;
  ErrorCode SetInputGain(double db) override;
  ErrorCode SetOutputGain(double db) override;
  ErrorCode SetAverageSourcePitch(double average_pitch) override;
  ErrorCode SetIntonationIntensity(double intonation_intensity) override;
  ErrorCode SetPitchCorrection(double pitch_correction) override;
  ErrorCode SetPitchCorrectionType(int pitch_correction_type) override;
  int n_speakers() const override { return n_speakers_; }
  // test hook, OFF unless a test enables it: the bins handed to GenerateWaveform1 (pitch transform output) in a ring of fixed
  // capacity allocated by EnablePitchTrace -- the per-hop path only writes a slot (no allocation on the audio thread,
  // src/common/resample.h:303-305); TakePitchTrace returns the newest <= capacity entries in order and empties the ring
  void EnablePitchTrace(int capacity) override;
  std::vector<int> TakePitchTrace() override;

 protected:
  bool IsLoaded() const { return !model_file_.empty(); }
  virtual void Hop(const float* in160, float* out240) = 0;
  // generation-specific guard of Process() behind the common ones (kSuccess = go on)
  virtual ErrorCode Preflight() const { return ErrorCode::kSuccess; }
  int TransformPitch(int q) const;   // shift, intonation, correction, clamp to [1, pitch_bins - 1]
  int NoteToBin(double note) const;  // MIDI note -> bin, same clamp

  std::filesystem::path model_file_;
  double sample_rate_;
  const int pitch_bins_;
  int target_speaker_ = 0, n_speakers_ = 0, pitch_correction_type_ = 0;
  double formant_shift_ = 0.0, pitch_shift_ = 0.0, average_source_pitch_ = 52.0, intonation_intensity_ = 1.0;
  double pitch_correction_ = 0.0, min_source_pitch_ = 33.125, max_source_pitch_ = 80.875;
  void RecordPitch(int q) {
    if (pitch_trace_.empty()) return;
    pitch_trace_[static_cast<size_t>(trace_count_ % pitch_trace_.size())] = q;
    ++trace_count_;
  }

 private:
  std::vector<int> pitch_trace_;      // ring, size = capacity (0: tracing off)
  unsigned long long trace_count_ = 0;  // entries written since the last TakePitchTrace
  void Block480(const float* in480, float* out480);
  void Reblock(const float* in, float* out, int n);
  RateBridge bridge_;
  GainRamp gain_in_, gain_out_;
  std::vector<float> fifo_;  // 480-sample block adapter
  int fifo_fill_ = 0;
  std::vector<float> io_, work_, scratch_;
  int reserved_block_ = 0;

 public:
  // The per-block buffers are sized HERE, off the audio thread, for host blocks of up to max_block samples (the VST shell
  // announces its largest block before processing starts, IAudioProcessor::setupProcessing; the reference pre-sizes its
  // buffers the same way, src/common/resample.h:303-305): Process() then never allocates.  Called by the constructor and
  // by SetSampleRate with the default below; a host with larger blocks calls it itself.  A block beyond the reserve still
  // works (the vectors grow, once).
  static constexpr int kDefaultMaxBlock = 8192;
  void ReserveBlocks(int max_block) override;
  unsigned long long BufferFingerprint() const override;   // test hook: storage addresses and capacities of the per-block buffers
};

class ProcessorCore2 final : public StreamingCore {
 public:
  explicit ProcessorCore2(double sample_rate);
  ~ProcessorCore2() override;

  int GetVersion() const override { return 2; }
  ErrorCode ResetContext() override;
  ErrorCode LoadModel(const std::filesystem::path& model_file) override;
  ErrorCode SetTargetSpeaker(int target_speaker) override;
  ErrorCode SetFormantShift(double formant_shift) override;
  ErrorCode SetMinSourcePitch(double min_source_pitch) override;
  ErrorCode SetMaxSourcePitch(double max_source_pitch) override;
  ErrorCode SetVQNumNeighbors(int vq_num_neighbors) override;
  static constexpr int kMaxNSpeakers = beatrice_amd::kMaxNSpeakers;
  static constexpr int kSphAvgMaxNSpeakers = 8;  // reference processor_core_2.h:26
  ErrorCode SetSpeakerMorphingWeights(const std::array<float, kMaxNSpeakers>& weights) override;
  void SetMorphSeed(std::uint32_t seed) override { lottery_.seed(seed); }  // the reference seeds from std::random_device

 private:
  void Hop(const float* in160, float* out240) override;
  bool InstallNextKeyValueBlock();
  void RecreateContexts();
  ErrorCode ApplySpeakerMorphingWeights();
  void MorphStep();

  int vq_num_neighbors_ = 0, kv_blocks_set_ = 0;
  bool ready_to_set_speaker_ = false;

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

// weight preparation of morphing: overflow speakers folded into the last one, < 0.01 dropped (reference
// voice_morph_state.h:87-104)
std::array<float, kMaxNSpeakers> PrepareVoiceMorphWeights(std::array<float, kMaxNSpeakers> weights, int speaker_count);

// The two older generations (reference src/common/processor_core_{0,1}.{h,cc}; the two files differ in the prefix of the
// library calls and GetVersion() only).  What differs from ProcessorCore2: no codebook, no key/value embeddings; the
// waveform generator takes ONE 256-float speaker vector per hop = speaker_embeddings[target] + formant_shift_embeddings[
// round(2 shift + 4)]; morphing (target == n_speakers) is the spherical mean of the speaker vectors over ALL speakers
// with a non-zero prepared weight, advanced one update per hop and written to slot n_speakers while it has not
// converged (processor_core_1.cc:121-130, 258-270); SetTargetSpeaker only rejects negative ids and Process() answers
// kSpeakerIDOutOfRange with zeros for an id beyond n_speakers (:35-40); ResetContext re-applies the source pitch
// range only (:145-163).  `Lib` names the generation's entry points (processor_core_legacy.cc).
struct LegacyLib;
class ProcessorCoreLegacy final : public StreamingCore {
 public:
  ProcessorCoreLegacy(double sample_rate, int version);  // version 0 = 2.0.0-alpha.2, 1 = 2.0.0-beta.1
  ~ProcessorCoreLegacy() override;
  int GetVersion() const override { return version_; }
  ErrorCode ResetContext() override;
  ErrorCode LoadModel(const std::filesystem::path& model_file) override;
  ErrorCode SetTargetSpeaker(int target_speaker) override;
  ErrorCode SetFormantShift(double formant_shift) override;
  ErrorCode SetMinSourcePitch(double min_source_pitch) override;
  ErrorCode SetMaxSourcePitch(double max_source_pitch) override;
  ErrorCode SetSpeakerMorphingWeights(const std::array<float, kMaxNSpeakers>& weights) override;

 private:
  void Hop(const float* in160, float* out240) override;
  ErrorCode Preflight() const override;
  ErrorCode ApplySpeakerMorphingWeights();
  const int version_;
  const LegacyLib& lib_;
  void *phone_extractor_, *pitch_estimator_, *waveform_generator_, *phone_context_, *pitch_context_, *waveform_context_;
  std::vector<float> speaker_embeddings_, formant_shift_embeddings_;  // (n_speakers + 1) x 256; 9 x 256
  std::array<float, kMaxNSpeakers> morph_weights_{};
  SphericalMean mean_;
};

// version of a package's `model.version` -> a core on this library; nullptr for a version this library does not know
std::unique_ptr<ProcessorCoreBase> MakeProcessorCore(int version, double sample_rate);

}  // namespace beatrice_amd
