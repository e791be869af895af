// parameter_state.h -- the host's parameter table and its preset / project-state wire format (counterpart of the
// reference's ParameterSchema + ParameterState, reference src/common/parameter_schema.{h,cc},
// parameter_state.{h,cc}).
//
// Wire format (parameter_state.cc:68-147), little endian, no padding, records in ascending id order:
//     int16 id | int32 type | payload        type 0: int32     type 1: float64     type 2: int32 length + bytes (UTF-8)
// A VST host wraps it as int32 size + blob (reference src/vst/processor.cc:233-268): FrameState / UnframeState.
// Reading starts from the defaults and overwrites what the stream holds (ReadOrSetDefault, :128-133); a truncated
// record is ErrorCode::kFileTooSmall, an unknown type ErrorCode::kUnknownError; ids the table does not know are kept
// (the reference keeps them too: SetValue inserts).  A record of a KNOWN id whose type is not the schema's is dropped (the
// default stays): state blobs are untrusted input, and a wrongly typed value would make every later typed read of that
// id throw (the reference's std::get would) -- here nothing typed is ever stored for an id of another kind.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <variant>
#include <vector>

#include "model_config.h"
#include "processor_core.h"

namespace beatrice_amd {

constexpr int kMaxNVoiceMorphMarkers = 8, kDefaultNVoiceMorphMarkers = 4;  // reference voice_morph_state.h:14-15

// reference parameter_schema.h:42-69
namespace param_id {
constexpr std::int16_t kModel = 1, kVoice = 2, kFormantShift = 3, kPitchShift = 4, kAverageSourcePitch = 5, kLock = 6, kInputGain = 7,
                       kOutputGain = 8, kIntonationIntensity = 9, kPitchCorrection = 10, kPitchCorrectionType = 11, kMinSourcePitch = 12,
                       kMaxSourcePitch = 13, kVQNumNeighbors = 14, kVoiceMorphCursorX = 15, kVoiceMorphCursorY = 16,
                       kVoiceMorphFalloff = 17, kVoiceMorphMarkerCount = 18, kVoiceMorphMarkerVoiceBase = 19,
                       kVoiceMorphMarkerXBase = kVoiceMorphMarkerVoiceBase + kMaxNVoiceMorphMarkers,
                       kVoiceMorphMarkerYBase = kVoiceMorphMarkerXBase + kMaxNVoiceMorphMarkers, kAverageTargetPitchBase = 100,
                       kEnd = kAverageTargetPitchBase + kMaxNSpeakers + 1;
}

struct ParameterInfo {
  enum Kind { kInt = 0, kNumber = 1, kString = 2 } kind;  // = the wire format's type index
  double def, lo, hi;                                      // numbers: default and range; lists: default, 0, count - 1
};

// id -> kind / default / range: the values of reference parameter_schema.cc:51-477
inline const std::map<std::int16_t, ParameterInfo>& Schema() {
  static const std::map<std::int16_t, ParameterInfo> schema = [] {
    std::map<std::int16_t, ParameterInfo> s;
    using namespace param_id;
    s[kModel] = {ParameterInfo::kString, 0, 0, 0};
    s[kVoice] = {ParameterInfo::kInt, 0, 0, kMaxNSpeakers};
    s[kFormantShift] = {ParameterInfo::kNumber, 0.0, -2.0, 2.0};
    s[kPitchShift] = {ParameterInfo::kNumber, 0.0, -24.0, 24.0};
    s[kAverageSourcePitch] = {ParameterInfo::kNumber, 52.0, 0.0, 128.0};
    s[kLock] = {ParameterInfo::kInt, 0, 0, 1};
    s[kInputGain] = {ParameterInfo::kNumber, 0.0, -60.0, 20.0};
    s[kOutputGain] = {ParameterInfo::kNumber, 0.0, -60.0, 20.0};
    s[kIntonationIntensity] = {ParameterInfo::kNumber, 1.0, -1.0, 3.0};
    s[kPitchCorrection] = {ParameterInfo::kNumber, 0.0, 0.0, 1.0};
    s[kPitchCorrectionType] = {ParameterInfo::kInt, 0, 0, 1};
    s[kMinSourcePitch] = {ParameterInfo::kNumber, 33.125, 0.0, 128.0};
    s[kMaxSourcePitch] = {ParameterInfo::kNumber, 80.875, 0.0, 128.0};
    s[kVQNumNeighbors] = {ParameterInfo::kNumber, 0.0, 0.0, 8.0};
    s[kVoiceMorphCursorX] = {ParameterInfo::kNumber, 0.5, 0.0, 1.0};
    s[kVoiceMorphCursorY] = {ParameterInfo::kNumber, 0.5, 0.0, 1.0};
    s[kVoiceMorphFalloff] = {ParameterInfo::kNumber, 2.0, 0.0, 4.0};
    s[kVoiceMorphMarkerCount] = {ParameterInfo::kNumber, (double)kDefaultNVoiceMorphMarkers, 1.0, (double)kMaxNVoiceMorphMarkers};
    // default markers (reference voice_morph_state.h:36-41; float literals widened to double as the schema does)
    const float mx[kMaxNVoiceMorphMarkers] = {0.18f, 0.82f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f};
    const float my[kMaxNVoiceMorphMarkers] = {0.5f, 0.5f, 0.18f, 0.82f, 0.5f, 0.5f, 0.5f, 0.5f};
    for (int i = 0; i < kMaxNVoiceMorphMarkers; ++i) {
      s[(std::int16_t)(kVoiceMorphMarkerVoiceBase + i)] = {ParameterInfo::kNumber, i < 4 ? (double)i : 0.0, 0.0, (double)(kMaxNSpeakers - 1)};
      s[(std::int16_t)(kVoiceMorphMarkerXBase + i)] = {ParameterInfo::kNumber, (double)mx[i], 0.0, 1.0};
      s[(std::int16_t)(kVoiceMorphMarkerYBase + i)] = {ParameterInfo::kNumber, (double)my[i], 0.0, 1.0};
    }
    for (int i = 0; i < kMaxNSpeakers + 1; ++i) s[(std::int16_t)(kAverageTargetPitchBase + i)] = {ParameterInfo::kNumber, 60.0, 0.0, 128.0};
    return s;
  }();
  return schema;
}

class ParameterState {
 public:
  using Value = std::variant<int, double, std::string>;
  void SetDefaultValues() {
    for (const auto& [id, info] : Schema()) {
      if (info.kind == ParameterInfo::kInt) values_[id] = (int)info.def;
      else if (info.kind == ParameterInfo::kNumber) values_[id] = info.def;
      else values_[id] = std::string();
    }
  }
  // false (nothing stored) when `id` is in the schema with another kind
  bool Set(std::int16_t id, Value v) {
    if (!KindOk(id, v)) return false;
    values_[id] = std::move(v);
    return true;
  }
  static bool KindOk(std::int16_t id, const Value& v) {
    const auto it = Schema().find(id);
    return it == Schema().end() || (int)v.index() == (int)it->second.kind;
  }
  // nullptr for an id that holds nothing
  const Value* Find(std::int16_t id) const { const auto it = values_.find(id); return it == values_.end() ? nullptr : &it->second; }
  const Value& Get(std::int16_t id) const { return values_.at(id); }   // (throws for an unknown id: callers inside try blocks only)
  // typed reads that never throw: the stored value when it has that type, else the schema default, else `fallback`
  double Number(std::int16_t id, double fallback = 0.0) const {
    if (const Value* v = Find(id)) if (const double* d = std::get_if<double>(v)) return *d;
    const auto it = Schema().find(id);
    return it != Schema().end() && it->second.kind == ParameterInfo::kNumber ? it->second.def : fallback;
  }
  bool Has(std::int16_t id) const { return values_.count(id) != 0; }
  const std::map<std::int16_t, Value>& all() const { return values_; }

  std::vector<unsigned char> Write() const {  // reference parameter_state.cc:136-147
    std::vector<unsigned char> out;
    auto put = [&out](const void* p, size_t n) { const auto* b = static_cast<const unsigned char*>(p); out.insert(out.end(), b, b + n); };
    for (const auto& [id, v] : values_) {
      const std::int32_t type = (std::int32_t)v.index();
      put(&id, 2);
      put(&type, 4);
      if (const int* i = std::get_if<int>(&v)) { const std::int32_t x = *i; put(&x, 4); }
      else if (const double* d = std::get_if<double>(&v)) put(d, 8);
      else { const std::string& s = std::get<std::string>(v); const std::int32_t n = (std::int32_t)s.size(); put(&n, 4); put(s.data(), s.size()); }
    }
    return out;
  }
  // reference parameter_state.cc:68-126 (Read) preceded by :128-133 (defaults first)
  ErrorCode ReadOrSetDefault(const unsigned char* p, size_t n) {
    values_.clear();
    SetDefaultValues();
    size_t at = 0;
    auto take = [&](void* dst, size_t k) { if (at + k > n) return false; std::memcpy(dst, p + at, k); at += k; return true; };
    for (;;) {
      std::int16_t id;
      std::int32_t type;
      if (!take(&id, 2) || !take(&type, 4)) return ErrorCode::kFileTooSmall;   // (an EMPTY stream is too small as well, like the reference)
      if (type == 0) { std::int32_t v; if (!take(&v, 4)) return ErrorCode::kFileTooSmall; (void)Set(id, (int)v); }
      else if (type == 1) { double v; if (!take(&v, 8)) return ErrorCode::kFileTooSmall; (void)Set(id, v); }
      else if (type == 2) {
        std::int32_t len;
        if (!take(&len, 4) || len < 0 || at + (size_t)len > n) return ErrorCode::kFileTooSmall;
        (void)Set(id, std::string(reinterpret_cast<const char*>(p + at), (size_t)len));
        at += (size_t)len;
      } else return ErrorCode::kUnknownError;
      if (at == n) return ErrorCode::kSuccess;
    }
  }

 private:
  std::map<std::int16_t, Value> values_;
};

// VST state framing: int32 size + blob (reference src/vst/processor.cc:233-268)
inline std::vector<unsigned char> FrameState(const std::vector<unsigned char>& blob) {
  std::vector<unsigned char> out(4 + blob.size());
  const std::int32_t n = (std::int32_t)blob.size();
  std::memcpy(out.data(), &n, 4);
  std::memcpy(out.data() + 4, blob.data(), blob.size());
  return out;
}
inline bool UnframeState(const unsigned char* p, size_t n, const unsigned char** blob, size_t* blob_n) {
  std::int32_t siz;
  if (n < 4) return false;
  std::memcpy(&siz, p, 4);
  if (siz < 0 || 4 + (size_t)siz > n) return false;
  *blob = p + 4;
  *blob_n = (size_t)siz;
  return true;
}

// voice-morph parameters -> per-speaker weights (reference voice_morph_parameter.cc:24-58 GetVoiceMorphState,
// voice_morph_state.h:50-85 CalculateMarkerWeights / CalculateWeights), float arithmetic as in the reference
inline std::array<float, kMaxNSpeakers> VoiceMorphWeights(const ParameterState& st) {
  using namespace param_id;
  auto num = [&st](std::int16_t id) { return st.Number(id); };
  const float cx = (float)std::clamp(num(kVoiceMorphCursorX), 0.0, 1.0), cy = (float)std::clamp(num(kVoiceMorphCursorY), 0.0, 1.0);
  const float falloff = std::clamp((float)num(kVoiceMorphFalloff), 0.0f, 4.0f);
  const int count = std::clamp((int)std::round(num(kVoiceMorphMarkerCount)), 1, kMaxNVoiceMorphMarkers);
  int voice[kMaxNVoiceMorphMarkers] = {};
  float mx[kMaxNVoiceMorphMarkers] = {}, my[kMaxNVoiceMorphMarkers] = {};
  for (int i = 0; i < count; ++i) {
    voice[i] = std::clamp((int)std::round(num((std::int16_t)(kVoiceMorphMarkerVoiceBase + i))), 0, kMaxNSpeakers - 1);
    mx[i] = (float)std::clamp(num((std::int16_t)(kVoiceMorphMarkerXBase + i)), 0.0, 1.0);
    my[i] = (float)std::clamp(num((std::int16_t)(kVoiceMorphMarkerYBase + i)), 0.0, 1.0);
  }
  std::array<float, kMaxNVoiceMorphMarkers> mw{};
  if (falloff <= 0.0f) {
    for (int i = 0; i < count; ++i) mw[i] = 1.0f / (float)count;
  } else {
    constexpr float kEpsilon = 0.0008f;
    float total = 0.0f;
    for (int i = 0; i < count; ++i) {
      const float dx = cx - mx[i], dy = cy - my[i];
      const float d2 = dx * dx + dy * dy;
      mw[i] = 1.0f / std::pow(d2 + kEpsilon, falloff);
      total += mw[i];
    }
    for (int i = 0; i < count; ++i) mw[i] /= total;
  }
  std::array<float, kMaxNSpeakers> w{};
  for (int i = 0; i < count; ++i) w[voice[i]] += mw[i];
  return w;
}

}  // namespace beatrice_amd
