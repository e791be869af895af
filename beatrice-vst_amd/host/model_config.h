// model_config.h -- what a model package's `.toml` says (counterpart of the reference's ModelConfig,
// reference src/common/model_config.h:17-138), read with toml_subset.h.
//
//   [model]             version (string), name, description
//   [voice.<id>]        name, description, average_pitch (float, 0..128)     id = 0, 1, 2, ... contiguous
//   [voice.<id>.portrait]  path, description
//
// Validation and failure classes are the reference's: a missing key or a value of the wrong TOML type is a type
// error, an average pitch outside [0, 128] or not finite / a voice id outside [0, 256) / voice ids that do not start
// at zero or are not contiguous / no voice at all are invalid arguments; all of them surface as
// ErrorCode::kInvalidModelConfig in ProcessorProxy::LoadModel.  NUL characters in display strings become spaces
// (model_config.h:63-69).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <stdexcept>
#include <string>

#include "toml_subset.h"

namespace beatrice_amd {

constexpr int kMaxNSpeakers = 256;  // reference model_config.h:17

struct ModelConfig {
  struct Model {
    std::string version, name, description;
    // reference model_config.h:25-35: which generation of the inference library the package targets
    int VersionInt() const {
      if (version == "2.0.0-alpha.2") return 0;
      if (version == "2.0.0-beta.1") return 1;
      if (version == "2.0.0-rc.0") return 2;
      return -1;
    }
  } model;
  struct Voice {
    struct Portrait { std::string path, description; } portrait;
    std::string name, description;
    double average_pitch = 0.0;
  };
  std::array<Voice, kMaxNSpeakers> voices;
};

inline int GetVoiceCount(const ModelConfig& c) {  // reference model_config.h:50-60
  for (int i = 0; i < kMaxNSpeakers; ++i) {
    const auto& v = c.voices[i];
    if (v.name.empty() && v.description.empty() && v.portrait.path.empty() && v.portrait.description.empty()) return i;
  }
  return kMaxNSpeakers;
}

namespace detail {
inline std::string DisplayText(const toml_subset::Value& t, const char* key) {
  std::string s = t.at(key).as_string();
  std::replace(s.begin(), s.end(), '\0', ' ');
  return s;
}
}  // namespace detail

// throws toml_subset::TypeError, std::invalid_argument, std::out_of_range (the classes the reference's reader throws)
inline ModelConfig ReadModelConfig(const toml_subset::Value& root) {
  using detail::DisplayText;
  ModelConfig c;
  const toml_subset::Table& voices = root.at("voice").as_table();
  for (const auto& [key, v] : voices) {
    size_t used = 0;
    const int id = std::stoi(key, &used);  // std::invalid_argument / std::out_of_range on a non-numeric key, like the reference
    if (id < 0 || id >= kMaxNSpeakers) throw std::out_of_range("speaker id out of range");
    ModelConfig::Voice voice;
    voice.name = DisplayText(v, "name");
    voice.description = DisplayText(v, "description");
    voice.average_pitch = v.at("average_pitch").as_float();
    const toml_subset::Value& p = v.at("portrait");
    voice.portrait.path = p.at("path").as_string();
    voice.portrait.description = DisplayText(p, "description");
    if (!std::isfinite(voice.average_pitch) || voice.average_pitch < 0.0 || voice.average_pitch > 128.0)
      throw std::invalid_argument("average_pitch must be finite and between 0 and 128");
    c.voices[id] = voice;
  }
  const toml_subset::Value& m = root.at("model");
  c.model.version = m.at("version").as_string();
  c.model.name = DisplayText(m, "name");
  c.model.description = DisplayText(m, "description");
  const int n = GetVoiceCount(c);
  if (n == 0 || (size_t)n != voices.size()) throw std::invalid_argument("voice ids must start at zero and be contiguous");
  return c;
}

}  // namespace beatrice_amd
