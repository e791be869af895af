// processor_proxy.h -- load a model package by its `.toml`, keep every externally set parameter, fan parameter ids
// out to the core, save / restore the whole state: this project's counterpart of the reference's ProcessorProxy
// (reference src/common/processor_proxy.{h,cc}).
//
//   * LoadModel(path): parse the TOML (toml_subset.h), read the ModelConfig (model_config.h), pick the core by
//     `model.version` (processor_proxy.h:55-70): "2.0.0-rc.0" builds a ProcessorCore2, "2.0.0-alpha.2" / "2.0.0-beta.1" a
//     ProcessorCoreLegacy on the Beatrice20a2_* / Beatrice20b1_* entry points (processor_core.h), all on the HIP
//     library.  ANY failure leaves the "unloaded" core in place, whose Process() writes zeros (processor_proxy.h:97-99, processor_core.h:95-104).
//   * SetParameter(id, value): store, then apply (SyncParameter, processor_proxy.cc:23-43) through the processor-side
//     rule of the parameter table (parameter_schema.cc: the `ProcessorSetValue` lambdas).
//   * Read / Write: the TLV state blob (parameter_state.h); Read re-applies every parameter, which reloads the
//     model named by kModel (processor_proxy.cc:58-63).
#pragma once
#include <filesystem>
#include <memory>
#include <string>
#include <vector>

#include "model_config.h"
#include "parameter_state.h"
#include "processor_core.h"

namespace beatrice_amd {

class ProcessorProxy {
 public:
  ProcessorProxy();
  double GetSampleRate() const { return sample_rate_; }
  ErrorCode SetSampleRate(double sr);
  ErrorCode SetParameter(std::int16_t id, ParameterState::Value value);
  const ParameterState::Value* FindParameter(std::int16_t id) const { return state_.Find(id); }   // nullptr: unknown id
  ErrorCode LoadModel(const std::filesystem::path& file);
  ErrorCode Read(const unsigned char* blob, size_t n);
  std::vector<unsigned char> Write() const { return state_.Write(); }
  const ParameterState& GetParameterState() const { return state_; }
  // the core's Process (reference ProcessorCoreBase::Process); zeros while no model is loaded
  ErrorCode Process(const float* in, float* out, int n);
  // One audio block as the VST shell hands it to the core (reference src/vst/processor.cc:183-225): channel 0 of the
  // output receives the input, down-mixed (L + R) * 0.5 when there are two input channels; a block whose down-mix is
  // all zeros is NOT converted -- the core is not called, its state and its 10 ms FIFO stand still, the output is that
  // silence (the reference marks this TODO(bug); reproduced as it is) -- otherwise Process runs in place; a second output
  // channel is a copy of the first.  in1 / out1 may be null.  *silent reports which of the two happened.
  ErrorCode ProcessChannels(const float* in0, const float* in1, float* out0, float* out1, int n, bool* silent);
  ErrorCode ResetContext();
  bool IsLoaded() const { return core_ != nullptr; }
  int CoreVersion() const { return core_ ? core_->GetVersion() : -1; }           // reference ProcessorCoreBase::GetVersion; -1 = unloaded
  const ModelConfig* Config() const { return core_ ? &config_ : nullptr; }
  ProcessorCoreBase* core() { return core_.get(); }

 private:
  ErrorCode SyncParameter(std::int16_t id);
  ErrorCode SyncAllParameters(std::int16_t ignore);
  double sample_rate_ = 0.0;
  ParameterState state_;
  std::unique_ptr<ProcessorCoreBase> core_;  // null = the reference's ProcessorCoreUnloaded
  ModelConfig config_;
};

}  // namespace beatrice_amd
