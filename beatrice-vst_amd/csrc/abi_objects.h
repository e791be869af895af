// abi_objects.h -- definitions of the opaque C-ABI objects (include/beatrice_abi.h) for the HIP
// implementation.  Shared by abi.hip (1-stream ABI) and batch.hip (batched extension).
#pragma once
#include <cstdint>
#include <vector>

#include "beatrice_abi.h"
#include "engine.h"

namespace bhip {
enum : uint32_t { KIND_PHONE = 1, KIND_PITCH = 2, KIND_WAVE = 3, KIND_EMBED = 4, KIND_SPEAKERS = 5 };
Beatrice_ErrorCode parse_model_bytes(const unsigned char* bytes, size_t size, uint32_t kind, long expect_floats,
                                     std::vector<float>* out);
Beatrice_ErrorCode read_model_file(const char* path, uint32_t kind, long expect_floats, std::vector<float>* out);
bool make_stream(hipStream_t* s);
struct CodebookEntry { const float* host; float* d_cbT; float* d_cnorm; };
}  // namespace bhip

// model objects: immutable after Read*Parameters, shareable between contexts and threads
struct Beatrice20rc0_PhoneExtractor { bhip::DeviceBlob blob; bhip::PhoneWeights w{}; bool loaded = false; };
struct Beatrice20rc0_PitchEstimator { bhip::DeviceBlob blob; bhip::PitchWeights w{}; bool loaded = false; };
struct Beatrice20rc0_WaveformGenerator { bhip::DeviceBlob blob; bhip::WaveWeights w{}; bool loaded = false; };
struct Beatrice20rc0_EmbeddingSetter { bhip::DeviceBlob blob; bhip::EmbedWeights w{}; bool loaded = false; };

// per-stream contexts: device state for ONE stream + a private HIP stream + pinned staging
struct Beatrice20rc0_PhoneContext1 {
  bhip::PhoneState st;
  hipStream_t stream = nullptr;
  float* h_io = nullptr;  // pinned: 160 in | step counter | 128 out
  int hop_count = 0;      // hops done; travels to the device with the input copy (no launch spent on counting)
  std::vector<bhip::CodebookEntry> cache;
  bool ok = false;
};
struct Beatrice20rc0_PitchContext1 {
  bhip::PitchState st;
  hipStream_t stream = nullptr;
  float* h_io = nullptr;  // pinned: 160 in | step counter | 4 feat | 1 bin
  int hop_count = 0;
  bool ok = false;
};
struct Beatrice20rc0_WaveformContext1 {
  bhip::WaveState st;
  hipStream_t stream = nullptr;
  float* d_inputs = nullptr;  // device: 128 phone | 4 feat | 1 bin | step counter
  float* h_io = nullptr;      // pinned: inputs | 240 out
  int hop_count = 0;
  bool ok = false;
};
struct Beatrice20rc0_EmbeddingContext {
  hipStream_t stream = nullptr;
  float* d_block = nullptr;
  float *d_kv_raw = nullptr, *d_tmp = nullptr, *d_add = nullptr, *d_frm = nullptr;
  bool ok = false;
};
