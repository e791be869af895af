// legacy.hip -- link-compatible entry points for the 20a2 / 20b1 generations (reference
// lib/beatricelib/beatrice.h:39-203).  The reference host links all three generations
// (reference src/common/processor_core_0.cc, processor_core_1.cc); kernels for the two legacy
// networks are out of scope, so the readers report
// Beatrice_kFileOpenError -- the host's ProcessorProxy then falls back to its "unloaded" core
// (reference src/common/processor_proxy.h:97-99) -- and the per-hop calls emit silence.
#include <cstring>

#include "beatrice_abi.h"

#define BEATRICE_LEGACY_OBJECT(G, Name)                                  \
  struct G##_##Name { int unused; };                                     \
  extern "C" G##_##Name* G##_Create##Name(void) { return new G##_##Name{0}; } \
  extern "C" void G##_Destroy##Name(G##_##Name* o) { delete o; }

#define BEATRICE_LEGACY_GENERATION(G, PHONE_CH)                                                          \
  BEATRICE_LEGACY_OBJECT(G, PhoneExtractor)                                                              \
  BEATRICE_LEGACY_OBJECT(G, PhoneContext1)                                                               \
  BEATRICE_LEGACY_OBJECT(G, PitchEstimator)                                                              \
  BEATRICE_LEGACY_OBJECT(G, PitchContext1)                                                               \
  BEATRICE_LEGACY_OBJECT(G, WaveformGenerator)                                                           \
  BEATRICE_LEGACY_OBJECT(G, WaveformContext1)                                                            \
  extern "C" Beatrice_ErrorCode G##_ReadPhoneExtractorParameters(G##_PhoneExtractor*, const char*) {     \
    return Beatrice_kFileOpenError; }                                                                    \
  extern "C" Beatrice_ErrorCode G##_ReadPitchEstimatorParameters(G##_PitchEstimator*, const char*) {     \
    return Beatrice_kFileOpenError; }                                                                    \
  extern "C" Beatrice_ErrorCode G##_ReadWaveformGeneratorParameters(G##_WaveformGenerator*, const char*) { \
    return Beatrice_kFileOpenError; }                                                                    \
  extern "C" Beatrice_ErrorCode G##_ReadNSpeakers(const char*, int*) { return Beatrice_kFileOpenError; } \
  extern "C" Beatrice_ErrorCode G##_ReadSpeakerEmbeddings(const char*, float*) { return Beatrice_kFileOpenError; } \
  extern "C" void G##_SetMinQuantizedPitch(G##_PitchContext1*, int) {}                                   \
  extern "C" void G##_SetMaxQuantizedPitch(G##_PitchContext1*, int) {}                                   \
  extern "C" void G##_ExtractPhone1(const G##_PhoneExtractor*, const float*, float* out, G##_PhoneContext1*) { \
    std::memset(out, 0, sizeof(float) * (PHONE_CH)); }                                                   \
  extern "C" void G##_EstimatePitch1(const G##_PitchEstimator*, const float*, int* q, float* feat, G##_PitchContext1*) { \
    *q = 1; std::memset(feat, 0, sizeof(float) * 4); }                                                   \
  extern "C" void G##_GenerateWaveform1(const G##_WaveformGenerator*, const float*, const int*, const float*, \
                                        const float*, float* out, G##_WaveformContext1*) {               \
    std::memset(out, 0, sizeof(float) * BEATRICE_OUT_HOP_LENGTH); }

BEATRICE_LEGACY_GENERATION(Beatrice20a2, BEATRICE_20A2_PHONE_CHANNELS)
BEATRICE_LEGACY_GENERATION(Beatrice20b1, BEATRICE_20B1_PHONE_CHANNELS)
