/*
 * beatrice_abi.h -- the drop-in boundary of this project.
 *
 * Declares, with C linkage, every symbol that the reference VST host code binds when it links
 * `beatricelib` (reference lib/beatricelib/beatrice.h).  The reference declares three ABI
 * generations by hand; here one parameterised macro emits each generation, so this file is a
 * specification of the interface, not a copy of the reference header.  Per entry the comment
 * gives the reference line it replaces (all "ref:" line numbers are in
 * /root/reference/lib/beatricelib/beatrice.h).
 *
 * Implementations in this repo:
 *   - beatrice-vst_amd/csrc/   -> libbeatrice_hip.so   (MI355X / gfx950, the product)
 *   - oracle/                  -> libbeatrice_oracle.so (CPU restatement, test infrastructure only)
 *
 * Contract summary:
 *   - Create* take no arguments and never fail visibly; the caller frees with Destroy*.
 *   - Read*Parameters / ReadNSpeakers / ReadSpeakerEmbeddings are the only fallible calls and
 *     return Beatrice_ErrorCode; every other entry returns void and must neither throw nor block
 *     unboundedly -- on an internal (HIP) failure the product writes zeros.
 *   - Model objects are immutable after Read* and may be shared by contexts on different threads;
 *     a context must not be used from two threads at once.
 *   - All embedding tables passed to Set* and Register* are owned by the caller.
 */
#ifndef BEATRICE_ABI_H_
#define BEATRICE_ABI_H_

#ifdef __cplusplus
#define BEATRICE_ABI_BEGIN extern "C" {
#define BEATRICE_ABI_END }
#else
#define BEATRICE_ABI_BEGIN
#define BEATRICE_ABI_END
#endif

/* ---- constants shared by all generations (ref:10-15) ---- */
#define BEATRICE_IN_HOP_LENGTH 160                       /* samples @16 kHz consumed per hop   */
#define BEATRICE_OUT_HOP_LENGTH 240                      /* samples @24 kHz produced per hop   */
#define BEATRICE_PITCH_BINS_PER_OCTAVE 96
#define BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS 256
#define BEATRICE_IN_SAMPLE_RATE 16000
#define BEATRICE_OUT_SAMPLE_RATE 24000
/* ---- per generation (ref:17-28) ---- */
#define BEATRICE_20A2_PHONE_CHANNELS 256
#define BEATRICE_20A2_PITCH_BINS 384
#define BEATRICE_20B1_PHONE_CHANNELS 256
#define BEATRICE_20B1_PITCH_BINS 384
#define BEATRICE_20RC0_PHONE_CHANNELS 128
#define BEATRICE_20RC0_PITCH_BINS 448
#define BEATRICE_20RC0_CODEBOOK_SIZE 512
#define BEATRICE_20RC0_KV_LENGTH 384
#define BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS 128
#define BEATRICE_20RC0_N_BLOCKS 4

/* ---- error codes of the file readers (ref:30-37); values are part of the ABI because the host
 *      casts them straight into its own enum (reference src/common/error.h:11-16) ---- */
typedef enum Beatrice_ErrorCode {
  Beatrice_kSuccess = 0,
  Beatrice_kFileOpenError = 1,
  Beatrice_kFileTooSmall = 2,
  Beatrice_kFileTooLarge = 3,
  Beatrice_kInvalidFileSize = 4
} Beatrice_ErrorCode;

/* Opaque object families common to every generation G (ref:41-53, 124-136, 211-227). */
#define BEATRICE_ABI_OPAQUE(G, Name) \
  struct G##_##Name;                 \
  typedef struct G##_##Name G##_##Name;

/* Create/Destroy pair for model object `Obj` (ref:56-58,71-73,103-105 and their 20b1/20rc0 twins) */
#define BEATRICE_ABI_LIFECYCLE(G, Obj)   \
  G##_##Obj* G##_Create##Obj(void);      \
  void G##_Destroy##Obj(G##_##Obj* obj);

/* The part of the interface whose shape is identical in 20a2, 20b1 and 20rc0. */
#define BEATRICE_ABI_COMMON(G)                                                                   \
  BEATRICE_ABI_OPAQUE(G, PhoneExtractor)                                                         \
  BEATRICE_ABI_OPAQUE(G, PhoneContext1)                                                          \
  BEATRICE_ABI_OPAQUE(G, PitchEstimator)                                                         \
  BEATRICE_ABI_OPAQUE(G, PitchContext1)                                                          \
  BEATRICE_ABI_OPAQUE(G, WaveformGenerator)                                                      \
  BEATRICE_ABI_OPAQUE(G, WaveformContext1)                                                       \
  BEATRICE_ABI_LIFECYCLE(G, PhoneExtractor)                                                      \
  BEATRICE_ABI_LIFECYCLE(G, PhoneContext1)                                                       \
  BEATRICE_ABI_LIFECYCLE(G, PitchEstimator)                                                      \
  BEATRICE_ABI_LIFECYCLE(G, PitchContext1)                                                       \
  BEATRICE_ABI_LIFECYCLE(G, WaveformGenerator)                                                   \
  BEATRICE_ABI_LIFECYCLE(G, WaveformContext1)                                                    \
  /* file -> model object; path is UTF-8 (ref:61-64,76-79,108-111 / 235-238,254-257,297-300) */  \
  Beatrice_ErrorCode G##_ReadPhoneExtractorParameters(G##_PhoneExtractor* m, const char* path);  \
  Beatrice_ErrorCode G##_ReadPitchEstimatorParameters(G##_PitchEstimator* m, const char* path);  \
  Beatrice_ErrorCode G##_ReadWaveformGeneratorParameters(G##_WaveformGenerator* m,               \
                                                         const char* path);                      \
  /* in: 160 samples @16 kHz; out: <gen>_PHONE_CHANNELS floats (ref:65-69 / 243-247) */          \
  void G##_ExtractPhone1(const G##_PhoneExtractor* m, const float* input, float* output,         \
                         G##_PhoneContext1* ctx);                                                \
  /* search range of the estimator, 1 .. <gen>_PITCH_BINS-1 (ref:80-87 / 258-265) */             \
  void G##_SetMinQuantizedPitch(G##_PitchContext1* ctx, int min_quantized_pitch);                \
  void G##_SetMaxQuantizedPitch(G##_PitchContext1* ctx, int max_quantized_pitch);                \
  /* in: 160 samples; out: 1 int bin + 4 float features (ref:88-93 / 266-271) */                 \
  void G##_EstimatePitch1(const G##_PitchEstimator* m, const float* input,                       \
                          int* output_quantized_pitch, float* output_pitch_feature,              \
                          G##_PitchContext1* ctx);                                               \
  /* speaker file: count (ref:95-97 / 273-275) */                                                \
  Beatrice_ErrorCode G##_ReadNSpeakers(const char* path, int* output);

BEATRICE_ABI_BEGIN

/* ===================== 20a2 and 20b1 (legacy generations, ref:39-203) ===================== */
/* Identical shape: the per-call speaker vector (256 floats) is an argument of GenerateWaveform1
 * (ref:112-120, 195-203) and ReadSpeakerEmbeddings fills one n_speakers*256 table (ref:98-101). */
#define BEATRICE_ABI_LEGACY(G)                                                                   \
  BEATRICE_ABI_COMMON(G)                                                                         \
  Beatrice_ErrorCode G##_ReadSpeakerEmbeddings(const char* path, float* output);                 \
  void G##_GenerateWaveform1(const G##_WaveformGenerator* m, const float* input_phone,           \
                             const int* input_quantized_pitch, const float* input_pitch_features,\
                             const float* input_speaker_embedding, float* output,                \
                             G##_WaveformContext1* ctx);

BEATRICE_ABI_LEGACY(Beatrice20a2)
BEATRICE_ABI_LEGACY(Beatrice20b1)

/* ===================== 20rc0 (the generation the hot path implements, ref:205-343) ========= */
BEATRICE_ABI_COMMON(Beatrice20rc0)
BEATRICE_ABI_OPAQUE(Beatrice20rc0, EmbeddingSetter)
BEATRICE_ABI_OPAQUE(Beatrice20rc0, EmbeddingContext)
BEATRICE_ABI_LIFECYCLE(Beatrice20rc0, EmbeddingSetter)  /* ref:309-311 */
BEATRICE_ABI_LIFECYCLE(Beatrice20rc0, EmbeddingContext) /* ref:312-313 */

/* ref:314-317 */
Beatrice_ErrorCode Beatrice20rc0_ReadEmbeddingSetterParameters(Beatrice20rc0_EmbeddingSetter* m,
                                                               const char* path);
/* ref:239-242 -- k = 0 disables the codebook lookup; header range is 0..512, the host clamps to
 * 0..8 (reference src/common/processor_core_2.cc:587). */
void Beatrice20rc0_SetVQNumNeighbors(Beatrice20rc0_PhoneContext1* ctx, int num_neighbors);
/* ref:276-290 -- fills four caller-owned tables:
 *   codebook  [n_speakers][512][128], additive [n_speakers][256], formant [9][256],
 *   key_value [n_speakers][384][128]. */
Beatrice_ErrorCode Beatrice20rc0_ReadSpeakerEmbeddings(const char* path, float* output_codebook,
                                                       float* output_additive_speaker_embedding,
                                                       float* output_formant_shift_embedding,
                                                       float* output_key_value_speaker_embedding);
/* ref:301-307 -- conditioning is whatever the Set* calls below installed in `ctx`. */
void Beatrice20rc0_GenerateWaveform1(const Beatrice20rc0_WaveformGenerator* m,
                                     const float* input_phone, const int* input_quantized_pitch,
                                     const float* input_pitch_features, float* output,
                                     Beatrice20rc0_WaveformContext1* ctx);
/* ref:318-322 -- borrows `codebook` ([512][128]) until the next SetCodebook / Destroy. */
void Beatrice20rc0_SetCodebook(Beatrice20rc0_PhoneContext1* phone_ctx, const float* codebook);
/* ref:323-327 and ref:328-332 -- `embedding` is 256 floats. */
void Beatrice20rc0_SetAdditiveSpeakerEmbedding(const Beatrice20rc0_EmbeddingSetter* m,
                                               const float* embedding,
                                               Beatrice20rc0_EmbeddingContext* embedding_ctx,
                                               Beatrice20rc0_WaveformContext1* waveform_ctx);
void Beatrice20rc0_SetFormantShiftEmbedding(const Beatrice20rc0_EmbeddingSetter* m,
                                            const float* embedding,
                                            Beatrice20rc0_EmbeddingContext* embedding_ctx,
                                            Beatrice20rc0_WaveformContext1* waveform_ctx);
/* ref:333-338 -- copies [384][128]; the caller may overwrite its buffer right after the call. */
void Beatrice20rc0_RegisterKeyValueSpeakerEmbedding(const Beatrice20rc0_EmbeddingSetter* m,
                                                    const float* kv_speaker_embedding,
                                                    Beatrice20rc0_EmbeddingContext* embedding_ctx);
/* ref:339-343 -- installs the registered embedding into block `block` (0..3) of waveform_ctx. */
void Beatrice20rc0_SetKeyValueSpeakerEmbedding(const Beatrice20rc0_EmbeddingSetter* m, int block,
                                               Beatrice20rc0_EmbeddingContext* embedding_ctx,
                                               Beatrice20rc0_WaveformContext1* waveform_ctx);

BEATRICE_ABI_END

#endif /* BEATRICE_ABI_H_ */
