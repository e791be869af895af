// batch_convert.cc -- a C++ host of the batched C-ABI, no Python anywhere: what a server or an offline converter links.
//
//   batch_convert <model dir> <n streams> <n hops> <raw float32 in> <raw float32 out> [speaker] [k]
//
// in:  [hops][streams][160] float32 @16 kHz        out: [hops][streams][240] float32 @24 kHz
// Loads the model package with the reference's own readers (processor_core_2.cc:302-351 does the same calls), creates one
// BeatriceBatch for all streams, and streams the hops through BeatriceBatch_StreamFrames (host buffers, tick pipeline).
// Built by `make -C beatrice-vst_amd` into examples/batch_convert; exercised by tests/test_gpu_cpp_example.py.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "beatrice_batch.h"
#include "beatricelib/beatrice.h"

static bool read_all(const std::string& path, std::vector<float>* v, size_t n) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  v->resize(n);
  const size_t got = std::fread(v->data(), sizeof(float), n, f);
  std::fclose(f);
  return got == n;
}

int main(int argc, char** argv) {
  if (argc < 6) { std::fprintf(stderr, "usage: %s <model dir> <streams> <hops> <in.f32> <out.f32> [speaker] [k]\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  const int B = std::atoi(argv[2]), hops = std::atoi(argv[3]);
  const int speaker = argc > 6 ? std::atoi(argv[6]) : 0, k = argc > 7 ? std::atoi(argv[7]) : 0;
  if (B < 1 || hops < 1) return 2;

  auto* pe = Beatrice20rc0_CreatePhoneExtractor();
  auto* pt = Beatrice20rc0_CreatePitchEstimator();
  auto* wg = Beatrice20rc0_CreateWaveformGenerator();
  auto* es = Beatrice20rc0_CreateEmbeddingSetter();
  int err = Beatrice20rc0_ReadPhoneExtractorParameters(pe, (dir + "/phone_extractor.bin").c_str());
  err = err ? err : Beatrice20rc0_ReadPitchEstimatorParameters(pt, (dir + "/pitch_estimator.bin").c_str());
  err = err ? err : Beatrice20rc0_ReadWaveformGeneratorParameters(wg, (dir + "/waveform_generator.bin").c_str());
  err = err ? err : Beatrice20rc0_ReadEmbeddingSetterParameters(es, (dir + "/embedding_setter.bin").c_str());
  int n_speakers = 0;
  const std::string spk = dir + "/speaker_embeddings.bin";
  err = err ? err : Beatrice20rc0_ReadNSpeakers(spk.c_str(), &n_speakers);
  if (err) { std::fprintf(stderr, "model package: Beatrice_ErrorCode %d\n", err); return 1; }
  // caller-owned tables with the reference's extra "morph" slot (processor_core_2.cc:335-351)
  const int slots = n_speakers + 1;
  std::vector<float> codebooks((size_t)slots * BEATRICE_20RC0_CODEBOOK_SIZE * BEATRICE_20RC0_PHONE_CHANNELS);
  std::vector<float> additive((size_t)slots * BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS);
  std::vector<float> formant((size_t)9 * BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS);
  std::vector<float> kv((size_t)slots * BEATRICE_20RC0_KV_LENGTH * BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS);
  err = Beatrice20rc0_ReadSpeakerEmbeddings(spk.c_str(), codebooks.data(), additive.data(), formant.data(), kv.data());
  if (err) { std::fprintf(stderr, "speaker table: Beatrice_ErrorCode %d\n", err); return 1; }

  BeatriceBatch* b = BeatriceBatch_Create(pe, pt, wg, es, B, slots);
  if (!b || !BeatriceBatch_IsHealthy(b)) { std::fprintf(stderr, "no usable GPU\n"); return 1; }
  int rc = BeatriceBatch_SetSpeakerTables(b, slots, codebooks.data(), additive.data(), formant.data(), kv.data());
  rc = rc ? rc : BeatriceBatch_SetTargetSpeaker(b, -1, speaker);
  rc = rc ? rc : BeatriceBatch_FlushSpeaker(b, -1);
  rc = rc ? rc : BeatriceBatch_SetVQNumNeighbors(b, -1, k);
  rc = rc ? rc : BeatriceBatch_SetMinSourcePitch(b, -1, 33.125);   // the reference host's defaults (processor_core_2.h:103-113)
  rc = rc ? rc : BeatriceBatch_SetMaxSourcePitch(b, -1, 80.875);
  rc = rc ? rc : BeatriceBatch_EnableHostStreaming(b, 1);
  if (rc) { std::fprintf(stderr, "batch setup: %d\n", rc); return 1; }

  std::vector<float> in, out((size_t)hops * B * 240);
  if (!read_all(argv[4], &in, (size_t)hops * B * 160)) { std::fprintf(stderr, "cannot read %s\n", argv[4]); return 1; }
  size_t done = 0;
  for (int h = 0; h < hops; ++h) {
    const int got = BeatriceBatch_StreamFrames(b, in.data() + (size_t)h * B * 160, out.data() + done * B * 240);
    if (got < 0) { std::fprintf(stderr, "StreamFrames: %d\n", got); return 1; }
    done += (size_t)got;
  }
  for (;;) {
    const int got = BeatriceBatch_StreamFlush(b, out.data() + done * B * 240);
    if (got < 0) { std::fprintf(stderr, "StreamFlush: %d\n", got); return 1; }
    if (got == 0) break;
    done += 1;
  }
  if (done != (size_t)hops) { std::fprintf(stderr, "%zu of %d hops came back\n", done, hops); return 1; }
  FILE* f = std::fopen(argv[5], "wb");
  if (!f || std::fwrite(out.data(), sizeof(float), out.size(), f) != out.size()) { std::fprintf(stderr, "cannot write %s\n", argv[5]); return 1; }
  std::fclose(f);
  std::printf("converted %d hops of %d streams (delay %d steps)\n", hops, B, BeatriceBatch_HostStreamDelay(b));
  BeatriceBatch_EnableHostStreaming(b, 0);
  BeatriceBatch_Destroy(b);
  Beatrice20rc0_DestroyEmbeddingSetter(es);
  Beatrice20rc0_DestroyWaveformGenerator(wg);
  Beatrice20rc0_DestroyPitchEstimator(pt);
  Beatrice20rc0_DestroyPhoneExtractor(pe);
  return 0;
}
