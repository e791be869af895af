/* bench_driver.c -- the CPU baseline's all-cores leg as a C loop (test / bench infrastructure, like everything under
 * oracle/): one pthread per stream, each with its own contexts of the ORACLE library, running the reference's per-hop
 * call sequence (ExtractPhone1 -> EstimatePitch1 -> GenerateWaveform1, reference src/common/processor_core_2.cc:181-255)
 * on synthetic audio.  A Python driver loop costs more than the scalar oracle's hop when 32 threads share the
 * interpreter; this measures the library.  Built into liboracle_bench.so (links libbeatrice_oracle.so). */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "beatrice_abi.h"

typedef struct {
  const Beatrice20rc0_PhoneExtractor* phone;
  const Beatrice20rc0_PitchEstimator* pitch;
  const Beatrice20rc0_WaveformGenerator* wave;
  const Beatrice20rc0_EmbeddingSetter* embed;
  const float *codebook, *additive, *formant, *kv;
  int hops, seed;
  double checksum;
} Job;

static void* run(void* arg) {
  Job* j = (Job*)arg;
  Beatrice20rc0_PhoneContext1* pc = Beatrice20rc0_CreatePhoneContext1();
  Beatrice20rc0_PitchContext1* tc = Beatrice20rc0_CreatePitchContext1();
  Beatrice20rc0_WaveformContext1* wc = Beatrice20rc0_CreateWaveformContext1();
  Beatrice20rc0_EmbeddingContext* ec = Beatrice20rc0_CreateEmbeddingContext();
  Beatrice20rc0_SetCodebook(pc, j->codebook);
  Beatrice20rc0_SetAdditiveSpeakerEmbedding(j->embed, j->additive, ec, wc);
  Beatrice20rc0_SetFormantShiftEmbedding(j->embed, j->formant + 4 * 256, ec, wc);
  Beatrice20rc0_RegisterKeyValueSpeakerEmbedding(j->embed, j->kv, ec);
  for (int b = 0; b < BEATRICE_20RC0_N_BLOCKS; ++b) Beatrice20rc0_SetKeyValueSpeakerEmbedding(j->embed, b, ec, wc);
  Beatrice20rc0_SetMinQuantizedPitch(tc, 1);
  Beatrice20rc0_SetMaxQuantizedPitch(tc, 383);
  float in[160], phone[128], feat[4], out[240];
  double sum = 0.0, ph = 0.0;
  unsigned s = 12345u + (unsigned)j->seed;
  for (int h = 0; h < j->hops; ++h) {
    for (int i = 0; i < 160; ++i) {
      s = s * 1664525u + 1013904223u;
      ph += 2.0 * 3.14159265358979 * (140.0 + 40.0 * sin(0.002 * h)) / 16000.0;
      in[i] = (float)(0.3 * sin(ph) + 0.3 * sin(2 * ph) * 0.5 + 0.01 * ((double)(s >> 8) / 8388608.0 - 1.0));
    }
    int q = 0;
    Beatrice20rc0_ExtractPhone1(j->phone, in, phone, pc);
    Beatrice20rc0_EstimatePitch1(j->pitch, in, &q, feat, tc);
    Beatrice20rc0_GenerateWaveform1(j->wave, phone, &q, feat, out, wc);
    sum += out[7];
  }
  j->checksum = sum;
  Beatrice20rc0_DestroyPhoneContext1(pc);
  Beatrice20rc0_DestroyPitchContext1(tc);
  Beatrice20rc0_DestroyWaveformContext1(wc);
  Beatrice20rc0_DestroyEmbeddingContext(ec);
  return NULL;
}

/* returns stream-hops per second over all threads; < 0 on a load error (the Beatrice_ErrorCode, negated) */
double oracle_bench_threads(const char* model_dir, int n_threads, int hops_per_thread) {
  char path[4096];
  Beatrice20rc0_PhoneExtractor* pe = Beatrice20rc0_CreatePhoneExtractor();
  Beatrice20rc0_PitchEstimator* pi = Beatrice20rc0_CreatePitchEstimator();
  Beatrice20rc0_WaveformGenerator* wg = Beatrice20rc0_CreateWaveformGenerator();
  Beatrice20rc0_EmbeddingSetter* es = Beatrice20rc0_CreateEmbeddingSetter();
  int err = 0, n_spk = 0;
  snprintf(path, sizeof path, "%s/phone_extractor.bin", model_dir); if (!err) err = (int)Beatrice20rc0_ReadPhoneExtractorParameters(pe, path);
  snprintf(path, sizeof path, "%s/pitch_estimator.bin", model_dir); if (!err) err = (int)Beatrice20rc0_ReadPitchEstimatorParameters(pi, path);
  snprintf(path, sizeof path, "%s/waveform_generator.bin", model_dir); if (!err) err = (int)Beatrice20rc0_ReadWaveformGeneratorParameters(wg, path);
  snprintf(path, sizeof path, "%s/embedding_setter.bin", model_dir); if (!err) err = (int)Beatrice20rc0_ReadEmbeddingSetterParameters(es, path);
  snprintf(path, sizeof path, "%s/speaker_embeddings.bin", model_dir); if (!err) err = (int)Beatrice20rc0_ReadNSpeakers(path, &n_spk);
  if (err || n_spk < 1) return -(double)(err ? err : 99);
  float* cb = (float*)calloc((size_t)n_spk * 512 * 128, sizeof(float));
  float* add = (float*)calloc((size_t)n_spk * 256, sizeof(float));
  float* frm = (float*)calloc(9 * 256, sizeof(float));
  float* kv = (float*)calloc((size_t)n_spk * 384 * 128, sizeof(float));
  err = (int)Beatrice20rc0_ReadSpeakerEmbeddings(path, cb, add, frm, kv);
  if (err) return -(double)err;
  Job* jobs = (Job*)calloc((size_t)n_threads, sizeof(Job));
  pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < n_threads; ++i) {
    jobs[i] = (Job){pe, pi, wg, es, cb, add, frm, kv, hops_per_thread, i, 0.0};
    pthread_create(&th[i], NULL, run, &jobs[i]);
  }
  for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  free(jobs); free(th); free(cb); free(add); free(frm); free(kv);
  Beatrice20rc0_DestroyPhoneExtractor(pe); Beatrice20rc0_DestroyPitchEstimator(pi);
  Beatrice20rc0_DestroyWaveformGenerator(wg); Beatrice20rc0_DestroyEmbeddingSetter(es);
  return (double)n_threads * hops_per_thread / sec;
}
