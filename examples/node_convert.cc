// node_convert.cc -- a C++ host for ONE NODE: N GPUs in one process, one host thread + one HIP stream per GPU, no Python.
// (the reference itself runs many instances per process, src/vst/factory.cc:21 kManyInstances.)
//
//   node_convert <model dir> <n gpus> <n streams> <n hops> <in.f32> <out.f32> [speaker] [k] [placement]
//
//   in:  [hops][streams][160] float32 @16 kHz        out: [hops][streams][240] float32 @24 kHz
//   speaker   >= 0: every stream converts to that speaker; -1: stream s converts to speaker s mod n_speakers
//   placement "range" (default): GPU g owns a contiguous range of streams; "speaker": GPU = speaker mod N (speaker-affine:
//             a GPU then only ever touches 1/N of the codebooks and K/V tables; needs speaker = -1 and n_speakers >= N)
//
// Load path: the thread of GPU 0 reads the model package ONCE (the reference's own readers, processor_core_2.cc:302-351);
// the packed parameter blobs and the raw speaker tables then go from GPU 0's memory straight into the other GPUs' -- an
// ncclBroadcast per blob over xGMI, called on librccl directly (BeatriceHip_ModelBlob / BeatriceBatch_SpeakerTablesDevice
// hand out the device pointers; what travels is exactly what the kernels read).  After that the GPUs share nothing: every
// thread streams its own streams' hops through its own batch (BeatriceBatch_StreamFrames: host buffers, tick pipeline), and
// the 8-byte frame counters are summed with one ncclAllReduce at the end.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "beatrice_batch.h"
#include "beatricelib/beatrice.h"

namespace {

bool read_all(const std::string& path, std::vector<float>* v, size_t n) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  v->resize(n);
  const size_t got = std::fread(v->data(), sizeof(float), n, f);
  std::fclose(f);
  return got == n;
}

struct Gpu {  // everything one GPU (one thread) owns
  int device = 0;
  Beatrice20rc0_PhoneExtractor* pe = nullptr;
  Beatrice20rc0_PitchEstimator* pt = nullptr;
  Beatrice20rc0_WaveformGenerator* wg = nullptr;
  Beatrice20rc0_EmbeddingSetter* es = nullptr;
  BeatriceBatch* batch = nullptr;
  hipStream_t stream = nullptr;          // for the load-time collectives
  ncclComm_t comm = nullptr;
  std::vector<int> streams;              // global indices of the streams this GPU converts, ascending
  unsigned long long* d_frames = nullptr;  // device: stream-hops converted (the all-reduced counter)
  int error = 0;
};

#define HIP_OK(x) ((x) == hipSuccess)
#define NCCL_OK(x) ((x) == ncclSuccess)

// one in-place broadcast from GPU 0 for all GPUs (grouped: one call per communicator)
bool broadcast(std::vector<Gpu>& gpus, const std::vector<void*>& ptr, size_t bytes) {
  if (bytes % 4 != 0) return false;
  bool ok = NCCL_OK(ncclGroupStart());
  for (size_t g = 0; g < gpus.size() && ok; ++g) {
    ok = HIP_OK(hipSetDevice(gpus[g].device)) &&
         NCCL_OK(ncclBroadcast(ptr[g], ptr[g], bytes / 4, ncclFloat, /*root=*/0, gpus[g].comm, gpus[g].stream));
  }
  ok = NCCL_OK(ncclGroupEnd()) && ok;
  for (Gpu& g : gpus) ok = HIP_OK(hipSetDevice(g.device)) && HIP_OK(hipStreamSynchronize(g.stream)) && ok;
  return ok;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 7) {
    std::fprintf(stderr, "usage: %s <model dir> <gpus> <streams> <hops> <in.f32> <out.f32> [speaker|-1] [k] [range|speaker]\n", argv[0]);
    return 2;
  }
  const std::string dir = argv[1];
  const int N = std::atoi(argv[2]), B = std::atoi(argv[3]), hops = std::atoi(argv[4]);
  const int speaker = argc > 7 ? std::atoi(argv[7]) : 0, k = argc > 8 ? std::atoi(argv[8]) : 0;
  const bool affine = argc > 9 && std::strcmp(argv[9], "speaker") == 0;
  int n_dev = 0;
  if (!HIP_OK(hipGetDeviceCount(&n_dev)) || N < 1 || N > n_dev || B < N || hops < 1) {
    std::fprintf(stderr, "need 1 <= gpus (%d) <= devices (%d) and at least one stream per GPU\n", N, n_dev);
    return 2;
  }

  // ---- GPU 0 reads the package
  std::vector<Gpu> gpus(N);
  for (int g = 0; g < N; ++g) gpus[g].device = g;
  if (BeatriceHip_SetDevice(0) != 0) return 1;
  Gpu& root = gpus[0];
  root.pe = Beatrice20rc0_CreatePhoneExtractor(); root.pt = Beatrice20rc0_CreatePitchEstimator();
  root.wg = Beatrice20rc0_CreateWaveformGenerator(); root.es = Beatrice20rc0_CreateEmbeddingSetter();
  int err = Beatrice20rc0_ReadPhoneExtractorParameters(root.pe, (dir + "/phone_extractor.bin").c_str());
  err = err ? err : Beatrice20rc0_ReadPitchEstimatorParameters(root.pt, (dir + "/pitch_estimator.bin").c_str());
  err = err ? err : Beatrice20rc0_ReadWaveformGeneratorParameters(root.wg, (dir + "/waveform_generator.bin").c_str());
  err = err ? err : Beatrice20rc0_ReadEmbeddingSetterParameters(root.es, (dir + "/embedding_setter.bin").c_str());
  int n_speakers = 0;
  const std::string spk = dir + "/speaker_embeddings.bin";
  err = err ? err : Beatrice20rc0_ReadNSpeakers(spk.c_str(), &n_speakers);
  if (err) { std::fprintf(stderr, "model package: Beatrice_ErrorCode %d\n", err); return 1; }
  const int slots = n_speakers + 1;  // the reference's extra "morph" slot (processor_core_2.cc:335-351)
  std::vector<float> codebooks((size_t)slots * BEATRICE_20RC0_CODEBOOK_SIZE * BEATRICE_20RC0_PHONE_CHANNELS);
  std::vector<float> additive((size_t)slots * BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS);
  std::vector<float> formant((size_t)9 * BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS);
  std::vector<float> kv((size_t)slots * BEATRICE_20RC0_KV_LENGTH * BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS);
  err = Beatrice20rc0_ReadSpeakerEmbeddings(spk.c_str(), codebooks.data(), additive.data(), formant.data(), kv.data());
  if (err) { std::fprintf(stderr, "speaker table: Beatrice_ErrorCode %d\n", err); return 1; }
  if (affine && (speaker >= 0 || n_speakers < N)) { std::fprintf(stderr, "speaker-affine placement needs speaker = -1 and n_speakers >= gpus\n"); return 2; }

  // ---- who converts what
  auto speaker_of = [&](int s) { return speaker >= 0 ? speaker : s % n_speakers; };
  for (int s = 0; s < B; ++s) {
    const int g = affine ? speaker_of(s) % N : (int)((long long)s * N / B);   // (ranges: floor(s N / B) is monotone, sizes differ by <= 1)
    gpus[g].streams.push_back(s);
  }
  for (const Gpu& g : gpus) if (g.streams.empty()) { std::fprintf(stderr, "GPU %d has no stream\n", g.device); return 2; }

  // ---- the other GPUs: empty model objects, one RCCL communicator per GPU, parameter blobs by broadcast
  std::vector<int> devs(N);
  for (int g = 0; g < N; ++g) devs[g] = g;
  std::vector<ncclComm_t> comms(N);
  if (!NCCL_OK(ncclCommInitAll(comms.data(), N, devs.data()))) { std::fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
  size_t broadcast_bytes = 0;
  for (int g = 0; g < N; ++g) {
    Gpu& G = gpus[g];
    G.comm = comms[g];
    if (!HIP_OK(hipSetDevice(g)) || !HIP_OK(hipStreamCreateWithFlags(&G.stream, hipStreamNonBlocking)) ||
        !HIP_OK(hipMalloc(reinterpret_cast<void**>(&G.d_frames), 8)) || BeatriceHip_SetDevice(g) != 0) { std::fprintf(stderr, "GPU %d setup failed\n", g); return 1; }
    if (g > 0) {
      G.pe = Beatrice20rc0_CreatePhoneExtractor(); G.pt = Beatrice20rc0_CreatePitchEstimator();
      G.wg = Beatrice20rc0_CreateWaveformGenerator(); G.es = Beatrice20rc0_CreateEmbeddingSetter();
    }
  }
  for (int kind = 1; kind <= 4; ++kind) {
    std::vector<void*> ptr(N);
    size_t bytes = 0;
    for (int g = 0; g < N; ++g) {
      Gpu& G = gpus[g];
      void* model = kind == 1 ? (void*)G.pe : kind == 2 ? (void*)G.pt : kind == 3 ? (void*)G.wg : (void*)G.es;
      size_t nb = 0;
      if (BeatriceHip_ModelBlob(kind, model, /*allocate=*/g > 0, &ptr[g], &nb) != 0 || (g > 0 && nb != bytes)) { std::fprintf(stderr, "ModelBlob(%d) on GPU %d failed\n", kind, g); return 1; }
      bytes = nb;
    }
    if (!broadcast(gpus, ptr, bytes)) { std::fprintf(stderr, "ncclBroadcast of parameter blob %d failed\n", kind); return 1; }
    for (int g = 1; g < N; ++g) {
      Gpu& G = gpus[g];
      void* model = kind == 1 ? (void*)G.pe : kind == 2 ? (void*)G.pt : kind == 3 ? (void*)G.wg : (void*)G.es;
      if (BeatriceHip_ModelBlobReady(kind, model) != 0) { std::fprintf(stderr, "ModelBlobReady(%d) on GPU %d failed\n", kind, g); return 1; }
    }
    broadcast_bytes += bytes;
  }

  // ---- batches; speaker tables: uploaded once on GPU 0, broadcast raw, projected on every GPU
  for (Gpu& G : gpus) {
    G.batch = BeatriceBatch_Create(G.pe, G.pt, G.wg, G.es, (int)G.streams.size(), slots);
    if (!G.batch || !BeatriceBatch_IsHealthy(G.batch) || BeatriceBatch_Device(G.batch) != G.device) { std::fprintf(stderr, "batch on GPU %d failed\n", G.device); return 1; }
  }
  if (BeatriceBatch_SetSpeakerTables(root.batch, slots, codebooks.data(), additive.data(), formant.data(), kv.data()) != 0) return 1;
  {
    std::vector<std::vector<void*>> tab(4, std::vector<void*>(N));
    size_t nb[4] = {0, 0, 0, 0};
    for (int g = 0; g < N; ++g) {
      void* p[4];
      size_t b4[4];
      if (BeatriceBatch_SpeakerTablesDevice(gpus[g].batch, p, b4) != 0) { std::fprintf(stderr, "SpeakerTablesDevice on GPU %d failed\n", g); return 1; }
      for (int i = 0; i < 4; ++i) { tab[i][g] = p[i]; nb[i] = b4[i]; }
    }
    for (int i = 0; i < 4; ++i) {
      if (!broadcast(gpus, tab[i], nb[i])) { std::fprintf(stderr, "ncclBroadcast of speaker table %d failed\n", i); return 1; }
      broadcast_bytes += nb[i];
    }
    for (int g = 1; g < N; ++g) if (BeatriceBatch_ProjectSpeakerTables(gpus[g].batch, slots) != 0) { std::fprintf(stderr, "ProjectSpeakerTables on GPU %d failed\n", g); return 1; }
  }

  // ---- input, per-stream settings, conversion: one thread per GPU
  std::vector<float> in, out((size_t)hops * B * 240);
  if (!read_all(argv[5], &in, (size_t)hops * B * 160)) { std::fprintf(stderr, "cannot read %s\n", argv[5]); return 1; }
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> threads;
  for (Gpu& G : gpus) {
    threads.emplace_back([&, gp = &G] {
      Gpu& g = *gp;
      BeatriceBatch* b = g.batch;
      const int n = (int)g.streams.size();
      int rc = 0;
      for (int i = 0; i < n && !rc; ++i) rc = BeatriceBatch_SetTargetSpeaker(b, i, speaker_of(g.streams[i]));
      rc = rc ? rc : BeatriceBatch_FlushSpeaker(b, -1);
      rc = rc ? rc : BeatriceBatch_SetVQNumNeighbors(b, -1, k);
      rc = rc ? rc : BeatriceBatch_SetMinSourcePitch(b, -1, 33.125);   // the reference host's defaults (processor_core_2.h:103-113)
      rc = rc ? rc : BeatriceBatch_SetMaxSourcePitch(b, -1, 80.875);
      rc = rc ? rc : BeatriceBatch_EnableHostStreaming(b, 1);
      if (rc) { g.error = rc; return; }
      std::vector<float> x((size_t)n * 160), y((size_t)n * 240);
      size_t done = 0;
      auto emit = [&] {  // the step that came back: rows back to their streams
        for (int i = 0; i < n; ++i) std::memcpy(&out[(done * B + g.streams[i]) * 240], &y[(size_t)i * 240], 240 * sizeof(float));
        ++done;
      };
      for (int h = 0; h < hops; ++h) {
        for (int i = 0; i < n; ++i) std::memcpy(&x[(size_t)i * 160], &in[((size_t)h * B + g.streams[i]) * 160], 160 * sizeof(float));
        const int got = BeatriceBatch_StreamFrames(b, x.data(), y.data());
        if (got < 0) { g.error = got; return; }
        if (got == 1) emit();
      }
      for (;;) {
        const int got = BeatriceBatch_StreamFlush(b, y.data());
        if (got < 0) { g.error = got; return; }
        if (got == 0) break;
        emit();
      }
      if (done != (size_t)hops) { g.error = -100; return; }
      BeatriceBatch_EnableHostStreaming(b, 0);
      const unsigned long long frames = (unsigned long long)done * n;
      if (!HIP_OK(hipSetDevice(g.device)) || !HIP_OK(hipMemcpy(g.d_frames, &frames, 8, hipMemcpyHostToDevice))) g.error = -101;
    });
  }
  for (std::thread& t : threads) t.join();
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (const Gpu& G : gpus) if (G.error) { std::fprintf(stderr, "GPU %d: error %d\n", G.device, G.error); return 1; }

  // ---- the node's frame counter: one all-reduce of the per-GPU counters
  bool ok = NCCL_OK(ncclGroupStart());
  for (Gpu& G : gpus) ok = ok && HIP_OK(hipSetDevice(G.device)) && NCCL_OK(ncclAllReduce(G.d_frames, G.d_frames, 1, ncclUint64, ncclSum, G.comm, G.stream));
  ok = NCCL_OK(ncclGroupEnd()) && ok;
  unsigned long long total = 0;
  ok = ok && HIP_OK(hipSetDevice(0)) && HIP_OK(hipStreamSynchronize(root.stream)) && HIP_OK(hipMemcpy(&total, root.d_frames, 8, hipMemcpyDeviceToHost));
  if (!ok || total != (unsigned long long)hops * B) { std::fprintf(stderr, "frame counter: %llu, expected %llu\n", total, (unsigned long long)hops * B); return 1; }

  FILE* f = std::fopen(argv[6], "wb");
  if (!f || std::fwrite(out.data(), sizeof(float), out.size(), f) != out.size()) { std::fprintf(stderr, "cannot write %s\n", argv[6]); return 1; }
  std::fclose(f);
  std::printf("{\"gpus\": %d, \"streams\": %d, \"hops\": %d, \"frames\": %llu, \"frames_per_s\": %.1f, \"placement\": \"%s\", \"broadcast_bytes\": %zu, "
              "\"streams_per_gpu\": [", N, B, hops, total, (double)total / seconds, affine ? "speaker" : "range", broadcast_bytes);
  for (int g = 0; g < N; ++g) std::printf("%s%zu", g ? ", " : "", gpus[g].streams.size());
  std::printf("]}\n");

  for (Gpu& G : gpus) {
    BeatriceBatch_Destroy(G.batch);
    Beatrice20rc0_DestroyEmbeddingSetter(G.es); Beatrice20rc0_DestroyWaveformGenerator(G.wg);
    Beatrice20rc0_DestroyPitchEstimator(G.pt); Beatrice20rc0_DestroyPhoneExtractor(G.pe);
    (void)hipSetDevice(G.device);
    (void)hipFree(G.d_frames);
    (void)hipStreamDestroy(G.stream);
    (void)ncclCommDestroy(G.comm);
  }
  return 0;
}
