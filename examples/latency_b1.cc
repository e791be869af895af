// latency_b1.cc -- BASELINE.json configs[1]: one stream, one speaker, hop-synchronous, through the reference's own three per-hop
// calls -- Beatrice20rc0_ExtractPhone1 / EstimatePitch1 / GenerateWaveform1 exactly as ProcessorCore2::Process1 issues them
// (reference src/common/processor_core_2.cc:184,188,253; a pending speaker switch installs one key/value block per hop, :179-181) --
// timed per hop with clock_gettime(CLOCK_MONOTONIC) from a plain C++ loop: no interpreter, no garbage collector, no ctypes between
// the clock and the calls (bench.py used to time this with perf_counter around ctypes calls and reported one 10.4 ms hop nobody could
// attribute).
//
//   latency_b1 <model dir> [hops = 100000] [warm-up hops = 2000] [speaker = 0] [--histogram] [--in <f32 file of k x 160 samples>]
//              [--period-us P] [--rt]
//   --period-us P : a hop every P microseconds (clock_nanosleep to an absolute deadline), as a host's audio callback arrives -- 10 000 = real
//                   time; the GPU idles between hops and runs each at whatever clocks it then has.  Default 0: hops back to back.
//   --rt          : what a DAW gives its audio thread: SCHED_FIFO (priority 80), memory locked, the thread pinned to the core it runs on.
//                   Only with --period-us (a FIFO thread that never sleeps is throttled by the kernel); reports whether it was granted.
//
// Attribution of slow hops: the loop notes the calling thread's involuntary context switches (getrusage, RUSAGE_THREAD) every 1 000 hops and right after
// every hop above 1 ms: a slow hop that coincides with one was the OS taking the core away from the audio thread, not the library
// (whose waits spin on hipStreamQuery and never sleep, csrc/abi.hip wait_stream).
// Prints ONE JSON object: p50 / p90 / p99 / p99.9 / max in microseconds, the counts of hops above 1 ms and above 10 ms (the hop's
// whole real-time budget), the index of the slowest hop and the ten slowest hops with their positions (a periodic cause shows as a
// pattern), mean and frames per second; and with --histogram the per-call split (phone / pitch / waveform medians).
// Built by `make -C beatrice-vst_amd` into examples/latency_b1; run by bench.py (latency_b1) and tests/test_gpu_cpp_example.py.
#include <dlfcn.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <time.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "beatricelib/beatrice.h"

static inline double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static double pct(const std::vector<double>& sorted, double p) {
  if (sorted.empty()) return 0.0;
  const double at = p * (sorted.size() - 1);
  const size_t lo = (size_t)at;
  const size_t hi = std::min(lo + 1, sorted.size() - 1);
  return sorted[lo] + (sorted[hi] - sorted[lo]) * (at - lo);
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s <model dir> [hops] [warm] [speaker] [--histogram] [--in file.f32]\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  long hops = 100000, warm = 2000;
  int speaker = 0, pos = 0;
  bool split = false, want_rt = false;
  long period_us = 0;
  double gap_us = 0.0;   // --gap-us G: the host idles G microseconds between ExtractPhone1 and EstimatePitch1 (a measurement aid: is the pitch hop done by then?)
  std::string in_path, dump_path;   // --dump <file>: every timed hop's microseconds (and with --histogram the three calls'), one line per hop
  for (int i = 2; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--histogram")) { split = true; continue; }
    if (!std::strcmp(argv[i], "--in") && i + 1 < argc) { in_path = argv[++i]; continue; }
    if (!std::strcmp(argv[i], "--period-us") && i + 1 < argc) { period_us = std::atol(argv[++i]); continue; }
    if (!std::strcmp(argv[i], "--rt")) { want_rt = true; continue; }
    if (!std::strcmp(argv[i], "--gap-us") && i + 1 < argc) { gap_us = std::atof(argv[++i]); continue; }
    if (!std::strcmp(argv[i], "--dump") && i + 1 < argc) { dump_path = argv[++i]; continue; }
    const long v = std::atol(argv[i]);
    if (pos == 0) hops = v; else if (pos == 1) warm = v; else if (pos == 2) speaker = (int)v;
    ++pos;
  }
  if (hops < 1 || warm < 0 || period_us < 0 || (want_rt && period_us == 0)) return 2;

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
  if (speaker < 0 || speaker >= n_speakers) { std::fprintf(stderr, "speaker %d of %d\n", speaker, n_speakers); return 2; }
  const int slots = n_speakers + 1;   // the reference's extra "morph" slot (processor_core_2.cc:335-351)
  const size_t CB = (size_t)BEATRICE_20RC0_CODEBOOK_SIZE * BEATRICE_20RC0_PHONE_CHANNELS, AD = BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS,
               KV = (size_t)BEATRICE_20RC0_KV_LENGTH * BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS;
  std::vector<float> codebooks(slots * CB), additive(slots * AD), formant(9 * AD), kv(slots * KV);
  err = Beatrice20rc0_ReadSpeakerEmbeddings(spk.c_str(), codebooks.data(), additive.data(), formant.data(), kv.data());
  if (err) { std::fprintf(stderr, "speaker table: Beatrice_ErrorCode %d\n", err); return 1; }

  auto* pc = Beatrice20rc0_CreatePhoneContext1();
  auto* tc = Beatrice20rc0_CreatePitchContext1();
  auto* wc = Beatrice20rc0_CreateWaveformContext1();
  auto* ec = Beatrice20rc0_CreateEmbeddingContext();
  // the host's load sequence (processor_core_2.cc:411-414, :431-466, :468-481, :564-588)
  Beatrice20rc0_SetCodebook(pc, codebooks.data() + speaker * CB);
  Beatrice20rc0_SetAdditiveSpeakerEmbedding(es, additive.data() + speaker * AD, ec, wc);
  Beatrice20rc0_RegisterKeyValueSpeakerEmbedding(es, kv.data() + speaker * KV, ec);
  for (int blk = 0; blk < BEATRICE_20RC0_N_BLOCKS; ++blk) Beatrice20rc0_SetKeyValueSpeakerEmbedding(es, blk, ec, wc);
  Beatrice20rc0_SetFormantShiftEmbedding(es, formant.data() + 4 * AD, ec, wc);
  Beatrice20rc0_SetMinQuantizedPitch(tc, 1);
  Beatrice20rc0_SetMaxQuantizedPitch(tc, 383);
  Beatrice20rc0_SetVQNumNeighbors(pc, 0);

  // input: a file of whole hops, or a deterministic voiced-like signal (a 150 Hz saw with a 4 Hz envelope)
  std::vector<float> x;
  if (!in_path.empty()) {
    if (FILE* f = std::fopen(in_path.c_str(), "rb")) {
      std::fseek(f, 0, SEEK_END);
      const long bytes = std::ftell(f);
      std::fseek(f, 0, SEEK_SET);
      x.resize((size_t)bytes / sizeof(float) / BEATRICE_IN_HOP_LENGTH * BEATRICE_IN_HOP_LENGTH);
      if (std::fread(x.data(), sizeof(float), x.size(), f) != x.size()) x.clear();
      std::fclose(f);
    }
    if (x.empty()) { std::fprintf(stderr, "cannot read %s\n", in_path.c_str()); return 1; }
  } else {
    x.resize((size_t)64 * BEATRICE_IN_HOP_LENGTH);
    for (size_t i = 0; i < x.size(); ++i) {
      const double t = i / 16000.0, ph = std::fmod(150.0 * t, 1.0);
      x[i] = (float)(0.3 * (2.0 * ph - 1.0) * (0.55 + 0.45 * std::sin(2.0 * M_PI * 4.0 * t)));
    }
  }
  const size_t n_in = x.size() / BEATRICE_IN_HOP_LENGTH;

  bool rt_granted = false;
  if (want_rt) {
    sched_param sp{};
    sp.sched_priority = 80;
    const int cpu = sched_getcpu();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (cpu >= 0) CPU_SET(cpu, &set);
    rt_granted = sched_setscheduler(0, SCHED_FIFO, &sp) == 0;
    if (cpu >= 0) (void)sched_setaffinity(0, sizeof(set), &set);
    (void)mlockall(MCL_CURRENT | MCL_FUTURE);
  }
  alignas(64) float phone[BEATRICE_20RC0_PHONE_CHANNELS], feat[4], out[BEATRICE_OUT_HOP_LENGTH];
  std::vector<double> lat((size_t)hops), t_phone, t_pitch, t_wave;
  if (split) { t_phone.resize(hops); t_pitch.resize(hops); t_wave.resize(hops); }
  double checksum = 0.0;
  auto switches = []() { rusage ru; getrusage(RUSAGE_THREAD, &ru);   // (the calling thread's own: the runtime's helper threads do not count)
    return std::pair<long, long>(ru.ru_nivcsw, ru.ru_nvcsw); };
  std::pair<long, long> sw_mark = switches();
  std::pair<long, long> sw_start = sw_mark;
  long slow_with_switch = 0, slow_without = 0;
  timespec next{};
  clock_gettime(CLOCK_MONOTONIC, &next);
  for (long i = -warm; i < hops; ++i) {
    if (period_us > 0) {   // the next callback's deadline (absolute: a slow hop does not shift the ones behind it)
      next.tv_nsec += period_us * 1000L;
      while (next.tv_nsec >= 1000000000L) { next.tv_nsec -= 1000000000L; next.tv_sec += 1; }
      (void)clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &next, nullptr);
    }
    const float* in = x.data() + ((size_t)(i + warm) % n_in) * BEATRICE_IN_HOP_LENGTH;
    int q = 0;
    const double t0 = now_us();
    Beatrice20rc0_ExtractPhone1(pe, in, phone, pc);
    double t1 = split ? now_us() : 0.0;
    if (gap_us > 0.0) { const double until = now_us() + gap_us; while (now_us() < until) {} t1 = now_us(); }
    Beatrice20rc0_EstimatePitch1(pt, in, &q, feat, tc);
    const double t2 = split ? now_us() : 0.0;
    q = q < 1 ? 1 : (q > 447 ? 447 : q);   // (identity pitch transform: intonation 1, shift 0, no correction -- :190-252 clamps to [1, 447])
    Beatrice20rc0_GenerateWaveform1(wg, phone, &q, feat, out, wc);
    const double t3 = now_us();
    if (i == 0) sw_start = sw_mark = switches();   // (the timed hops start here)
    if (i >= 0) {
      lat[i] = t3 - t0;
      if (lat[i] > 1000.0 || i % 1000 == 999) {
        const std::pair<long, long> now = switches();
        if (lat[i] > 1000.0) { if (now.first != sw_mark.first) ++slow_with_switch; else ++slow_without; }
        sw_mark = now;
      }
      if (split) { t_phone[i] = t1 - t0; t_pitch[i] = t2 - t1; t_wave[i] = t3 - t2; }
    }
    checksum += out[17];
  }
  double peak = 0.0;
  for (float v : out) peak = std::max(peak, (double)std::fabs(v));
  if (!dump_path.empty()) {
    if (FILE* f = std::fopen(dump_path.c_str(), "w")) {
      for (long i = 0; i < hops; ++i) {
        if (split) std::fprintf(f, "%.1f %.1f %.1f %.1f\n", lat[i], t_phone[i], t_pitch[i], t_wave[i]);
        else std::fprintf(f, "%.1f\n", lat[i]);
      }
      std::fclose(f);
    }
  }

  std::vector<long> order((size_t)hops);
  for (long i = 0; i < hops; ++i) order[i] = i;
  const long n_top = std::min<long>(10, hops);
  std::partial_sort(order.begin(), order.begin() + n_top, order.end(), [&](long a, long b) { return lat[a] > lat[b]; });
  std::vector<double> s = lat;
  std::sort(s.begin(), s.end());
  double sum = 0.0;
  long over1 = 0, over10 = 0;
  for (double v : lat) { sum += v; over1 += v > 1000.0; over10 += v > 10000.0; }
  std::printf("{\"workload\": \"configs[1]: 1 stream through Beatrice20rc0_ExtractPhone1/EstimatePitch1/GenerateWaveform1, %ld hops after %ld warm-up, C loop, clock_gettime\", "
              "\"period_us\": %ld, \"realtime_thread\": %s, \"hops\": %ld, \"p50_us\": %.1f, \"p90_us\": %.1f, \"p99_us\": %.1f, \"p999_us\": %.1f, \"max_us\": %.1f, \"mean_us\": %.1f, "
              "\"hops_over_1ms\": %ld, \"hops_over_10ms\": %ld, \"frames_per_s\": %.1f, \"x_realtime\": %.1f, \"slowest\": [",
              hops, warm, period_us, want_rt ? (rt_granted ? "\"SCHED_FIFO 80, pinned, memory locked\"" : "\"asked for, not granted\"") : "\"no\"", hops, pct(s, 0.50), pct(s, 0.90), pct(s, 0.99), pct(s, 0.999), s.back(), sum / hops, over1, over10, 1e6 / (sum / hops), 1e4 / (sum / hops));
  for (long j = 0; j < n_top; ++j) std::printf("%s{\"hop\": %ld, \"us\": %.1f}", j ? ", " : "", order[j], lat[order[j]]);
  std::printf("]");
  const std::pair<long, long> sw_end = switches();
  std::printf(", \"involuntary_context_switches\": %ld, \"voluntary_context_switches\": %ld, \"hops_over_1ms_with_an_involuntary_switch_nearby\": %ld, "
              "\"hops_over_1ms_without\": %ld", sw_end.first - sw_start.first, sw_end.second - sw_start.second, slow_with_switch, slow_without);
  if (split) {
    std::sort(t_phone.begin(), t_phone.end()); std::sort(t_pitch.begin(), t_pitch.end()); std::sort(t_wave.begin(), t_wave.end());
    std::printf(", \"per_call_p50_us\": {\"ExtractPhone1\": %.1f, \"EstimatePitch1\": %.1f, \"GenerateWaveform1\": %.1f}, "
                "\"per_call_p99_us\": {\"ExtractPhone1\": %.1f, \"EstimatePitch1\": %.1f, \"GenerateWaveform1\": %.1f}",
                pct(t_phone, 0.5), pct(t_pitch, 0.5), pct(t_wave, 0.5), pct(t_phone, 0.99), pct(t_pitch, 0.99), pct(t_wave, 0.99));
  }
  {  // hops of the pitch estimator that ran beside the phone call and were claimed by EstimatePitch1 (include/beatrice_batch.h BeatriceHip_PitchSpeculation;
     // looked up at run time: the oracle build of this tool has no such entry)
    using Fn = int (*)(Beatrice20rc0_PitchContext1*, long long*, long long*);
    long long claimed = 0, dropped = 0;
    if (Fn fn = reinterpret_cast<Fn>(dlsym(RTLD_DEFAULT, "BeatriceHip_PitchSpeculation"))) (void)fn(tc, &claimed, &dropped);
    std::printf(", \"pitch_hops_claimed\": %lld, \"pitch_hops_dropped\": %lld", claimed, dropped);
  }
  std::printf(", \"last_hop_peak\": %.6g, \"checksum\": %.9g}\n", peak, checksum);

  Beatrice20rc0_DestroyPhoneContext1(pc); Beatrice20rc0_DestroyPitchContext1(tc);
  Beatrice20rc0_DestroyWaveformContext1(wc); Beatrice20rc0_DestroyEmbeddingContext(ec);
  Beatrice20rc0_DestroyPhoneExtractor(pe); Beatrice20rc0_DestroyPitchEstimator(pt);
  Beatrice20rc0_DestroyWaveformGenerator(wg); Beatrice20rc0_DestroyEmbeddingSetter(es);
  return peak > 0.0 ? 0 : 3;   // (silence = the context was created without a usable GPU)
}
