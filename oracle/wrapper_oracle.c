/*
 * oracle/wrapper_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement, in plain C, of the open-source
 * wrapper that surrounds the three per-hop library calls in the reference host:
 *
 *   wo_fraction        <- ComputeSimpleFraction            reference src/common/resample.h:25-46
 *   resampler tables   <- DownUpSamplerImpl::Reset/SetSampleRates          resample.h:209-270
 *   rs_down / rs_up    <- DownUpSamplerImpl::Downsample / Upsample         resample.h:130-206
 *   history windows    <- Buffer                                           resample.h:48-73
 *   stage_frequency    <- ConvertStreamFunctionFrequency::operator()       resample.h:301-318
 *   stage_block480     <- ConvertStreamFunctionBlockSize<480>::operator()  resample.h:343-363
 *   stage_6n           <- ConvertStreamFunctionFrom2In3OutTo6InOut<80>     resample.h:380-394
 *   wo_create cutoffs  <- AnyFreqInOut constructor                         resample.h:412-417
 *   wo_gain_*          <- Gain::Process / Gain::Context                    src/common/gain.h:19-72
 *   wo_pitch_transform <- ProcessorCore2::Process1 pitch math   src/common/processor_core_2.cc:190-252
 *   wo_process         <- ProcessorCore2::Process               src/common/processor_core_2.cc:44-46
 *
 * PINNED: unlike the neural core, these reference sources compile here; oracle/ref_drivers/
 * ref_wrapper.cc builds them unmodified into oracle/_ref/libref_wrapper.so and
 * tests/test_wrapper_oracle.py requires this restatement to match that library bit for bit
 * (resampler, block adapter, gain) over every host rate / block size of SURVEY.md section 8c, and
 * to match the committed fixtures under tests/golden/ that were minted from it.  The pitch
 * transform is the exception: processor_core_2.cc needs the absent toml11 submodule, so it is
 * unbuildable here and its restatement is checked against analytic known answers only.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define WO_FILTER_SIZE 32
#define WO_BLOCK 480
#define WO_PI 3.14159265358979323846

typedef void (*wo_hop_fn)(const float* in160, float* out240, void* user);

/* ---- Stern-Brocot search for numer/denom < 1000 ---------------------------------------------- */
void wo_fraction(double ratio, int* numer, int* denom) {
  int ln = 0, ld = 1, rn = 1, rd = 0;
  for (;;) {
    const int mn = ln + rn, md = ld + rd;
    const int too_big = mn >= 1000 || md >= 1000;
    if (ratio * md < mn) {
      if (too_big) { *numer = ln; *denom = ld; return; }
      rn = mn; rd = md;
    } else {
      if (too_big) { *numer = rn; *denom = rd; return; }
      ln = mn; ld = md;
    }
  }
}

/* ---- sliding window over the most recent `size` samples (zeros before the stream starts) ----- */
typedef struct { float* v; int size, head; } Window;
static void win_init(Window* w, int size) { w->v = (float*)calloc((size_t)size, sizeof(float)); w->size = size; w->head = 0; }
static void win_free(Window* w) { free(w->v); w->v = NULL; }
static void win_push(Window* w, float x) { w->v[w->head] = x; w->head = (w->head + 1) % w->size; }
/* back = 1 is the newest sample */
static float win_back(const Window* w, int back) { return w->v[(w->head - back + 2 * w->size) % w->size]; }

/* ---- rational resampler pair ------------------------------------------------------------------ */
typedef struct {
  int ready, down_first, hi, lo, clock_down, clock_up, ncoef;
  float *coef_down, *coef_up;
  Window high, low;
} Resampler;

static double sinc_norm(double x) { return fabs(x) < 1e-8 ? 1.0 : sin(x * WO_PI) / (x * WO_PI); }

static void rs_init(Resampler* r, double rate_outer, double rate_inner, double cutoff_in, double cutoff_out) {
  memset(r, 0, sizeof(*r));
  if (rate_outer <= 0.0 || rate_inner <= 0.0) return;
  double rate_high, rate_low, cut_down, cut_up;
  r->down_first = rate_outer >= rate_inner;
  if (r->down_first) { rate_high = rate_outer; rate_low = rate_inner; cut_down = cutoff_in; cut_up = cutoff_out; }
  else { rate_high = rate_inner; rate_low = rate_outer; cut_down = cutoff_out; cut_up = cutoff_in; }
  wo_fraction(rate_high / rate_low, &r->hi, &r->lo);
  if (r->hi == 0 || r->lo == 0) return;
  r->ncoef = WO_FILTER_SIZE * r->hi + 1;
  const int center = r->ncoef / 2;
  r->coef_down = (float*)malloc(sizeof(float) * (size_t)r->ncoef);
  r->coef_up = (float*)malloc(sizeof(float) * (size_t)r->ncoef);
  for (int i = 0; i < r->ncoef; ++i) {
    const double pos = (double)(i - center) / (double)r->hi;
    const double hann = 0.5 - 0.5 * cos(WO_PI * 2.0 / (double)(r->ncoef - 1) * (double)i);
    r->coef_down[i] = (float)(cut_down * sinc_norm(pos * cut_down) * hann);
    r->coef_up[i] = (float)(cut_up * sinc_norm(pos * cut_up) * hann);
  }
  r->clock_down = r->clock_up = r->hi - 1;
  win_init(&r->high, WO_FILTER_SIZE * r->hi / r->lo + 1);
  win_init(&r->low, WO_FILTER_SIZE + 1);
  r->ready = 1;
}
static void rs_free(Resampler* r) {
  free(r->coef_down); free(r->coef_up);
  if (r->ready) { win_free(&r->high); win_free(&r->low); }
  r->ready = 0;
}

/* high rate -> low rate; returns the number of samples produced */
static int rs_down(Resampler* r, const float* in, int n, float* out) {
  const float gain = (float)r->lo / (float)r->hi;
  int produced = 0;
  for (int i = 0; i < n; ++i) {
    win_push(&r->high, in[i]);
    r->clock_down += r->lo;
    if (r->clock_down >= r->hi) {
      r->clock_down -= r->hi;
      float acc = 0.0f;
      int back = 1;
      for (int f = r->lo - r->clock_down; f < r->ncoef - 1; f += r->lo) acc += win_back(&r->high, back++) * r->coef_down[f];
      out[produced++] = acc * gain;
    }
  }
  return produced;
}
/* low rate -> high rate; n_out as the reference derives it from the clocks */
static int rs_up_count(const Resampler* r, int n_in) {
  if (r->down_first) return (n_in * r->hi + r->clock_down - r->clock_up) / r->lo;
  return ((n_in + 1) * r->hi - r->clock_up - 1) / r->lo;
}
static int rs_up(Resampler* r, const float* in, int n_in, float* out) {
  const int n_out = rs_up_count(r, n_in);
  int consumed = 0;
  for (int o = 0; o < n_out; ++o) {
    r->clock_up += r->lo;
    if (r->clock_up >= r->hi) { r->clock_up -= r->hi; win_push(&r->low, in[consumed++]); }
    float acc = 0.0f;
    int back = 1;
    for (int f = r->clock_up; f < r->ncoef - 1; f += r->hi) acc += win_back(&r->low, back++) * r->coef_up[f];
    out[o] = acc;
  }
  return n_out;
}

/* ---- gain ramp --------------------------------------------------------------------------------- */
typedef struct { double sample_rate, target_db, current_db; } WoGain;
static double db_to_amp(double db) { return pow(10.0, db * 0.05); }
static double amp_to_db(double amp) { return 20.0 * log10(amp); }
void wo_gain_process(WoGain* g, const float* in, float* out, int n) {
  const double target = db_to_amp(g->target_db);
  double cur = db_to_amp(g->current_db);
  int i = 0;
  if (cur < target) {
    const double ratio = db_to_amp(2.0 / (g->sample_rate * 0.001));
    for (; i < n && cur < target; ++i) { cur = fmin(cur * ratio, target); out[i] = (float)(in[i] * cur); }
  } else if (cur > target) {
    const double ratio = db_to_amp(-2.0 / (g->sample_rate * 0.001));
    for (; i < n && cur > target; ++i) { cur = fmax(cur * ratio, target); out[i] = (float)(in[i] * cur); }
  }
  for (; i < n; ++i) out[i] = (float)(in[i] * cur);
  g->current_db = amp_to_db(cur);
}
WoGain* wo_gain_create(double sample_rate, double db) {
  WoGain* g = (WoGain*)malloc(sizeof(WoGain));
  g->sample_rate = sample_rate; g->target_db = db; g->current_db = db;
  return g;
}
void wo_gain_set_target(WoGain* g, double db) { g->target_db = db; }
void wo_gain_destroy(WoGain* g) { free(g); }

/* ---- pitch transform between EstimatePitch1 and GenerateWaveform1 ------------------------------ */
int wo_pitch_transform(int q, double average_source_pitch, double intonation_intensity, double pitch_shift,
                       double pitch_correction, int pitch_correction_type) {
  const double per = 96.0 / 12.0; /* bins per semitone */
  double t = average_source_pitch + ((double)q - average_source_pitch) * intonation_intensity + per * pitch_shift;
  if (pitch_correction != 0.0) {
    if (pitch_correction_type == 0) {
      const double ref = (floor(t / per) + 0.5) * per; /* midpoint between semitones */
      const double d = (t - ref) * (2.0 / per);
      t = fabs(d) < 1e-4 ? ref : ref + d * pow(fabs(d), -pitch_correction) * (per / 2.0);
    } else if (pitch_correction_type == 1) {
      const double ref = round(t / per) * per; /* nearest semitone */
      const double d = (t - ref) * (2.0 / per);
      if (pitch_correction > 1 - 1e-4) t = ref;
      else if (d >= 0.0) t = ref + pow(d, 1.0 / (1.0 - pitch_correction)) * (per / 2.0);
      else t = ref - pow(-d, 1.0 / (1.0 - pitch_correction)) * (per / 2.0);
    }
  }
  const int r = (int)round(t);
  return r < 1 ? 1 : (r > 447 ? 447 : r);
}

/* ---- the whole Process() chain ----------------------------------------------------------------- */
typedef struct {
  double sample_rate;
  Resampler rs;
  float block[WO_BLOCK];
  int block_fill;
  WoGain gin, gout;
  wo_hop_fn hop;
  void* user;
  float *io, *work;
  int cap;
} WoProcessor;

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

WoProcessor* wo_create(double sample_rate, wo_hop_fn hop, void* user) {
  WoProcessor* p = (WoProcessor*)calloc(1, sizeof(WoProcessor));
  p->sample_rate = sample_rate;
  rs_init(&p->rs, sample_rate, 48000.0, 0.99 * 16000.0 / clampd(sample_rate, 16000.0, 48000.0),
          0.99 * 24000.0 / clampd(sample_rate, 24000.0, 48000.0));
  p->gin.sample_rate = p->gout.sample_rate = sample_rate;
  p->hop = hop;
  p->user = user;
  return p;
}
void wo_destroy(WoProcessor* p) {
  if (!p) return;
  rs_free(&p->rs);
  free(p->io); free(p->work);
  free(p);
}
void wo_set_input_gain(WoProcessor* p, double db) { p->gin.target_db = db; }
void wo_set_output_gain(WoProcessor* p, double db) { p->gout.target_db = db; }
int wo_ratio(const WoProcessor* p, int* hi, int* lo) { *hi = p->rs.hi; *lo = p->rs.lo; return p->rs.ready; }

/* 480 samples @48 kHz -> every 3rd sample (160 @16 kHz) -> model hop -> 240 @24 kHz zero-stuffed */
static void stage_6n(WoProcessor* p, const float* in480, float* out480) {
  float in160[160], out240[240];
  for (int i = 0; i < 160; ++i) in160[i] = in480[(i + 1) * 3 - 1];
  p->hop(in160, out240, p->user);
  memset(out480, 0, sizeof(float) * WO_BLOCK);
  for (int i = 0; i < 240; ++i) out480[i * 2] = out240[i];
}
/* FIFO to exact 480-sample blocks: emits the previous block's result, i.e. +480 samples latency */
static void stage_block480(WoProcessor* p, const float* in, float* out, int n) {
  int done = 0;
  while (done < n) {
    int take = WO_BLOCK - p->block_fill;
    if (take > n - done) take = n - done;
    memcpy(out + done, p->block + p->block_fill, sizeof(float) * (size_t)take);
    memcpy(p->block + p->block_fill, in + done, sizeof(float) * (size_t)take);
    p->block_fill += take;
    done += take;
    if (p->block_fill == WO_BLOCK) {
      float processed[WO_BLOCK];
      p->block_fill = 0;
      stage_6n(p, p->block, processed);
      memcpy(p->block, processed, sizeof(processed));
    }
  }
}
/* host rate -> 48 kHz -> block stage -> host rate */
static void stage_frequency(WoProcessor* p, const float* in, float* out, int m) {
  const int need = (int)((double)m * 48000.0 / (p->sample_rate > 1.0 ? p->sample_rate : 1.0)) + m + 2048;
  if (need > p->cap) {
    p->cap = need;
    p->io = (float*)realloc(p->io, sizeof(float) * (size_t)need);
    p->work = (float*)realloc(p->work, sizeof(float) * (size_t)need);
  }
  int n;
  if (p->rs.down_first) n = rs_down(&p->rs, in, m, p->work);
  else n = rs_up(&p->rs, in, m, p->work);
  stage_block480(p, p->work, p->io, n);
  if (p->rs.down_first) rs_up(&p->rs, p->io, n, p->work);
  else rs_down(&p->rs, p->io, n, p->work);
  memcpy(out, p->work, sizeof(float) * (size_t)m);
}

/* returns 0 on success; on a not-ready chain writes zeros and returns the reference's error code
 * (kResamplerNotReady = 10, kGainNotReady = 11; reference src/common/error.h:11-25) */
int wo_process(WoProcessor* p, const float* in, float* out, int n) {
  if (!p->rs.ready) { memset(out, 0, sizeof(float) * (size_t)n); return 10; }
  if (!(p->gin.sample_rate > 1e-5) || !(p->gout.sample_rate > 1e-5)) { memset(out, 0, sizeof(float) * (size_t)n); return 11; }
  float* tmp = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  wo_gain_process(&p->gin, in, tmp, n);
  stage_frequency(p, tmp, out, n);
  wo_gain_process(&p->gout, out, out, n);
  free(tmp);
  return 0;
}
