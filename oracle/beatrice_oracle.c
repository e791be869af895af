/*
 * oracle/beatrice_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (beatrice-vst_amd/) never links, imports or falls back to it.
 *
 * PARITY UNPINNED for the neural core: the reference's implementation of these entry points
 * (closed-source beatricelib, reference Makefile:24-28, LICENSES_BUNDLED.txt:31-32) is absent
 * from /root/reference, has no Linux build, and the reference holds no test or golden vector for
 * it.  What this file follows is therefore:
 *   - the boundary and call protocol: reference lib/beatricelib/beatrice.h:205-343 and its
 *     callers in reference src/common/processor_core_2.cc:181-255,293-351,431-481,561-590;
 *   - the arithmetic: this repo's frozen MODEL_SPEC.md (sections cited per function below),
 *     which honours every ABI-visible constant of beatrice.h:10-28.
 * The HIP implementation is checked against THIS file (<= 1e-4 max-abs, in practice bit-exact).
 *
 * Style: scalar, single-threaded, obviously-correct loops.  Every dot product is one k-ascending
 * fused-multiply-add chain per 256-long segment of the reduction index, segments added in order
 * (MODEL_SPEC section 2.2), written n-innermost so gcc can vectorise across output channels
 * without changing any per-output rounding.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/beatrice_abi.h"
#include "spec_math.h"

#define IN_HOP BEATRICE_IN_HOP_LENGTH
#define OUT_HOP BEATRICE_OUT_HOP_LENGTH
#define HID BEATRICE_WAVEFORM_GENERATOR_HIDDEN_CHANNELS
#define PHONE_CH BEATRICE_20RC0_PHONE_CHANNELS
#define PITCH_BINS BEATRICE_20RC0_PITCH_BINS
#define CODEBOOK BEATRICE_20RC0_CODEBOOK_SIZE
#define KV_LEN BEATRICE_20RC0_KV_LENGTH
#define KV_CH BEATRICE_20RC0_KV_SPEAKER_EMBEDDING_CHANNELS
#define N_BLOCKS BEATRICE_20RC0_N_BLOCKS
#define FFT_N 1024
#define SPEC_BINS 512
#define PITCH_HIST (FFT_N - IN_HOP)

#define SPEC_SEG 256

#define FILE_MAGIC 0x43525442u
#define FILE_VERSION 1u
enum { KIND_PHONE = 1, KIND_PITCH = 2, KIND_WAVE = 3, KIND_EMBED = 4, KIND_SPEAKERS = 5 };

/* ------------------------------------------------------------------------------------------ */
/* Model files (MODEL_SPEC section 5).  Error convention: reference beatrice.h:30-37.          */
/* ------------------------------------------------------------------------------------------ */
static Beatrice_ErrorCode read_file(const char* path, uint32_t kind, long expect_floats,
                                    float** out, long* out_floats) {
  FILE* f = fopen(path, "rb");
  if (!f) return Beatrice_kFileOpenError;
  fseek(f, 0, SEEK_END);
  const long size = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (size < 16) { fclose(f); return Beatrice_kFileTooSmall; }
  uint32_t hdr[4];
  if (fread(hdr, 4, 4, f) != 4) { fclose(f); return Beatrice_kFileOpenError; }
  if (hdr[0] != FILE_MAGIC || hdr[1] != kind || hdr[2] != FILE_VERSION) {
    fclose(f);
    return Beatrice_kInvalidFileSize;
  }
  const long payload = size - 16;
  if (expect_floats >= 0) {
    if (payload < expect_floats * 4) { fclose(f); return Beatrice_kFileTooSmall; }
    if (payload > expect_floats * 4) { fclose(f); return Beatrice_kFileTooLarge; }
  }
  if (payload % 4 != 0 || (long)hdr[3] * 4 != payload) { fclose(f); return Beatrice_kInvalidFileSize; }
  float* buf = (float*)malloc((size_t)payload > 0 ? (size_t)payload : 4);
  if (fread(buf, 1, (size_t)payload, f) != (size_t)payload) { free(buf); fclose(f); return Beatrice_kFileOpenError; }
  fclose(f);
  *out = buf;
  *out_floats = payload / 4;
  return Beatrice_kSuccess;
}

/* ------------------------------------------------------------------------------------------ */
/* Causal streaming convolution (MODEL_SPEC section 3.1).                                      */
/* y[t][n] = bias[n] + SUM_{j<k, c<cin} x[(t+1)*stride-1-(k-1-j)*dil][c] * w[j*cin+c][n]       */
/* The reduction over the flat index kk = j*cin+c is SEGMENTED (MODEL_SPEC 2.2): every run of   */
/* SPEC_SEG = 256 consecutive kk is one fma chain started at 0; the segment results are added in */
/* ascending order, ((s0 + s1) + s2) + ..., then the bias.  (Defined this way so that parallel   */
/* hardware can evaluate the segments concurrently and still match bit for bit.)                 */
/* `hist` holds the H = (k-1)*dil-(stride-1) frames preceding this hop (zeros at stream start).*/
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int cin, cout, k, stride, dil;
  const float *w, *b;
} Conv;

static int conv_hist_frames(const Conv* c) { return (c->k - 1) * c->dil - (c->stride - 1); }

static void conv_run(const Conv* c, float* hist, const float* xin, int n_in, int pre_lrelu, float* y) {
  const int H = conv_hist_frames(c), cin = c->cin, cout = c->cout;
  const int T = n_in / c->stride;
  float* ext = (float*)malloc(sizeof(float) * (size_t)(H + n_in) * cin);
  memcpy(ext, hist, sizeof(float) * (size_t)H * cin);
  memcpy(ext + (size_t)H * cin, xin, sizeof(float) * (size_t)n_in * cin);
  float* acc = (float*)malloc(sizeof(float) * cout);
  float* sum = (float*)malloc(sizeof(float) * cout);
  for (int t = 0; t < T; ++t) {
    int kk = 0; /* flat reduction index j*cin + ci; a new segment starts every SPEC_SEG indices */
    for (int j = 0; j < c->k; ++j) {
      const int frame = H + (t + 1) * c->stride - 1 - (c->k - 1 - j) * c->dil;
      const float* xr = ext + (size_t)frame * cin;
      for (int ci = 0; ci < cin; ++ci, ++kk) {
        if (kk % SPEC_SEG == 0) {
          if (kk == SPEC_SEG) for (int n = 0; n < cout; ++n) sum[n] = acc[n];
          else if (kk > SPEC_SEG) for (int n = 0; n < cout; ++n) sum[n] = sum[n] + acc[n];
          for (int n = 0; n < cout; ++n) acc[n] = 0.0f;
        }
        float a = xr[ci];
        if (pre_lrelu) a = sp_lrelu(a);
        const float* wr = c->w + ((size_t)j * cin + ci) * cout;
        for (int n = 0; n < cout; ++n) acc[n] = sp_fma(a, wr[n], acc[n]);
      }
    }
    if (kk > SPEC_SEG) for (int n = 0; n < cout; ++n) acc[n] = sum[n] + acc[n];
    for (int n = 0; n < cout; ++n) y[(size_t)t * cout + n] = acc[n] + c->b[n];
  }
  memcpy(hist, ext + (size_t)n_in * cin, sizeof(float) * (size_t)H * cin);
  free(sum);
  free(acc);
  free(ext);
}

/* y[n] = bias[n] + chain_c x[c]*w[c][n] */
static void linear_run(const float* w, const float* b, int cin, int cout, const float* x, float* y) {
  float* sum = (float*)malloc(sizeof(float) * cout);
  for (int ci = 0; ci < cin; ++ci) {
    if (ci % SPEC_SEG == 0) {
      if (ci == SPEC_SEG) for (int n = 0; n < cout; ++n) sum[n] = y[n];
      else if (ci > SPEC_SEG) for (int n = 0; n < cout; ++n) sum[n] = sum[n] + y[n];
      for (int n = 0; n < cout; ++n) y[n] = 0.0f;
    }
    const float a = x[ci];
    const float* wr = w + (size_t)ci * cout;
    for (int n = 0; n < cout; ++n) y[n] = sp_fma(a, wr[n], y[n]);
  }
  if (cin > SPEC_SEG) for (int n = 0; n < cout; ++n) y[n] = sum[n] + y[n];
  if (b) for (int n = 0; n < cout; ++n) y[n] = y[n] + b[n];
  free(sum);
}

/* GRU cell, PyTorch gate order r,z,n (MODEL_SPEC section 3.2).  h is updated in place. */
static void gru_run(const float* wih, const float* whh, const float* bih, const float* bhh,
                    int in_dim, int H, const float* x, float* h) {
  float* gi = (float*)malloc(sizeof(float) * 3 * H);
  float* gh = (float*)malloc(sizeof(float) * 3 * H);
  linear_run(wih, bih, in_dim, 3 * H, x, gi);
  linear_run(whh, bhh, H, 3 * H, h, gh);
  for (int j = 0; j < H; ++j) {
    const float r = sp_sigmoid(gi[j] + gh[j]);
    const float z = sp_sigmoid(gi[H + j] + gh[H + j]);
    const float nn = sp_tanh(sp_fma(r, gh[2 * H + j], gi[2 * H + j]));
    h[j] = sp_fma(z, h[j] - nn, nn);
  }
  free(gi);
  free(gh);
}

/* ------------------------------------------------------------------------------------------ */
/* Phone extractor (MODEL_SPEC section 4.1; boundary: reference beatrice.h:229-247,318-322)     */
/* ------------------------------------------------------------------------------------------ */
struct Beatrice20rc0_PhoneExtractor {
  float* blob;
  Conv f[5], rb[4];
  const float *gru_wih, *gru_whh, *gru_bih, *gru_bhh, *out_w, *out_b;
};
static const int kPhoneF[5][4] = {/*cin,cout,k,stride*/ {1, 64, 10, 5}, {64, 128, 8, 4},
                                  {128, 256, 4, 2}, {256, 256, 4, 2}, {256, 256, 4, 2}};
static long phone_n_floats(void) {
  long n = 0;
  for (int i = 0; i < 5; ++i) n += (long)kPhoneF[i][0] * kPhoneF[i][2] * kPhoneF[i][1] + kPhoneF[i][1];
  n += 4 * (5L * 256 * 256 + 256);
  n += 2 * 256L * 768 + 2 * 768;
  n += 256L * PHONE_CH + PHONE_CH;
  return n;
}
struct Beatrice20rc0_PhoneContext1 {
  float* fh[5];   /* front-end histories */
  float* rbh[4];  /* residual-block histories, 4 frames x 256 */
  float h[256];
  int vq_k;
  const float* codebook; /* borrowed, [512][128] */
  float cnorm[CODEBOOK];
};

Beatrice20rc0_PhoneExtractor* Beatrice20rc0_CreatePhoneExtractor(void) {
  return (Beatrice20rc0_PhoneExtractor*)calloc(1, sizeof(Beatrice20rc0_PhoneExtractor));
}
void Beatrice20rc0_DestroyPhoneExtractor(Beatrice20rc0_PhoneExtractor* m) {
  if (m) { free(m->blob); free(m); }
}
Beatrice_ErrorCode Beatrice20rc0_ReadPhoneExtractorParameters(Beatrice20rc0_PhoneExtractor* m,
                                                              const char* path) {
  float* blob; long n;
  const Beatrice_ErrorCode e = read_file(path, KIND_PHONE, phone_n_floats(), &blob, &n);
  if (e) return e;
  free(m->blob);
  m->blob = blob;
  const float* p = blob;
  for (int i = 0; i < 5; ++i) {
    Conv* c = &m->f[i];
    c->cin = kPhoneF[i][0]; c->cout = kPhoneF[i][1]; c->k = kPhoneF[i][2]; c->stride = kPhoneF[i][3]; c->dil = 1;
    c->w = p; p += (long)c->cin * c->k * c->cout;
    c->b = p; p += c->cout;
  }
  for (int i = 0; i < 4; ++i) {
    Conv* c = &m->rb[i];
    c->cin = 256; c->cout = 256; c->k = 5; c->stride = 1; c->dil = 1;
    c->w = p; p += 5L * 256 * 256;
    c->b = p; p += 256;
  }
  m->gru_wih = p; p += 256L * 768;
  m->gru_whh = p; p += 256L * 768;
  m->gru_bih = p; p += 768;
  m->gru_bhh = p; p += 768;
  m->out_w = p; p += 256L * PHONE_CH;
  m->out_b = p; p += PHONE_CH;
  return Beatrice_kSuccess;
}
Beatrice20rc0_PhoneContext1* Beatrice20rc0_CreatePhoneContext1(void) {
  Beatrice20rc0_PhoneContext1* c = (Beatrice20rc0_PhoneContext1*)calloc(1, sizeof(*c));
  for (int i = 0; i < 5; ++i) {
    const int H = (kPhoneF[i][2] - 1) - (kPhoneF[i][3] - 1);
    c->fh[i] = (float*)calloc((size_t)H * kPhoneF[i][0], sizeof(float));
  }
  for (int i = 0; i < 4; ++i) c->rbh[i] = (float*)calloc(4 * 256, sizeof(float));
  return c;
}
void Beatrice20rc0_DestroyPhoneContext1(Beatrice20rc0_PhoneContext1* c) {
  if (!c) return;
  for (int i = 0; i < 5; ++i) free(c->fh[i]);
  for (int i = 0; i < 4; ++i) free(c->rbh[i]);
  free(c);
}
void Beatrice20rc0_SetVQNumNeighbors(Beatrice20rc0_PhoneContext1* ctx, int k) {
  ctx->vq_k = k < 0 ? 0 : (k > CODEBOOK ? CODEBOOK : k);
}
void Beatrice20rc0_SetCodebook(Beatrice20rc0_PhoneContext1* ctx, const float* codebook) {
  ctx->codebook = codebook;
  for (int j = 0; j < CODEBOOK; ++j) {
    float a = 0.0f;
    for (int c = 0; c < PHONE_CH; ++c) a = sp_fma(codebook[j * PHONE_CH + c], codebook[j * PHONE_CH + c], a);
    ctx->cnorm[j] = a;
  }
}

/* k-nearest-neighbour lookup (MODEL_SPEC section 4.1.3): d_j = |c_j|^2 - 2 x.c_j, k smallest,
 * ties to the lowest index, output = (sum in rank order) / k. */
static void vq_run(const Beatrice20rc0_PhoneContext1* ctx, float* phone) {
  float d[CODEBOOK];
  unsigned char used[CODEBOOK];
  memset(used, 0, sizeof(used));
  for (int j = 0; j < CODEBOOK; ++j) {
    float dot = 0.0f;
    for (int c = 0; c < PHONE_CH; ++c) dot = sp_fma(phone[c], ctx->codebook[j * PHONE_CH + c], dot);
    d[j] = sp_fma(-2.0f, dot, ctx->cnorm[j]);
  }
  float acc[PHONE_CH];
  for (int c = 0; c < PHONE_CH; ++c) acc[c] = 0.0f;
  for (int r = 0; r < ctx->vq_k; ++r) {
    int best = -1;
    for (int j = 0; j < CODEBOOK; ++j)
      if (!used[j] && (best < 0 || d[j] < d[best])) best = j;
    used[best] = 1;
    for (int c = 0; c < PHONE_CH; ++c) acc[c] = acc[c] + ctx->codebook[best * PHONE_CH + c];
  }
  const float kf = (float)ctx->vq_k;
  for (int c = 0; c < PHONE_CH; ++c) phone[c] = acc[c] / kf;
}

void Beatrice20rc0_ExtractPhone1(const Beatrice20rc0_PhoneExtractor* m, const float* input,
                                 float* output, Beatrice20rc0_PhoneContext1* ctx) {
  if (!m->blob) { memset(output, 0, sizeof(float) * PHONE_CH); return; }
  float bufa[32 * 64], bufb[32 * 64];
  const float* cur = input;
  int n_in = IN_HOP;
  float* dst = bufa;
  for (int i = 0; i < 5; ++i) {
    conv_run(&m->f[i], ctx->fh[i], cur, n_in, 0, dst);
    n_in /= m->f[i].stride;
    for (int e = 0; e < n_in * m->f[i].cout; ++e) dst[e] = sp_gelu(dst[e]);
    cur = dst;
    dst = (dst == bufa) ? bufb : bufa;
  }
  float x[256], y[256];
  memcpy(x, cur, sizeof(x));
  for (int i = 0; i < 4; ++i) {
    conv_run(&m->rb[i], ctx->rbh[i], x, 1, 0, y);
    for (int n = 0; n < 256; ++n) x[n] = x[n] + sp_gelu(y[n]);
  }
  gru_run(m->gru_wih, m->gru_whh, m->gru_bih, m->gru_bhh, 256, 256, x, ctx->h);
  linear_run(m->out_w, m->out_b, 256, PHONE_CH, ctx->h, output);
  if (ctx->vq_k > 0 && ctx->codebook) vq_run(ctx, output);
}

/* ------------------------------------------------------------------------------------------ */
/* Pitch estimator (MODEL_SPEC section 4.2; boundary: reference beatrice.h:248-271)             */
/* ------------------------------------------------------------------------------------------ */
struct Beatrice20rc0_PitchEstimator {
  float* blob;
  const float *window, *twiddle;
  Conv p[3];
  const float *gru_wih, *gru_whh, *gru_bih, *gru_bhh, *out_w, *out_b, *voi_w, *voi_b;
};
static long pitch_n_floats(void) {
  long n = FFT_N + FFT_N;
  n += 3L * SPEC_BINS * 128 + 128 + 2 * (3L * 128 * 128 + 128);
  n += 2 * 128L * 384 + 2 * 384;
  n += 128L * PITCH_BINS + PITCH_BINS + 128 + 1;
  return n;
}
struct Beatrice20rc0_PitchContext1 {
  float audio[PITCH_HIST];
  float* ph[3];
  float h[128];
  int min_q, max_q, prev_q;
};
Beatrice20rc0_PitchEstimator* Beatrice20rc0_CreatePitchEstimator(void) {
  return (Beatrice20rc0_PitchEstimator*)calloc(1, sizeof(Beatrice20rc0_PitchEstimator));
}
void Beatrice20rc0_DestroyPitchEstimator(Beatrice20rc0_PitchEstimator* m) {
  if (m) { free(m->blob); free(m); }
}
Beatrice_ErrorCode Beatrice20rc0_ReadPitchEstimatorParameters(Beatrice20rc0_PitchEstimator* m,
                                                              const char* path) {
  float* blob; long n;
  const Beatrice_ErrorCode e = read_file(path, KIND_PITCH, pitch_n_floats(), &blob, &n);
  if (e) return e;
  free(m->blob);
  m->blob = blob;
  const float* p = blob;
  m->window = p; p += FFT_N;
  m->twiddle = p; p += FFT_N;
  for (int i = 0; i < 3; ++i) {
    Conv* c = &m->p[i];
    c->cin = i == 0 ? SPEC_BINS : 128; c->cout = 128; c->k = 3; c->stride = 1; c->dil = 1;
    c->w = p; p += 3L * c->cin * 128;
    c->b = p; p += 128;
  }
  m->gru_wih = p; p += 128L * 384;
  m->gru_whh = p; p += 128L * 384;
  m->gru_bih = p; p += 384;
  m->gru_bhh = p; p += 384;
  m->out_w = p; p += 128L * PITCH_BINS;
  m->out_b = p; p += PITCH_BINS;
  m->voi_w = p; p += 128;
  m->voi_b = p; p += 1;
  return Beatrice_kSuccess;
}
Beatrice20rc0_PitchContext1* Beatrice20rc0_CreatePitchContext1(void) {
  Beatrice20rc0_PitchContext1* c = (Beatrice20rc0_PitchContext1*)calloc(1, sizeof(*c));
  c->ph[0] = (float*)calloc(2 * SPEC_BINS, sizeof(float));
  c->ph[1] = (float*)calloc(2 * 128, sizeof(float));
  c->ph[2] = (float*)calloc(2 * 128, sizeof(float));
  c->min_q = 1;
  c->max_q = PITCH_BINS - 1;
  return c;
}
void Beatrice20rc0_DestroyPitchContext1(Beatrice20rc0_PitchContext1* c) {
  if (!c) return;
  for (int i = 0; i < 3; ++i) free(c->ph[i]);
  free(c);
}
static int clamp_bin(int q) { return q < 1 ? 1 : (q > PITCH_BINS - 1 ? PITCH_BINS - 1 : q); }
void Beatrice20rc0_SetMinQuantizedPitch(Beatrice20rc0_PitchContext1* ctx, int q) { ctx->min_q = clamp_bin(q); }
void Beatrice20rc0_SetMaxQuantizedPitch(Beatrice20rc0_PitchContext1* ctx, int q) { ctx->max_q = clamp_bin(q); }

/* In-place radix-2 decimation-in-time FFT of a real frame (MODEL_SPEC section 4.2.1). */
static void fft1024(const float* tw, float* re, float* im) {
  for (int i = 0; i < FFT_N; ++i) {
    int r = 0;
    for (int b = 0; b < 10; ++b) r |= ((i >> b) & 1) << (9 - b);
    if (r > i) { float t = re[i]; re[i] = re[r]; re[r] = t; t = im[i]; im[i] = im[r]; im[r] = t; }
  }
  for (int half = 1; half < FFT_N; half <<= 1) {
    const int step = FFT_N / (2 * half);
    for (int start = 0; start < FFT_N; start += 2 * half) {
      for (int j = 0; j < half; ++j) {
        const float wr = tw[2 * (j * step)], wi = tw[2 * (j * step) + 1];
        const int a = start + j, b = a + half;
        const float tr = sp_fma(-wi, im[b], wr * re[b]);
        const float ti = sp_fma(wi, re[b], wr * im[b]);
        const float ar = re[a], ai = im[a];
        re[a] = ar + tr; im[a] = ai + ti;
        re[b] = ar - tr; im[b] = ai - ti;
      }
    }
  }
}

void Beatrice20rc0_EstimatePitch1(const Beatrice20rc0_PitchEstimator* m, const float* input,
                                  int* out_q, float* out_feat, Beatrice20rc0_PitchContext1* ctx) {
  if (!m->blob) { *out_q = 1; memset(out_feat, 0, 4 * sizeof(float)); return; }
  static _Thread_local float re[FFT_N], im[FFT_N];
  float frame[FFT_N];
  memcpy(frame, ctx->audio, sizeof(float) * PITCH_HIST);
  memcpy(frame + PITCH_HIST, input, sizeof(float) * IN_HOP);
  memcpy(ctx->audio, frame + IN_HOP, sizeof(float) * PITCH_HIST);
  for (int i = 0; i < FFT_N; ++i) { re[i] = frame[i] * m->window[i]; im[i] = 0.0f; }
  fft1024(m->twiddle, re, im);
  float spec[SPEC_BINS];
  for (int k = 0; k < SPEC_BINS; ++k) {
    const float pw = sp_fma(im[k], im[k], re[k] * re[k]);
    spec[k] = 0.5f * sp_log(pw + 1e-5f);
  }
  float x[128], y[128];
  conv_run(&m->p[0], ctx->ph[0], spec, 1, 0, y);
  for (int n = 0; n < 128; ++n) x[n] = sp_gelu(y[n]);
  for (int i = 1; i < 3; ++i) {
    conv_run(&m->p[i], ctx->ph[i], x, 1, 0, y);
    for (int n = 0; n < 128; ++n) x[n] = x[n] + sp_gelu(y[n]);
  }
  gru_run(m->gru_wih, m->gru_whh, m->gru_bih, m->gru_bhh, 128, 128, x, ctx->h);
  float logit[PITCH_BINS];
  linear_run(m->out_w, m->out_b, 128, PITCH_BINS, ctx->h, logit);
  /* masked argmax, ties to the lowest bin (section 4.2.3) */
  int lo = ctx->min_q, hi = ctx->max_q;
  if (hi < lo) hi = lo;
  int q = lo;
  for (int j = lo + 1; j <= hi; ++j)
    if (logit[j] > logit[q]) q = j;
  /* feature 0: softmax probability of the chosen bin over all 448 logits */
  float mx = logit[0];
  for (int j = 1; j < PITCH_BINS; ++j) mx = logit[j] > mx ? logit[j] : mx;
  float part[64];
  for (int l = 0; l < 64; ++l) {
    float a = 0.0f;
    for (int j = l; j < PITCH_BINS; j += 64) a = a + sp_exp(logit[j] - mx);
    part[l] = a;
  }
  out_feat[0] = sp_exp(logit[q] - mx) / sp_wsum64(part);
  /* feature 1: log energy of the hop */
  for (int l = 0; l < 64; ++l) {
    float a = 0.0f;
    for (int i = l; i < IN_HOP; i += 64) a = sp_fma(input[i], input[i], a);
    part[l] = a;
  }
  out_feat[1] = 0.1f * sp_log(sp_fma(sp_wsum64(part), 1.0f / 160.0f, 1e-8f));
  /* feature 2: clipped bin delta */
  float dq = (float)(q - ctx->prev_q) * 0.125f;
  out_feat[2] = dq < -1.0f ? -1.0f : (dq > 1.0f ? 1.0f : dq);
  ctx->prev_q = q;
  /* feature 3: voicing */
  for (int l = 0; l < 64; ++l) part[l] = sp_fma(ctx->h[l + 64], m->voi_w[l + 64], sp_fma(ctx->h[l], m->voi_w[l], 0.0f));
  out_feat[3] = sp_sigmoid(sp_wsum64(part) + m->voi_b[0]);
  *out_q = q;
}

/* ------------------------------------------------------------------------------------------ */
/* Embedding setter (MODEL_SPEC section 4.3; boundary: reference beatrice.h:308-343)            */
/* ------------------------------------------------------------------------------------------ */
struct Beatrice20rc0_EmbeddingSetter {
  float* blob;
  const float *add_w, *add_b, *frm_w, *frm_b;
  const float *k_w[N_BLOCKS], *k_b[N_BLOCKS], *v_w[N_BLOCKS], *v_b[N_BLOCKS];
};
struct Beatrice20rc0_EmbeddingContext {
  float kv_raw[KV_LEN * KV_CH];
  float add[HID], frm[HID];
};
static long embed_n_floats(void) { return 2 * (HID * (long)HID + HID) + N_BLOCKS * 2 * (KV_CH * (long)HID + HID); }

/* ------------------------------------------------------------------------------------------ */
/* Waveform generator (MODEL_SPEC section 4.4; boundary: reference beatrice.h:291-307)          */
/* ------------------------------------------------------------------------------------------ */
static const int kUpRate[4] = {5, 4, 4, 3};
static const int kUpCh[5] = {256, 128, 64, 32, 16};
static const int kBlockDil[N_BLOCKS] = {1, 2, 4, 8};
struct Beatrice20rc0_WaveformGenerator {
  float* blob;
  const float *inp_w, *inp_b, *pitch_emb, *feat_w;
  Conv c1[N_BLOCKS];
  const float *c2_w[N_BLOCKS], *c2_b[N_BLOCKS], *q_w[N_BLOCKS], *q_b[N_BLOCKS], *o_w[N_BLOCKS], *o_b[N_BLOCKS];
  Conv up[4], ra[4], rb[4], fin;
};
static long wave_n_floats(void) {
  long n = PHONE_CH * (long)HID + HID + PITCH_BINS * (long)HID + 4 * HID;
  n += N_BLOCKS * ((3L * HID * HID + HID) + 3 * (HID * (long)HID + HID));
  for (int s = 0; s < 4; ++s) {
    const long cin = kUpCh[s], cout = kUpCh[s + 1], r = kUpRate[s];
    n += 2 * cin * r * cout + r * cout + 2 * (3 * cout * cout + cout);
  }
  n += 7 * 16 + 1;
  return n;
}
struct Beatrice20rc0_WaveformContext1 {
  float add[HID], frm[HID];
  float* kt[N_BLOCKS]; /* [256][384] keys, transposed */
  float* v[N_BLOCKS];  /* [384][256] */
  float* c1h[N_BLOCKS];
  float* uph[4];
  float* rah[4];
  float* rbh[4];
  float finh[6 * 16];
};

Beatrice20rc0_WaveformGenerator* Beatrice20rc0_CreateWaveformGenerator(void) {
  return (Beatrice20rc0_WaveformGenerator*)calloc(1, sizeof(Beatrice20rc0_WaveformGenerator));
}
void Beatrice20rc0_DestroyWaveformGenerator(Beatrice20rc0_WaveformGenerator* m) {
  if (m) { free(m->blob); free(m); }
}
Beatrice_ErrorCode Beatrice20rc0_ReadWaveformGeneratorParameters(Beatrice20rc0_WaveformGenerator* m,
                                                                 const char* path) {
  float* blob; long n;
  const Beatrice_ErrorCode e = read_file(path, KIND_WAVE, wave_n_floats(), &blob, &n);
  if (e) return e;
  free(m->blob);
  m->blob = blob;
  const float* p = blob;
  m->inp_w = p; p += PHONE_CH * (long)HID;
  m->inp_b = p; p += HID;
  m->pitch_emb = p; p += PITCH_BINS * (long)HID;
  m->feat_w = p; p += 4 * HID;
  for (int b = 0; b < N_BLOCKS; ++b) {
    Conv* c = &m->c1[b];
    c->cin = HID; c->cout = HID; c->k = 3; c->stride = 1; c->dil = kBlockDil[b];
    c->w = p; p += 3L * HID * HID;
    c->b = p; p += HID;
    m->c2_w[b] = p; p += HID * (long)HID; m->c2_b[b] = p; p += HID;
    m->q_w[b] = p; p += HID * (long)HID; m->q_b[b] = p; p += HID;
    m->o_w[b] = p; p += HID * (long)HID; m->o_b[b] = p; p += HID;
  }
  for (int s = 0; s < 4; ++s) {
    const int cin = kUpCh[s], cout = kUpCh[s + 1], r = kUpRate[s];
    Conv* u = &m->up[s];
    u->cin = cin; u->cout = r * cout; u->k = 2; u->stride = 1; u->dil = 1;
    u->w = p; p += 2L * cin * r * cout;
    u->b = p; p += r * cout;
    Conv* a = &m->ra[s];
    a->cin = cout; a->cout = cout; a->k = 3; a->stride = 1; a->dil = 1;
    a->w = p; p += 3L * cout * cout;
    a->b = p; p += cout;
    Conv* bb = &m->rb[s];
    bb->cin = cout; bb->cout = cout; bb->k = 3; bb->stride = 1; bb->dil = 3;
    bb->w = p; p += 3L * cout * cout;
    bb->b = p; p += cout;
  }
  m->fin.cin = 16; m->fin.cout = 1; m->fin.k = 7; m->fin.stride = 1; m->fin.dil = 1;
  m->fin.w = p; p += 7 * 16;
  m->fin.b = p; p += 1;
  return Beatrice_kSuccess;
}
Beatrice20rc0_WaveformContext1* Beatrice20rc0_CreateWaveformContext1(void) {
  Beatrice20rc0_WaveformContext1* c = (Beatrice20rc0_WaveformContext1*)calloc(1, sizeof(*c));
  for (int b = 0; b < N_BLOCKS; ++b) {
    c->kt[b] = (float*)calloc((size_t)HID * KV_LEN, sizeof(float));
    c->v[b] = (float*)calloc((size_t)KV_LEN * HID, sizeof(float));
    c->c1h[b] = (float*)calloc((size_t)2 * kBlockDil[b] * HID, sizeof(float));
  }
  for (int s = 0; s < 4; ++s) {
    c->uph[s] = (float*)calloc((size_t)kUpCh[s], sizeof(float));
    c->rah[s] = (float*)calloc((size_t)2 * kUpCh[s + 1], sizeof(float));
    c->rbh[s] = (float*)calloc((size_t)6 * kUpCh[s + 1], sizeof(float));
  }
  return c;
}
void Beatrice20rc0_DestroyWaveformContext1(Beatrice20rc0_WaveformContext1* c) {
  if (!c) return;
  for (int b = 0; b < N_BLOCKS; ++b) { free(c->kt[b]); free(c->v[b]); free(c->c1h[b]); }
  for (int s = 0; s < 4; ++s) { free(c->uph[s]); free(c->rah[s]); free(c->rbh[s]); }
  free(c);
}

Beatrice20rc0_EmbeddingSetter* Beatrice20rc0_CreateEmbeddingSetter(void) {
  return (Beatrice20rc0_EmbeddingSetter*)calloc(1, sizeof(Beatrice20rc0_EmbeddingSetter));
}
void Beatrice20rc0_DestroyEmbeddingSetter(Beatrice20rc0_EmbeddingSetter* m) {
  if (m) { free(m->blob); free(m); }
}
Beatrice20rc0_EmbeddingContext* Beatrice20rc0_CreateEmbeddingContext(void) {
  return (Beatrice20rc0_EmbeddingContext*)calloc(1, sizeof(Beatrice20rc0_EmbeddingContext));
}
void Beatrice20rc0_DestroyEmbeddingContext(Beatrice20rc0_EmbeddingContext* c) { free(c); }
Beatrice_ErrorCode Beatrice20rc0_ReadEmbeddingSetterParameters(Beatrice20rc0_EmbeddingSetter* m,
                                                               const char* path) {
  float* blob; long n;
  const Beatrice_ErrorCode e = read_file(path, KIND_EMBED, embed_n_floats(), &blob, &n);
  if (e) return e;
  free(m->blob);
  m->blob = blob;
  const float* p = blob;
  m->add_w = p; p += HID * (long)HID; m->add_b = p; p += HID;
  m->frm_w = p; p += HID * (long)HID; m->frm_b = p; p += HID;
  for (int b = 0; b < N_BLOCKS; ++b) {
    m->k_w[b] = p; p += KV_CH * (long)HID; m->k_b[b] = p; p += HID;
    m->v_w[b] = p; p += KV_CH * (long)HID; m->v_b[b] = p; p += HID;
  }
  return Beatrice_kSuccess;
}
void Beatrice20rc0_SetAdditiveSpeakerEmbedding(const Beatrice20rc0_EmbeddingSetter* m,
                                               const float* embedding,
                                               Beatrice20rc0_EmbeddingContext* ec,
                                               Beatrice20rc0_WaveformContext1* wc) {
  if (!m->blob) return;
  linear_run(m->add_w, m->add_b, HID, HID, embedding, ec->add);
  memcpy(wc->add, ec->add, sizeof(ec->add));
}
void Beatrice20rc0_SetFormantShiftEmbedding(const Beatrice20rc0_EmbeddingSetter* m,
                                            const float* embedding,
                                            Beatrice20rc0_EmbeddingContext* ec,
                                            Beatrice20rc0_WaveformContext1* wc) {
  if (!m->blob) return;
  linear_run(m->frm_w, m->frm_b, HID, HID, embedding, ec->frm);
  memcpy(wc->frm, ec->frm, sizeof(ec->frm));
}
void Beatrice20rc0_RegisterKeyValueSpeakerEmbedding(const Beatrice20rc0_EmbeddingSetter* m,
                                                    const float* kv,
                                                    Beatrice20rc0_EmbeddingContext* ec) {
  (void)m;
  memcpy(ec->kv_raw, kv, sizeof(ec->kv_raw));
}
void Beatrice20rc0_SetKeyValueSpeakerEmbedding(const Beatrice20rc0_EmbeddingSetter* m, int block,
                                               Beatrice20rc0_EmbeddingContext* ec,
                                               Beatrice20rc0_WaveformContext1* wc) {
  if (!m->blob || block < 0 || block >= N_BLOCKS) return;
  float row[HID];
  for (int j = 0; j < KV_LEN; ++j) {
    linear_run(m->k_w[block], m->k_b[block], KV_CH, HID, ec->kv_raw + j * KV_CH, row);
    for (int c = 0; c < HID; ++c) wc->kt[block][(size_t)c * KV_LEN + j] = row[c];
    linear_run(m->v_w[block], m->v_b[block], KV_CH, HID, ec->kv_raw + j * KV_CH, wc->v[block] + (size_t)j * HID);
  }
}

void Beatrice20rc0_GenerateWaveform1(const Beatrice20rc0_WaveformGenerator* m, const float* phone,
                                     const int* qp, const float* feat, float* output,
                                     Beatrice20rc0_WaveformContext1* ctx) {
  if (!m->blob) { memset(output, 0, sizeof(float) * OUT_HOP); return; }
  int q = *qp;
  q = q < 0 ? 0 : (q > PITCH_BINS - 1 ? PITCH_BINS - 1 : q);
  float x[HID], t0[HID], t1[HID];
  /* input mix (section 4.4.1) */
  linear_run(m->feat_w, NULL, 4, HID, feat, t0);
  linear_run(m->inp_w, m->inp_b, PHONE_CH, HID, phone, t1);
  for (int n = 0; n < HID; ++n) {
    const float e = (m->pitch_emb[(size_t)q * HID + n] + t0[n]) + (ctx->add[n] + ctx->frm[n]);
    x[n] = t1[n] + e;
  }
  /* conditioned blocks (section 4.4.2) */
  for (int b = 0; b < N_BLOCKS; ++b) {
    float h1[HID], xa[HID], qv[HID], o[HID], sc[KV_LEN];
    conv_run(&m->c1[b], ctx->c1h[b], x, 1, 0, t0);
    for (int n = 0; n < HID; ++n) h1[n] = sp_gelu(t0[n]);
    linear_run(m->c2_w[b], m->c2_b[b], HID, HID, h1, t0);
    for (int n = 0; n < HID; ++n) xa[n] = x[n] + t0[n];
    linear_run(m->q_w[b], m->q_b[b], HID, HID, xa, qv);
    linear_run(ctx->kt[b], NULL, HID, KV_LEN, qv, sc);
    float mx = -INFINITY;
    for (int j = 0; j < KV_LEN; ++j) { sc[j] = sc[j] * 0.0625f; mx = sc[j] > mx ? sc[j] : mx; }
    for (int j = 0; j < KV_LEN; ++j) sc[j] = sp_exp(sc[j] - mx);
    float part[64];
    for (int l = 0; l < 64; ++l) {
      float a = 0.0f;
      for (int j = l; j < KV_LEN; j += 64) a = a + sc[j];
      part[l] = a;
    }
    const float inv = 1.0f / sp_wsum64(part);
    linear_run(ctx->v[b], NULL, KV_LEN, HID, sc, o);
    for (int n = 0; n < HID; ++n) o[n] = o[n] * inv;
    linear_run(m->o_w[b], m->o_b[b], HID, HID, o, t0);
    for (int n = 0; n < HID; ++n) x[n] = xa[n] + t0[n];
  }
  /* upsampler (section 4.4.3): polyphase transposed conv + two dilated residual convs per stage */
  static _Thread_local float ya[OUT_HOP * 16 > 5 * 128 ? OUT_HOP * 16 : 5 * 128], yb[OUT_HOP * 16], yc[OUT_HOP * 16];
  const float* cur = x;
  int T = 1;
  for (int s = 0; s < 4; ++s) {
    const int cout = kUpCh[s + 1];
    conv_run(&m->up[s], ctx->uph[s], cur, T, 1, ya); /* [T][r*cout] == [T*r][cout] */
    T *= kUpRate[s];
    conv_run(&m->ra[s], ctx->rah[s], ya, T, 1, yb);
    for (int e = 0; e < T * cout; ++e) yb[e] = ya[e] + yb[e];
    conv_run(&m->rb[s], ctx->rbh[s], yb, T, 1, yc);
    for (int e = 0; e < T * cout; ++e) yc[e] = yb[e] + yc[e];
    cur = yc;
  }
  float fin[OUT_HOP];
  conv_run(&m->fin, ctx->finh, cur, T, 1, fin);
  for (int i = 0; i < OUT_HOP; ++i) output[i] = sp_tanh(fin[i]);
}

/* ------------------------------------------------------------------------------------------ */
/* Speaker file (boundary: reference beatrice.h:272-290; layout: MODEL_SPEC section 5)          */
/* ------------------------------------------------------------------------------------------ */
#define SPK_FLOATS ((long)CODEBOOK * PHONE_CH + HID + (long)KV_LEN * KV_CH)
static Beatrice_ErrorCode speakers_open(const char* path, float** blob, int* n_speakers) {
  long n;
  const Beatrice_ErrorCode e = read_file(path, KIND_SPEAKERS, -1, blob, &n);
  if (e) return e;
  const long body = n - 9L * HID;
  if (body < SPK_FLOATS) { free(*blob); return Beatrice_kFileTooSmall; }
  if (body % SPK_FLOATS != 0) { free(*blob); return Beatrice_kInvalidFileSize; }
  *n_speakers = (int)(body / SPK_FLOATS);
  return Beatrice_kSuccess;
}
Beatrice_ErrorCode Beatrice20rc0_ReadNSpeakers(const char* path, int* output) {
  float* blob; int n;
  const Beatrice_ErrorCode e = speakers_open(path, &blob, &n);
  if (e) return e;
  free(blob);
  *output = n;
  return Beatrice_kSuccess;
}
Beatrice_ErrorCode Beatrice20rc0_ReadSpeakerEmbeddings(const char* path, float* codebook,
                                                       float* additive, float* formant, float* kv) {
  float* blob; int n;
  const Beatrice_ErrorCode e = speakers_open(path, &blob, &n);
  if (e) return e;
  const float* p = blob;
  memcpy(formant, p, sizeof(float) * 9 * HID); p += 9 * HID;
  for (int s = 0; s < n; ++s) {
    memcpy(codebook + (size_t)s * CODEBOOK * PHONE_CH, p, sizeof(float) * CODEBOOK * PHONE_CH); p += CODEBOOK * PHONE_CH;
    memcpy(additive + (size_t)s * HID, p, sizeof(float) * HID); p += HID;
    memcpy(kv + (size_t)s * KV_LEN * KV_CH, p, sizeof(float) * KV_LEN * KV_CH); p += KV_LEN * KV_CH;
  }
  free(blob);
  return Beatrice_kSuccess;
}

/* ------------------------------------------------------------------------------------------ */
/* Legacy generations 2.0.0-alpha.2 / 2.0.0-beta.1 (boundary: reference beatrice.h:39-203;      */
/* callers: reference src/common/processor_core_0.cc, processor_core_1.cc).  Arithmetic:       */
/* MODEL_SPEC section 6 -- the rc.0 modules with a 256-d phone vector and no codebook, 384      */
/* pitch bins, and a waveform generator conditioned by ONE additive vector handed over per hop  */
/* (no key/value attention).  Both generations share it; PARITY UNPINNED like the rc.0 core.    */
/* ------------------------------------------------------------------------------------------ */
#define L_PHONE_CH BEATRICE_20B1_PHONE_CHANNELS
#define L_PITCH_BINS BEATRICE_20B1_PITCH_BINS
enum { KIND_L_PHONE = 11, KIND_L_PITCH = 12, KIND_L_WAVE = 13, KIND_L_ROWS = 15 };

typedef struct {
  float* blob;
  Conv f[5], rb[4];
  const float *gru_wih, *gru_whh, *gru_bih, *gru_bhh, *out_w, *out_b;
} LPhone;
typedef struct { float* fh[5]; float* rbh[4]; float h[256]; } LPhoneCtx;
static long lphone_n_floats(void) { return phone_n_floats() - (256L * PHONE_CH + PHONE_CH) + (256L * L_PHONE_CH + L_PHONE_CH); }
static Beatrice_ErrorCode lphone_read(LPhone* m, const char* path) {
  float* blob; long n;
  const Beatrice_ErrorCode e = read_file(path, KIND_L_PHONE, lphone_n_floats(), &blob, &n);
  if (e) return e;
  free(m->blob);
  m->blob = blob;
  const float* p = blob;
  for (int i = 0; i < 5; ++i) {
    Conv* c = &m->f[i];
    c->cin = kPhoneF[i][0]; c->cout = kPhoneF[i][1]; c->k = kPhoneF[i][2]; c->stride = kPhoneF[i][3]; c->dil = 1;
    c->w = p; p += (long)c->cin * c->k * c->cout;
    c->b = p; p += c->cout;
  }
  for (int i = 0; i < 4; ++i) {
    Conv* c = &m->rb[i];
    c->cin = 256; c->cout = 256; c->k = 5; c->stride = 1; c->dil = 1;
    c->w = p; p += 5L * 256 * 256;
    c->b = p; p += 256;
  }
  m->gru_wih = p; p += 256L * 768; m->gru_whh = p; p += 256L * 768; m->gru_bih = p; p += 768; m->gru_bhh = p; p += 768;
  m->out_w = p; p += 256L * L_PHONE_CH; m->out_b = p; p += L_PHONE_CH;
  return Beatrice_kSuccess;
}
static void lphone_ctx_init(LPhoneCtx* c) {
  for (int i = 0; i < 5; ++i) c->fh[i] = (float*)calloc((size_t)((kPhoneF[i][2] - 1) - (kPhoneF[i][3] - 1)) * kPhoneF[i][0], sizeof(float));
  for (int i = 0; i < 4; ++i) c->rbh[i] = (float*)calloc(4 * 256, sizeof(float));
}
static void lphone_ctx_free(LPhoneCtx* c) { for (int i = 0; i < 5; ++i) free(c->fh[i]); for (int i = 0; i < 4; ++i) free(c->rbh[i]); }
/* MODEL_SPEC 6.1: section 4.1 steps 1-2 with Linear(256 -> 256); no codebook step */
static void lphone_run(const LPhone* m, const float* input, float* output, LPhoneCtx* ctx) {
  if (!m->blob) { memset(output, 0, sizeof(float) * L_PHONE_CH); return; }
  float bufa[32 * 64], bufb[32 * 64];
  const float* cur = input;
  int n_in = IN_HOP;
  float* dst = bufa;
  for (int i = 0; i < 5; ++i) {
    conv_run(&m->f[i], ctx->fh[i], cur, n_in, 0, dst);
    n_in /= m->f[i].stride;
    for (int e = 0; e < n_in * m->f[i].cout; ++e) dst[e] = sp_gelu(dst[e]);
    cur = dst;
    dst = (dst == bufa) ? bufb : bufa;
  }
  float x[256], y[256];
  memcpy(x, cur, sizeof(x));
  for (int i = 0; i < 4; ++i) {
    conv_run(&m->rb[i], ctx->rbh[i], x, 1, 0, y);
    for (int n = 0; n < 256; ++n) x[n] = x[n] + sp_gelu(y[n]);
  }
  gru_run(m->gru_wih, m->gru_whh, m->gru_bih, m->gru_bhh, 256, 256, x, ctx->h);
  linear_run(m->out_w, m->out_b, 256, L_PHONE_CH, ctx->h, output);
}

typedef struct {
  float* blob;
  const float *window, *twiddle;
  Conv p[3];
  const float *gru_wih, *gru_whh, *gru_bih, *gru_bhh, *out_w, *out_b, *voi_w, *voi_b;
} LPitch;
typedef struct { float audio[PITCH_HIST]; float* ph[3]; float h[128]; int min_q, max_q, prev_q; } LPitchCtx;
static long lpitch_n_floats(void) { return pitch_n_floats() - (128L * PITCH_BINS + PITCH_BINS) + (128L * L_PITCH_BINS + L_PITCH_BINS); }
static Beatrice_ErrorCode lpitch_read(LPitch* m, const char* path) {
  float* blob; long n;
  const Beatrice_ErrorCode e = read_file(path, KIND_L_PITCH, lpitch_n_floats(), &blob, &n);
  if (e) return e;
  free(m->blob);
  m->blob = blob;
  const float* p = blob;
  m->window = p; p += FFT_N;
  m->twiddle = p; p += FFT_N;
  for (int i = 0; i < 3; ++i) {
    Conv* c = &m->p[i];
    c->cin = i == 0 ? SPEC_BINS : 128; c->cout = 128; c->k = 3; c->stride = 1; c->dil = 1;
    c->w = p; p += 3L * c->cin * 128;
    c->b = p; p += 128;
  }
  m->gru_wih = p; p += 128L * 384; m->gru_whh = p; p += 128L * 384; m->gru_bih = p; p += 384; m->gru_bhh = p; p += 384;
  m->out_w = p; p += 128L * L_PITCH_BINS; m->out_b = p; p += L_PITCH_BINS;
  m->voi_w = p; p += 128; m->voi_b = p; p += 1;
  return Beatrice_kSuccess;
}
static int lclamp_bin(int q) { return q < 1 ? 1 : (q > L_PITCH_BINS - 1 ? L_PITCH_BINS - 1 : q); }
/* MODEL_SPEC 6.2: section 4.2 with 384 logits, bins in [1, 383] */
static void lpitch_run(const LPitch* m, const float* input, int* out_q, float* out_feat, LPitchCtx* ctx) {
  if (!m->blob) { *out_q = 1; memset(out_feat, 0, 4 * sizeof(float)); return; }
  static _Thread_local float re[FFT_N], im[FFT_N];
  float frame[FFT_N];
  memcpy(frame, ctx->audio, sizeof(float) * PITCH_HIST);
  memcpy(frame + PITCH_HIST, input, sizeof(float) * IN_HOP);
  memcpy(ctx->audio, frame + IN_HOP, sizeof(float) * PITCH_HIST);
  for (int i = 0; i < FFT_N; ++i) { re[i] = frame[i] * m->window[i]; im[i] = 0.0f; }
  fft1024(m->twiddle, re, im);
  float spec[SPEC_BINS];
  for (int k = 0; k < SPEC_BINS; ++k) spec[k] = 0.5f * sp_log(sp_fma(im[k], im[k], re[k] * re[k]) + 1e-5f);
  float x[128], y[128];
  conv_run(&m->p[0], ctx->ph[0], spec, 1, 0, y);
  for (int n = 0; n < 128; ++n) x[n] = sp_gelu(y[n]);
  for (int i = 1; i < 3; ++i) {
    conv_run(&m->p[i], ctx->ph[i], x, 1, 0, y);
    for (int n = 0; n < 128; ++n) x[n] = x[n] + sp_gelu(y[n]);
  }
  gru_run(m->gru_wih, m->gru_whh, m->gru_bih, m->gru_bhh, 128, 128, x, ctx->h);
  float logit[L_PITCH_BINS];
  linear_run(m->out_w, m->out_b, 128, L_PITCH_BINS, ctx->h, logit);
  int lo = ctx->min_q, hi = ctx->max_q;
  if (hi < lo) hi = lo;
  int q = lo;
  for (int j = lo + 1; j <= hi; ++j) if (logit[j] > logit[q]) q = j;
  float mx = logit[0];
  for (int j = 1; j < L_PITCH_BINS; ++j) mx = logit[j] > mx ? logit[j] : mx;
  float part[64];
  for (int l = 0; l < 64; ++l) {
    float a = 0.0f;
    for (int j = l; j < L_PITCH_BINS; j += 64) a = a + sp_exp(logit[j] - mx);
    part[l] = a;
  }
  out_feat[0] = sp_exp(logit[q] - mx) / sp_wsum64(part);
  for (int l = 0; l < 64; ++l) {
    float a = 0.0f;
    for (int i = l; i < IN_HOP; i += 64) a = sp_fma(input[i], input[i], a);
    part[l] = a;
  }
  out_feat[1] = 0.1f * sp_log(sp_fma(sp_wsum64(part), 1.0f / 160.0f, 1e-8f));
  float dq = (float)(q - ctx->prev_q) * 0.125f;
  out_feat[2] = dq < -1.0f ? -1.0f : (dq > 1.0f ? 1.0f : dq);
  ctx->prev_q = q;
  for (int l = 0; l < 64; ++l) part[l] = sp_fma(ctx->h[l + 64], m->voi_w[l + 64], sp_fma(ctx->h[l], m->voi_w[l], 0.0f));
  out_feat[3] = sp_sigmoid(sp_wsum64(part) + m->voi_b[0]);
  *out_q = q;
}

typedef struct {
  float* blob;
  const float *inp_w, *inp_b, *pitch_emb, *feat_w;
  Conv c1[N_BLOCKS];
  const float *c2_w[N_BLOCKS], *c2_b[N_BLOCKS];
  Conv up[4], ra[4], rb[4], fin;
} LWave;
typedef struct { float* c1h[N_BLOCKS]; float* uph[4]; float* rah[4]; float* rbh[4]; float finh[6 * 16]; } LWaveCtx;
static long lwave_n_floats(void) {
  long n = L_PHONE_CH * (long)HID + HID + L_PITCH_BINS * (long)HID + 4 * HID;
  n += N_BLOCKS * ((3L * HID * HID + HID) + (HID * (long)HID + HID));
  for (int s = 0; s < 4; ++s) {
    const long cin = kUpCh[s], cout = kUpCh[s + 1], r = kUpRate[s];
    n += 2 * cin * r * cout + r * cout + 2 * (3 * cout * cout + cout);
  }
  return n + 7 * 16 + 1;
}
static Beatrice_ErrorCode lwave_read(LWave* m, const char* path) {
  float* blob; long n;
  const Beatrice_ErrorCode e = read_file(path, KIND_L_WAVE, lwave_n_floats(), &blob, &n);
  if (e) return e;
  free(m->blob);
  m->blob = blob;
  const float* p = blob;
  m->inp_w = p; p += L_PHONE_CH * (long)HID; m->inp_b = p; p += HID;
  m->pitch_emb = p; p += L_PITCH_BINS * (long)HID;
  m->feat_w = p; p += 4 * HID;
  for (int b = 0; b < N_BLOCKS; ++b) {
    Conv* c = &m->c1[b];
    c->cin = HID; c->cout = HID; c->k = 3; c->stride = 1; c->dil = kBlockDil[b];
    c->w = p; p += 3L * HID * HID; c->b = p; p += HID;
    m->c2_w[b] = p; p += HID * (long)HID; m->c2_b[b] = p; p += HID;
  }
  for (int s = 0; s < 4; ++s) {
    const int cin = kUpCh[s], cout = kUpCh[s + 1], r = kUpRate[s];
    Conv* u = &m->up[s];
    u->cin = cin; u->cout = r * cout; u->k = 2; u->stride = 1; u->dil = 1;
    u->w = p; p += 2L * cin * r * cout; u->b = p; p += r * cout;
    Conv* a = &m->ra[s];
    a->cin = cout; a->cout = cout; a->k = 3; a->stride = 1; a->dil = 1;
    a->w = p; p += 3L * cout * cout; a->b = p; p += cout;
    Conv* bb = &m->rb[s];
    bb->cin = cout; bb->cout = cout; bb->k = 3; bb->stride = 1; bb->dil = 3;
    bb->w = p; p += 3L * cout * cout; bb->b = p; p += cout;
  }
  m->fin.cin = 16; m->fin.cout = 1; m->fin.k = 7; m->fin.stride = 1; m->fin.dil = 1;
  m->fin.w = p; p += 7 * 16; m->fin.b = p; p += 1;
  return Beatrice_kSuccess;
}
static void lwave_ctx_init(LWaveCtx* c) {
  for (int b = 0; b < N_BLOCKS; ++b) c->c1h[b] = (float*)calloc((size_t)2 * kBlockDil[b] * HID, sizeof(float));
  for (int s = 0; s < 4; ++s) {
    c->uph[s] = (float*)calloc((size_t)kUpCh[s], sizeof(float));
    c->rah[s] = (float*)calloc((size_t)2 * kUpCh[s + 1], sizeof(float));
    c->rbh[s] = (float*)calloc((size_t)6 * kUpCh[s + 1], sizeof(float));
  }
}
static void lwave_ctx_free(LWaveCtx* c) {
  for (int b = 0; b < N_BLOCKS; ++b) free(c->c1h[b]);
  for (int s = 0; s < 4; ++s) { free(c->uph[s]); free(c->rah[s]); free(c->rbh[s]); }
}
/* MODEL_SPEC 6.3: e = (PitchEmb[bin] + W_f.feat) + s;  x = Linear(256 -> 256)(phone) + e;  four blocks
 * x <- x + Linear(gelu(Conv(k3, dil d_b)(x)));  the upsampler of section 4.4.3 */
static void lwave_run(const LWave* m, const float* phone, const int* qp, const float* feat, const float* spk, float* output, LWaveCtx* ctx) {
  if (!m->blob) { memset(output, 0, sizeof(float) * OUT_HOP); return; }
  int q = *qp;
  q = q < 0 ? 0 : (q > L_PITCH_BINS - 1 ? L_PITCH_BINS - 1 : q);
  float x[HID], t0[HID], t1[HID];
  linear_run(m->feat_w, NULL, 4, HID, feat, t0);
  linear_run(m->inp_w, m->inp_b, L_PHONE_CH, HID, phone, t1);
  for (int n = 0; n < HID; ++n) x[n] = t1[n] + ((m->pitch_emb[(size_t)q * HID + n] + t0[n]) + spk[n]);
  for (int b = 0; b < N_BLOCKS; ++b) {
    float h1[HID];
    conv_run(&m->c1[b], ctx->c1h[b], x, 1, 0, t0);
    for (int n = 0; n < HID; ++n) h1[n] = sp_gelu(t0[n]);
    linear_run(m->c2_w[b], m->c2_b[b], HID, HID, h1, t0);
    for (int n = 0; n < HID; ++n) x[n] = x[n] + t0[n];
  }
  static _Thread_local float ya[OUT_HOP * 16 > 5 * 128 ? OUT_HOP * 16 : 5 * 128], yb[OUT_HOP * 16], yc[OUT_HOP * 16];
  const float* cur = x;
  int T = 1;
  for (int s = 0; s < 4; ++s) {
    const int cout = kUpCh[s + 1];
    conv_run(&m->up[s], ctx->uph[s], cur, T, 1, ya);
    T *= kUpRate[s];
    conv_run(&m->ra[s], ctx->rah[s], ya, T, 1, yb);
    for (int e = 0; e < T * cout; ++e) yb[e] = ya[e] + yb[e];
    conv_run(&m->rb[s], ctx->rbh[s], yb, T, 1, yc);
    for (int e = 0; e < T * cout; ++e) yc[e] = yb[e] + yc[e];
    cur = yc;
  }
  float fin[OUT_HOP];
  conv_run(&m->fin, ctx->finh, cur, T, 1, fin);
  for (int i = 0; i < OUT_HOP; ++i) output[i] = sp_tanh(fin[i]);
}
/* embedding rows: [n][256]; the same reader serves speaker_embeddings.bin and formant_shift_embeddings.bin
 * (reference processor_core_1.cc:189-216) */
static Beatrice_ErrorCode lrows_open(const char* path, float** blob, int* rows) {
  long n;
  const Beatrice_ErrorCode e = read_file(path, KIND_L_ROWS, -1, blob, &n);
  if (e) return e;
  if (n < HID) { free(*blob); return Beatrice_kFileTooSmall; }
  if (n % HID != 0) { free(*blob); return Beatrice_kInvalidFileSize; }
  *rows = (int)(n / HID);
  return Beatrice_kSuccess;
}

#define LEGACY_GENERATION(G)                                                                                     \
  struct G##_PhoneExtractor { LPhone m; }; struct G##_PhoneContext1 { LPhoneCtx c; };                            \
  struct G##_PitchEstimator { LPitch m; }; struct G##_PitchContext1 { LPitchCtx c; };                            \
  struct G##_WaveformGenerator { LWave m; }; struct G##_WaveformContext1 { LWaveCtx c; };                        \
  G##_PhoneExtractor* G##_CreatePhoneExtractor(void) { return (G##_PhoneExtractor*)calloc(1, sizeof(G##_PhoneExtractor)); } \
  void G##_DestroyPhoneExtractor(G##_PhoneExtractor* o) { if (o) { free(o->m.blob); free(o); } }                \
  G##_PhoneContext1* G##_CreatePhoneContext1(void) { G##_PhoneContext1* o = (G##_PhoneContext1*)calloc(1, sizeof(*o)); lphone_ctx_init(&o->c); return o; } \
  void G##_DestroyPhoneContext1(G##_PhoneContext1* o) { if (o) { lphone_ctx_free(&o->c); free(o); } }           \
  G##_PitchEstimator* G##_CreatePitchEstimator(void) { return (G##_PitchEstimator*)calloc(1, sizeof(G##_PitchEstimator)); } \
  void G##_DestroyPitchEstimator(G##_PitchEstimator* o) { if (o) { free(o->m.blob); free(o); } }                \
  G##_PitchContext1* G##_CreatePitchContext1(void) {                                                             \
    G##_PitchContext1* o = (G##_PitchContext1*)calloc(1, sizeof(*o));                                            \
    o->c.ph[0] = (float*)calloc(2 * SPEC_BINS, sizeof(float)); o->c.ph[1] = (float*)calloc(2 * 128, sizeof(float)); \
    o->c.ph[2] = (float*)calloc(2 * 128, sizeof(float)); o->c.min_q = 1; o->c.max_q = L_PITCH_BINS - 1;          \
    return o;                                                                                                    \
  }                                                                                                              \
  void G##_DestroyPitchContext1(G##_PitchContext1* o) { if (o) { for (int i = 0; i < 3; ++i) free(o->c.ph[i]); free(o); } } \
  G##_WaveformGenerator* G##_CreateWaveformGenerator(void) { return (G##_WaveformGenerator*)calloc(1, sizeof(G##_WaveformGenerator)); } \
  void G##_DestroyWaveformGenerator(G##_WaveformGenerator* o) { if (o) { free(o->m.blob); free(o); } }          \
  G##_WaveformContext1* G##_CreateWaveformContext1(void) { G##_WaveformContext1* o = (G##_WaveformContext1*)calloc(1, sizeof(*o)); lwave_ctx_init(&o->c); return o; } \
  void G##_DestroyWaveformContext1(G##_WaveformContext1* o) { if (o) { lwave_ctx_free(&o->c); free(o); } }      \
  Beatrice_ErrorCode G##_ReadPhoneExtractorParameters(G##_PhoneExtractor* m, const char* p) { return lphone_read(&m->m, p); } \
  Beatrice_ErrorCode G##_ReadPitchEstimatorParameters(G##_PitchEstimator* m, const char* p) { return lpitch_read(&m->m, p); } \
  Beatrice_ErrorCode G##_ReadWaveformGeneratorParameters(G##_WaveformGenerator* m, const char* p) { return lwave_read(&m->m, p); } \
  Beatrice_ErrorCode G##_ReadNSpeakers(const char* p, int* o) {                                                  \
    float* blob; int rows;                                                                                       \
    const Beatrice_ErrorCode e = lrows_open(p, &blob, &rows);                                                    \
    if (e) return e;                                                                                             \
    free(blob); *o = rows; return Beatrice_kSuccess;                                                             \
  }                                                                                                              \
  Beatrice_ErrorCode G##_ReadSpeakerEmbeddings(const char* p, float* o) {                                        \
    float* blob; int rows;                                                                                       \
    const Beatrice_ErrorCode e = lrows_open(p, &blob, &rows);                                                    \
    if (e) return e;                                                                                             \
    memcpy(o, blob, sizeof(float) * (size_t)rows * HID); free(blob); return Beatrice_kSuccess;                   \
  }                                                                                                              \
  void G##_ExtractPhone1(const G##_PhoneExtractor* m, const float* in, float* out, G##_PhoneContext1* c) { lphone_run(&m->m, in, out, &c->c); } \
  void G##_SetMinQuantizedPitch(G##_PitchContext1* c, int q) { c->c.min_q = lclamp_bin(q); }                     \
  void G##_SetMaxQuantizedPitch(G##_PitchContext1* c, int q) { c->c.max_q = lclamp_bin(q); }                     \
  void G##_EstimatePitch1(const G##_PitchEstimator* m, const float* in, int* q, float* f, G##_PitchContext1* c) { lpitch_run(&m->m, in, q, f, &c->c); } \
  void G##_GenerateWaveform1(const G##_WaveformGenerator* m, const float* ph, const int* q, const float* f, const float* s, float* out, G##_WaveformContext1* c) { lwave_run(&m->m, ph, q, f, s, out, &c->c); }

LEGACY_GENERATION(Beatrice20a2)
LEGACY_GENERATION(Beatrice20b1)
