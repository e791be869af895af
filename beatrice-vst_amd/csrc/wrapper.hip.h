// wrapper.hip.h -- the reference host's wrapper around the model hop, for B streams on the device, at ANY host rate
// and block size and with the dB-ramped gains: what ProcessorCore2::Process does per plugin instance
// (reference src/common/processor_core_2.cc:24-48) --
//
//   [stereo downmix (L+R)*0.5, src/vst/processor.cc:183-192] -> input gain (gain.h:41-71) -> host rate to 48 kHz
//   (resample.h:130-206, tables :209-237) -> exact-480 FIFO (+10 ms, :343-363) -> every third sample -> MODEL HOP ->
//   zero-stuffing x2 (:380-394) -> 48 kHz to host rate -> output gain -> every output channel
//
// -- with the arithmetic in the reference's order (float mul-then-add over the taps in ascending tap order, the gain
// ramp as a sequence of double multiplications clamped at the goal), so that results equal the host chain bit for bit.
//
// All streams of a batch share the host rate and the block size and were started together, so everything that is
// CONTROL -- the two fractional clocks of the resampler pair, how many 48 kHz samples a block yields, the FIFO fill,
// when a model hop fires -- is the same for every stream and is tracked on the host (WrapPlan); only data lives on the
// device (per stream: two filter histories and the 480-sample FIFO).  The gains are per-stream settings; the host
// keeps each stream's current gain in dB as the reference does (the dB <-> amplitude conversions stay on the host's
// libm so that they round as the reference's), the device re-creates the per-sample ramp from {start, goal, step}.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

namespace wrapn {

constexpr int kTapsPerOutput = 32;   // filter length in low-rate samples (reference resample.h:415)
constexpr int kBlock = 480;          // 10 ms at 48 kHz
constexpr int kMaxSamples = 4096;    // largest host block per call
constexpr int kMaxChunks = 12;      // FIFO pieces one call can cut its inner samples into: ceil(4096 / 480) + 2
constexpr int kMaxHist = kTapsPerOutput * 8 + 1;   // history of the high-rate side: 32 * hi / lo + 1, host rates up to 384 kHz

struct GainSeg { double amp0, goal, step; };   // per stream and call: amplitude before the first sample, goal, per-sample factor
                                                // (step > 1 ramps up, < 1 down, == 1: constant amp0)

struct StreamState {               // per stream
  float hist_high[kMaxHist];       // newest samples of the high-rate side's filter history (ring replaced by "last H samples")
  float hist_low[kTapsPerOutput + 1];
  float hist_high_out[kMaxHist];   // the second direction keeps its own histories
  float hist_low_out[kTapsPerOutput + 1];
  float fifo[kBlock];
};

// what a resampling direction needs for one call (identical for every stream)
struct Dir {
  int decimate;      // 1: high -> low rate (reference Downsample), 0: low -> high (Upsample)
  int hi, lo;        // rate ratio
  int phase0;        // the direction's fractional clock before the call
  int n_in, n_out;
  int n_taps;        // table length (32 * hi + 1)
  int hist;          // history length kept between calls
  float scale;       // Downsample's make-up gain lo / hi
};

// out[o] of one direction; `x` = [hist | n_in new samples] in LDS.  Tap order and accumulation exactly as the host loops
// (reference resample.h:130-159 / :168-206): acc += x * tap (mul, then add), ascending tap index.
__device__ __forceinline__ float resample_one(const Dir& d, const float* __restrict__ x, const float* __restrict__ taps, const int o) {
  const int last = d.n_taps - 1;
  float acc = 0.0f;
  if (d.decimate) {
    // the o-th output fires when the clock crosses hi for the (o+1)-th time: after k inputs, k = ceil(((o+1) hi - phase0) / lo)
    const long long need = (long long)(o + 1) * d.hi - d.phase0;
    const int k = (int)((need + d.lo - 1) / d.lo);
    const int ph = (int)(d.phase0 + (long long)k * d.lo - (long long)(o + 1) * d.hi);
    int pos = d.hist + k - 1;  // newest pushed sample
    for (int tap = d.lo - ph; tap < last; tap += d.lo) acc = acc + x[pos--] * taps[tap];
    return acc * d.scale;
  }
  const long long clock = d.phase0 + (long long)(o + 1) * d.lo;
  const int pushed = (int)(clock / d.hi), ph = (int)(clock % d.hi);
  int pos = d.hist + pushed - 1;
  for (int tap = ph; tap < last; tap += d.hi) acc = acc + x[pos--] * taps[tap];
  return acc;
}

// The direction's tap table for this workgroup: staged into LDS when it fits (44.1 <-> 48 kHz: 5 121 floats).  A thread's output walks
// 32 taps hi apart in ascending order, each one a dependent round trip to global memory otherwise -- ~0.4 us each, 13 of the 15 us of
// a wrapper launch (profiles/r05_notes.md section 16).  Same values, same order.  Visible after the caller's next barrier.
constexpr int kTapsLds = 7168;
__device__ __forceinline__ const float* stage_taps(const Dir& d, const float* __restrict__ taps, const int tid) {
  __shared__ float tl[kTapsLds];
  if (d.n_taps > kTapsLds) return taps;
  for (int i = tid; i < d.n_taps; i += 256) tl[i] = taps[i];
  return tl;
}

// host block -> 48 kHz: downmix, input gain, first resampling direction.  One workgroup per stream.
__device__ __forceinline__ void wrap_in_body(float* __restrict__ x /* LDS [kMaxHist + kMaxSamples] */, double* __restrict__ amp /* LDS [kMaxSamples] */,
                                             const int b, const int tid, const float* __restrict__ in, const int channels, const int n,
                                             StreamState* __restrict__ st, const GainSeg* __restrict__ gain, const float* __restrict__ taps, const Dir& d,
                                             float* __restrict__ inner /* [B][stride] */, const int stride) {
  const float* src = in + (size_t)b * channels * n;
  float* hist = d.decimate ? st[b].hist_high : st[b].hist_low;
  const GainSeg g = gain[b];
  if (tid == 0 && g.step != 1.0) {  // the ramp is a recurrence: amp <- clamp(amp * step) until the goal is reached (gain.h:52-66)
    double a = g.amp0;
    for (int i = 0; i < n; ++i) {
      if (g.step > 1.0) { if (a < g.goal) a = fmin(a * g.step, g.goal); }
      else if (a > g.goal) a = fmax(a * g.step, g.goal);
      amp[i] = a;
    }
  }
  const float* tl_ = stage_taps(d, taps, tid);
  for (int i = tid; i < d.hist; i += 256) x[i] = hist[i];
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    float m = src[i];
    if (channels >= 2) { m = m + src[n + i]; m = m * 0.5f; }
    const double a = g.step != 1.0 ? amp[i] : g.amp0;
    x[d.hist + i] = (float)(m * a);
  }
  __syncthreads();
  for (int o = tid; o < d.n_out; o += 256) inner[(size_t)b * stride + o] = resample_one(d, x, tl_, o);
  __syncthreads();
  for (int i = tid; i < d.hist; i += 256) hist[i] = x[n + i];  // the newest `hist` samples
}
static __global__ __launch_bounds__(256) void wrap_in_kernel(const float* __restrict__ in, const int channels, const int n, StreamState* __restrict__ st,
                                                             const GainSeg* __restrict__ gain, const float* __restrict__ taps, const Dir d,
                                                             float* __restrict__ inner /* [B][stride] */, const int stride) {
  __shared__ float x[kMaxHist + kMaxSamples];
  __shared__ double amp[kMaxSamples];
  wrap_in_body(x, amp, blockIdx.x, threadIdx.x, in, channels, n, st, gain, taps, d, inner, stride);
}

// 48 kHz -> host block: second direction, output gain, every channel
static __global__ __launch_bounds__(256) void wrap_out_kernel(const float* __restrict__ inner, const int stride, StreamState* __restrict__ st,
                                                              const GainSeg* __restrict__ gain, const float* __restrict__ taps, const Dir d,
                                                              float* __restrict__ out, const int channels) {
  __shared__ float x[kMaxHist + kMaxSamples];
  __shared__ double amp[kMaxSamples];
  const int b = blockIdx.x, tid = threadIdx.x, n = d.n_out;
  float* hist = d.decimate ? st[b].hist_high_out : st[b].hist_low_out;
  const GainSeg g = gain[b];
  if (tid == 0 && g.step != 1.0) {
    double a = g.amp0;
    for (int i = 0; i < n; ++i) {
      if (g.step > 1.0) { if (a < g.goal) a = fmin(a * g.step, g.goal); }
      else if (a > g.goal) a = fmax(a * g.step, g.goal);
      amp[i] = a;
    }
  }
  const float* tl_ = stage_taps(d, taps, tid);
  for (int i = tid; i < d.hist; i += 256) x[i] = hist[i];
  for (int i = tid; i < d.n_in; i += 256) x[d.hist + i] = inner[(size_t)b * stride + i];
  __syncthreads();
  float* dst = out + (size_t)b * channels * n;
  for (int o = tid; o < n; o += 256) {
    const float y = resample_one(d, x, tl_, o);
    const double a = g.step != 1.0 ? amp[o] : g.amp0;
    const float v = (float)(y * a);
    for (int c = 0; c < channels; ++c) dst[c * n + o] = v;
  }
  __syncthreads();
  for (int i = tid; i < d.hist; i += 256) hist[i] = x[d.n_in + i];
}

// The output half of the chain for a batch whose model hops run in the tick pipeline (BeatriceBatch_BindResidentBlocks): the
// 480-sample FIFO only DELAYS -- 48 kHz sample t of the stream leaving it is sample t % 480 of the zero-stuffed output of
// model hop t / 480 - 1 (zeros before the first hop; reference resample.h:343-363, 380-394) -- so the block's inner samples
// [t0, t0 + m) are gathered straight from the resident model outputs (hop k in slot k mod io_slots) once those hops have
// left the pipeline, and go through the second resampling direction and the output gain exactly as in wrap_out_kernel.
// H hops per step (a batch of BeatriceBatch_CreateBlock): hop k is hop k % H of step k / H, in slot (k / H) mod io_slots.
__device__ __forceinline__ void wrap_post_body(float* __restrict__ x, double* __restrict__ amp, const int b, const int tid,
                                               const float* __restrict__ out24 /* [io_slots][B][H][240] */, const int io_slots, const int B, const int H,
                                               const long long t0, StreamState* __restrict__ st, const GainSeg* __restrict__ gain,
                                               const float* __restrict__ taps, const Dir& d, float* __restrict__ out, const int channels) {
  const int n = d.n_out;
  float* hist = d.decimate ? st[b].hist_high_out : st[b].hist_low_out;
  const GainSeg g = gain[b];
  if (tid == 0 && g.step != 1.0) {
    double a = g.amp0;
    for (int i = 0; i < n; ++i) {
      if (g.step > 1.0) { if (a < g.goal) a = fmin(a * g.step, g.goal); }
      else if (a > g.goal) a = fmax(a * g.step, g.goal);
      amp[i] = a;
    }
  }
  const float* tl_ = stage_taps(d, taps, tid);
  for (int i = tid; i < d.hist; i += 256) x[i] = hist[i];
  for (int i = tid; i < d.n_in; i += 256) {
    const long long t = t0 + i;
    const long long k = t / kBlock - 1;
    const int off = (int)(t % kBlock);
    x[d.hist + i] = (k < 0 || (off & 1)) ? 0.0f : out24[(((size_t)((k / H) % io_slots) * B + b) * H + (int)(k % H)) * 240 + (off >> 1)];
  }
  __syncthreads();
  float* dst = out + (size_t)b * channels * n;
  for (int o = tid; o < n; o += 256) {
    const float y = resample_one(d, x, tl_, o);
    const double a = g.step != 1.0 ? amp[o] : g.amp0;
    const float v = (float)(y * a);
    for (int c = 0; c < channels; ++c) dst[c * n + o] = v;
  }
  __syncthreads();
  for (int i = tid; i < d.hist; i += 256) hist[i] = x[d.n_in + i];
}
static __global__ __launch_bounds__(256) void wrap_post_kernel(const float* __restrict__ out24 /* [io_slots][B][H][240] */, const int io_slots, const int B, const int H,
                                                               const long long t0, StreamState* __restrict__ st, const GainSeg* __restrict__ gain,
                                                               const float* __restrict__ taps, const Dir d, float* __restrict__ out, const int channels) {
  __shared__ float x[kMaxHist + kMaxSamples];
  __shared__ double amp[kMaxSamples];
  wrap_post_body(x, amp, blockIdx.x, threadIdx.x, out24, io_slots, B, H, t0, st, gain, taps, d, out, channels);
}

// The exact-480 FIFO (reference resample.h:343-363): samples [at, at + take) of the 48 kHz stream swap places with
// FIFO positions [fill, fill + take) -- the stream gets what the previous block left there (its processed output), the
// FIFO gets the new input.  When that completes the block (fires != 0) every third sample goes to the model's input
// (in16: the hop's place in the first stream's row of the step's slot, row16 floats per stream: 160 x hops per step).
__device__ __forceinline__ void wrap_fifo_body(const int b, const int tid, float* __restrict__ inner, const int stride, StreamState* __restrict__ st, const int at,
                                               const int fill, const int take, const int fires, float* __restrict__ in16, const int row16) {
  float* f = st[b].fifo;
  for (int i = tid; i < take; i += 256) {
    const float fresh = inner[(size_t)b * stride + at + i];
    inner[(size_t)b * stride + at + i] = f[fill + i];
    f[fill + i] = fresh;
  }
  if (!fires) return;
  __syncthreads();
  for (int i = tid; i < 160; i += 256) in16[(size_t)b * row16 + i] = f[3 * i + 2];
}
static __global__ __launch_bounds__(256) void wrap_fifo_kernel(float* __restrict__ inner, const int stride, StreamState* __restrict__ st, const int at,
                                                               const int fill, const int take, const int fires, float* __restrict__ in16, const int row16) {
  wrap_fifo_body(blockIdx.x, threadIdx.x, inner, stride, st, at, fill, take, fires, in16, row16);
}
// One launch per call of the wrapper around the tick pipeline (BeatriceBatch_BindResidentBlocks): workgroups [0, B) run the call's
// input half -- gains, first resampling direction, then every piece the 480-sample FIFO cuts the inner samples into, each fired hop
// straight into its place of its step's resident slot (the pieces only move wrapper state: none depends on a tick) --, workgroups
// [B, 2 B) the output half of the call made `delay` calls ago (n_post = 0: none is due).  The gain segments are read where the host
// wrote them (pinned memory): no copy command.  Three to four launches and a copy per call were serial with the tick launch before.
struct WrapCallArgs {
  const float* src; int channels, n;
  StreamState* st;
  const GainSeg* gain_in;
  const float* taps_in; Dir din;
  float* inner; int stride;
  int n_chunks;
  short at[kMaxChunks], fill[kMaxChunks], take[kMaxChunks];
  unsigned char fires[kMaxChunks];
  long long in16_off[kMaxChunks];   // floats from in16 to the hop's place in the first stream's row of its slot
  float* in16; int row16;
  int n_post;
  const float* out24; int io_slots, B, H;
  long long t0;
  const GainSeg* gain_out;
  const float* taps_out; Dir dout;
  float* out;
};
static __global__ __launch_bounds__(256) void wrap_call_kernel(const WrapCallArgs a) {
  __shared__ float x[kMaxHist + kMaxSamples];
  __shared__ double amp[kMaxSamples];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < a.B) {
    const int b = blockIdx.x;
    wrap_in_body(x, amp, b, tid, a.src, a.channels, a.n, a.st, a.gain_in, a.taps_in, a.din, a.inner, a.stride);
    for (int c = 0; c < a.n_chunks; ++c) {
      __syncthreads();   // (the pieces follow each other in the stream's inner samples and its FIFO)
      wrap_fifo_body(b, tid, a.inner, a.stride, a.st, a.at[c], a.fill[c], a.take[c], a.fires[c], a.in16 + a.in16_off[c], a.row16);
    }
  } else {
    wrap_post_body(x, amp, (int)blockIdx.x - a.B, tid, a.out24, a.io_slots, a.B, a.H, a.t0, a.st, a.gain_out, a.taps_out, a.dout, a.out, a.channels);
  }
}
// the model's 240 samples, zero-stuffed, become the FIFO's content (reference resample.h:390-393)
static __global__ void wrap_refill_kernel(StreamState* __restrict__ st, const float* __restrict__ out24, const int B) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * kBlock) return;
  const int b = idx / kBlock, i = idx % kBlock;
  st[b].fifo[i] = (i & 1) ? 0.0f : out24[(size_t)b * 240 + (i >> 1)];
}

// ---- the same chain with clocks PER STREAM (BeatriceBatch_ConfigureWrapperRates / ProcessBlocksRagged): every stream has its
// own host rate, its own block length per call and its own place in the 480-sample FIFO -- what the reference gives every
// plugin instance (src/common/resample.h:401-438, processor_core_2.h:28) -- and may sit a call out altogether (no block, or
// a block the shell's rule calls silent: src/vst/processor.cc:204-214).  Control stays on the host, now per stream; the
// kernels read a per-stream record instead of one set of arguments.
struct RagStream {
  Dir din, dout;                 // this call's two directions (n_in == 0: the stream sits the call out)
  long long io_off;              // floats from the start of the call's buffers to this stream's planar block [channels][n]
  int n;                         // host samples of the block
  int taps_in, taps_out;         // float offsets of the two tables in the batch's tap buffer
  int n_chunks;
  short at[kMaxChunks], fill[kMaxChunks], take[kMaxChunks];   // chunk i: inner samples [at, at + take) <-> FIFO [fill, fill + take)
  unsigned char fires[kMaxChunks];
  int active;                    // 0: nothing of this stream moves in this call (its output block is zeros)
  // around the tick pipeline (BeatriceBatch_BindResidentBlocksRagged): the stream's own 48 kHz sample count and model-hop count
  // at the start of the call
  int hop0;
  long long t0;
};

__device__ __forceinline__ void wrapr_in_body(float* __restrict__ x, double* __restrict__ amp, const int b, const int tid, const float* __restrict__ in,
                                              const int channels, StreamState* __restrict__ st, const GainSeg* __restrict__ gain,
                                              const float* __restrict__ taps_all, const RagStream& r, float* __restrict__ inner, const int stride) {
  if (!r.active) return;
  const Dir d = r.din;
  const int n = r.n;
  const float* src = in + r.io_off;
  const float* taps = taps_all + r.taps_in;
  float* hist = d.decimate ? st[b].hist_high : st[b].hist_low;
  const GainSeg g = gain[b];
  if (tid == 0 && g.step != 1.0) {
    double a = g.amp0;
    for (int i = 0; i < n; ++i) {
      if (g.step > 1.0) { if (a < g.goal) a = fmin(a * g.step, g.goal); }
      else if (a > g.goal) a = fmax(a * g.step, g.goal);
      amp[i] = a;
    }
  }
  const float* tl_ = stage_taps(d, taps, tid);
  for (int i = tid; i < d.hist; i += 256) x[i] = hist[i];
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    float m = src[i];
    if (channels >= 2) { m = m + src[n + i]; m = m * 0.5f; }
    const double a = g.step != 1.0 ? amp[i] : g.amp0;
    x[d.hist + i] = (float)(m * a);
  }
  __syncthreads();
  for (int o = tid; o < d.n_out; o += 256) inner[(size_t)b * stride + o] = resample_one(d, x, tl_, o);
  __syncthreads();
  for (int i = tid; i < d.hist; i += 256) hist[i] = x[n + i];
}
static __global__ __launch_bounds__(256) void wrapr_in_kernel(const float* __restrict__ in, const int channels, StreamState* __restrict__ st,
                                                              const GainSeg* __restrict__ gain, const float* __restrict__ taps_all,
                                                              const RagStream* __restrict__ rs, float* __restrict__ inner, const int stride) {
  __shared__ float x[kMaxHist + kMaxSamples];
  __shared__ double amp[kMaxSamples];
  const RagStream r = rs[blockIdx.x];
  wrapr_in_body(x, amp, blockIdx.x, threadIdx.x, in, channels, st, gain, taps_all, r, inner, stride);
}
static __global__ __launch_bounds__(256) void wrapr_out_kernel(const float* __restrict__ inner, const int stride, StreamState* __restrict__ st,
                                                               const GainSeg* __restrict__ gain, const float* __restrict__ taps_all,
                                                               const RagStream* __restrict__ rs, float* __restrict__ out, const int channels) {
  __shared__ float x[kMaxHist + kMaxSamples];
  __shared__ double amp[kMaxSamples];
  const int b = blockIdx.x, tid = threadIdx.x;
  const RagStream r = rs[b];
  float* dst = out + r.io_off;
  if (!r.active) {   // the stream sat the call out: its block comes back as silence (the shell hands the zero block through)
    for (int i = tid; i < channels * r.n; i += 256) dst[i] = 0.0f;
    return;
  }
  const Dir d = r.dout;
  const int n = d.n_out;
  const float* taps = taps_all + r.taps_out;
  float* hist = d.decimate ? st[b].hist_high_out : st[b].hist_low_out;
  const GainSeg g = gain[b];
  if (tid == 0 && g.step != 1.0) {
    double a = g.amp0;
    for (int i = 0; i < n; ++i) {
      if (g.step > 1.0) { if (a < g.goal) a = fmin(a * g.step, g.goal); }
      else if (a > g.goal) a = fmax(a * g.step, g.goal);
      amp[i] = a;
    }
  }
  const float* tl_ = stage_taps(d, taps, tid);
  for (int i = tid; i < d.hist; i += 256) x[i] = hist[i];
  for (int i = tid; i < d.n_in; i += 256) x[d.hist + i] = inner[(size_t)b * stride + i];
  __syncthreads();
  for (int o = tid; o < n; o += 256) {
    const float y = resample_one(d, x, tl_, o);
    const double a = g.step != 1.0 ? amp[o] : g.amp0;
    const float v = (float)(y * a);
    for (int c = 0; c < channels; ++c) dst[c * n + o] = v;
  }
  __syncthreads();
  for (int i = tid; i < d.hist; i += 256) hist[i] = x[d.n_in + i];
}
// chunk `ci` of every stream that has one; frozen[b] = 1 for the streams that do NOT complete a 480-block in this chunk (the
// model step that follows stands still for them); in16: the step's input.  Around the tick pipeline (slot_map != nullptr): the hop a
// stream fires here is the stream's hop number g = hop0 + its fires in earlier chunks, and the step it rides in is the one in resident
// slot `slot`: slot_map[b][g mod map_ring] = slot, for the output half of the calls that will read that hop (wrapr_post_kernel).
__device__ __forceinline__ void wrapr_fifo_body(const int b, const int tid, float* __restrict__ inner, const int stride, StreamState* __restrict__ st,
                                                const RagStream& r, const int ci, float* __restrict__ in16, unsigned char* __restrict__ frozen,
                                                int* __restrict__ slot_map, const int map_ring, const int slot) {
  const bool has = r.active && ci < r.n_chunks;
  const bool fires = has && r.fires[ci];
  if (frozen != nullptr && tid == 0) frozen[b] = fires ? 0 : 1;
  if (slot_map != nullptr && fires && tid == 0) {
    int g = r.hop0;
    for (int j = 0; j < ci; ++j) g += r.fires[j];
    slot_map[(size_t)b * map_ring + g % map_ring] = slot;
  }
  if (!has) return;
  const int at = r.at[ci], fill = r.fill[ci], take = r.take[ci];
  float* f = st[b].fifo;
  for (int i = tid; i < take; i += 256) {
    const float fresh = inner[(size_t)b * stride + at + i];
    inner[(size_t)b * stride + at + i] = f[fill + i];
    f[fill + i] = fresh;
  }
  if (!fires) return;
  __syncthreads();
  for (int i = tid; i < 160; i += 256) in16[(size_t)b * 160 + i] = f[3 * i + 2];
}
static __global__ __launch_bounds__(256) void wrapr_fifo_kernel(float* __restrict__ inner, const int stride, StreamState* __restrict__ st,
                                                                const RagStream* __restrict__ rs, const int ci, float* __restrict__ in16,
                                                                unsigned char* __restrict__ frozen, int* __restrict__ slot_map, const int map_ring,
                                                                const int slot) {
  wrapr_fifo_body(blockIdx.x, threadIdx.x, inner, stride, st, rs[blockIdx.x], ci, in16, frozen, slot_map, map_ring, slot);
}
// One launch for the input half of a call with per-stream clocks around the tick pipeline: a stream's gains and first resampling
// direction, then every piece ITS FIFO cuts its inner samples into (the pieces only move wrapper state; the steps they fill are fed
// after the launch); piece c of any stream belongs to the call's c-th step, whose resident slot is slot[c].  Records and gain segments
// are read where the host wrote them (pinned memory).
struct WraprCallArgs {
  const float* in; int channels;
  StreamState* st;
  const GainSeg* gain;
  const float* taps_all;
  const RagStream* rs;
  float* inner; int stride;
  float* in16; int B;
  int slot[kMaxChunks];
  int* slot_map; int map_ring;
};
static __global__ __launch_bounds__(256) void wrapr_call_kernel(const WraprCallArgs a) {
  __shared__ float x[kMaxHist + kMaxSamples];
  __shared__ double amp[kMaxSamples];
  const int b = blockIdx.x, tid = threadIdx.x;
  const RagStream r = a.rs[b];
  if (!r.active) return;
  wrapr_in_body(x, amp, b, tid, a.in, a.channels, a.st, a.gain, a.taps_all, r, a.inner, a.stride);
  for (int c = 0; c < r.n_chunks; ++c) {
    __syncthreads();
    wrapr_fifo_body(b, tid, a.inner, a.stride, a.st, r, c, a.in16 + (size_t)a.slot[c] * a.B * 160, nullptr, a.slot_map, a.map_ring, a.slot[c]);
  }
}
// The output half of a call with per-stream clocks around the tick pipeline: as wrap_post_kernel, with the stream's own sample
// count t0 and its own hops -- 48 kHz sample t of stream b is sample t % 480 of the zero-stuffed output of ITS hop t / 480 - 1, which
// rode in the step of resident slot slot_map[b][hop mod map_ring].
static __global__ __launch_bounds__(256) void wrapr_post_kernel(const float* __restrict__ out24 /* [io_slots][B][240] */, const int B,
                                                                const int* __restrict__ slot_map, const int map_ring, StreamState* __restrict__ st,
                                                                const GainSeg* __restrict__ gain, const float* __restrict__ taps_all,
                                                                const RagStream* __restrict__ rs, float* __restrict__ out, const int channels) {
  __shared__ float x[kMaxHist + kMaxSamples];
  __shared__ double amp[kMaxSamples];
  const int b = blockIdx.x, tid = threadIdx.x;
  const RagStream r = rs[b];
  float* dst = out + r.io_off;
  if (!r.active) {   // the stream sat the call out: its block comes back as silence
    for (int i = tid; i < channels * r.n; i += 256) dst[i] = 0.0f;
    return;
  }
  const Dir d = r.dout;
  const int n = d.n_out;
  const float* taps = taps_all + r.taps_out;
  float* hist = d.decimate ? st[b].hist_high_out : st[b].hist_low_out;
  const GainSeg g = gain[b];
  if (tid == 0 && g.step != 1.0) {
    double a = g.amp0;
    for (int i = 0; i < n; ++i) {
      if (g.step > 1.0) { if (a < g.goal) a = fmin(a * g.step, g.goal); }
      else if (a > g.goal) a = fmax(a * g.step, g.goal);
      amp[i] = a;
    }
  }
  const float* tl_ = stage_taps(d, taps, tid);
  for (int i = tid; i < d.hist; i += 256) x[i] = hist[i];
  for (int i = tid; i < d.n_in; i += 256) {
    const long long t = r.t0 + i;
    const long long k = t / kBlock - 1;
    const int off = (int)(t % kBlock);
    x[d.hist + i] = (k < 0 || (off & 1)) ? 0.0f : out24[((size_t)slot_map[(size_t)b * map_ring + (int)(k % map_ring)] * B + b) * 240 + (off >> 1)];
  }
  __syncthreads();
  for (int o = tid; o < n; o += 256) {
    const float y = resample_one(d, x, tl_, o);
    const double a = g.step != 1.0 ? amp[o] : g.amp0;
    const float v = (float)(y * a);
    for (int c = 0; c < channels; ++c) dst[c * n + o] = v;
  }
  __syncthreads();
  for (int i = tid; i < d.hist; i += 256) hist[i] = x[d.n_in + i];
}
static __global__ void wrapr_refill_kernel(StreamState* __restrict__ st, const float* __restrict__ out24, const int B, const unsigned char* __restrict__ frozen) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * kBlock) return;
  const int b = idx / kBlock, i = idx % kBlock;
  if (frozen[b]) return;
  st[b].fifo[i] = (i & 1) ? 0.0f : out24[(size_t)b * 240 + (i >> 1)];
}

// ---- host side: the clocks and the tables -------------------------------------------------------------------------
// smallest-denominator search on the Stern-Brocot tree, parts < 1000 (reference resample.h:25-46)
inline void simple_fraction(double ratio, int* numer, int* denom) {
  int a = 0, b = 1, c = 1, d = 0;
  for (;;) {
    const int mn = a + c, md = b + d;
    const bool big = mn >= 1000 || md >= 1000;
    if (ratio * md < mn) {
      if (big) { *numer = a; *denom = b; return; }
      c = mn; d = md;
    } else {
      if (big) { *numer = c; *denom = d; return; }
      a = mn; b = md;
    }
  }
}

struct WrapPlan {
  bool ready = false, high_is_outer = true;
  double rate = 0.0;
  int hi = 1, lo = 1, phase_down = 0, phase_up = 0, fill = 0;
  std::vector<float> taps_down, taps_up;
  int hist_high() const { return kTapsPerOutput * hi / lo + 1; }
  int hist_low() const { return kTapsPerOutput + 1; }
  // reference resample.h:209-237 with the cutoffs of :412-417
  bool configure(double sr) {
    ready = false;
    rate = sr;
    if (!(sr > 0.0)) return false;
    const double inner = 48000.0, pi = 3.14159265358979323846;
    const double cutoff_in = 0.99 * 16000.0 / std::fmin(std::fmax(sr, 16000.0), 48000.0), cutoff_out = 0.99 * 24000.0 / std::fmin(std::fmax(sr, 24000.0), 48000.0);
    high_is_outer = sr >= inner;
    const double high = high_is_outer ? sr : inner, low = high_is_outer ? inner : sr;
    const double cut_down = high_is_outer ? cutoff_in : cutoff_out, cut_up = high_is_outer ? cutoff_out : cutoff_in;
    simple_fraction(high / low, &hi, &lo);
    if (hi == 0 || lo == 0 || hist_high() > kMaxHist) return false;
    const int n = kTapsPerOutput * hi + 1, mid = n / 2;
    taps_down.resize(n);
    taps_up.resize(n);
    auto sinc = [pi](double x) { return std::abs(x) < 1e-8 ? 1.0 : std::sin(x * pi) / (x * pi); };
    for (int i = 0; i < n; ++i) {
      const double x = static_cast<double>(i - mid) / static_cast<double>(hi);
      const double hann = 0.5 - 0.5 * std::cos(pi * 2.0 / static_cast<double>(n - 1) * static_cast<double>(i));
      taps_down[i] = static_cast<float>(cut_down * sinc(x * cut_down) * hann);
      taps_up[i] = static_cast<float>(cut_up * sinc(x * cut_up) * hann);
    }
    phase_down = phase_up = hi - 1;
    fill = 0;
    ready = true;
    return true;
  }
  // the direction host -> 48 kHz for a block of n host samples; advances that direction's clock
  Dir to_inner(int n) {
    Dir d{};
    d.hi = hi; d.lo = lo; d.n_in = n; d.n_taps = (int)taps_down.size();
    if (high_is_outer) {   // Downsample (reference resample.h:130-159)
      d.decimate = 1; d.phase0 = phase_down; d.hist = hist_high(); d.scale = static_cast<float>(lo) / static_cast<float>(hi);
      const long long total = phase_down + (long long)n * lo;
      d.n_out = (int)(total / hi);
      phase_down = (int)(total % hi);
    } else {               // Upsample (:168-206)
      d.decimate = 0; d.phase0 = phase_up; d.hist = hist_low(); d.scale = 1.0f;
      d.n_out = (int)((((long long)n + 1) * hi - phase_up - 1) / lo);
      phase_up = (int)((phase_up + (long long)d.n_out * lo) % hi);
    }
    return d;
  }
  // the direction 48 kHz -> host for m inner samples
  Dir to_outer(int m) {
    Dir d{};
    d.hi = hi; d.lo = lo; d.n_in = m; d.n_taps = (int)taps_up.size();
    if (high_is_outer) {   // Upsample; the count couples the two clocks (reference resample.h:171-177)
      d.decimate = 0; d.phase0 = phase_up; d.hist = hist_low(); d.scale = 1.0f;
      d.n_out = (int)(((long long)m * hi + phase_down - phase_up) / lo);
      phase_up = (int)((phase_up + (long long)d.n_out * lo) % hi);
    } else {
      d.decimate = 1; d.phase0 = phase_down; d.hist = hist_high(); d.scale = static_cast<float>(lo) / static_cast<float>(hi);
      const long long total = phase_down + (long long)m * lo;
      d.n_out = (int)(total / hi);
      phase_down = (int)(total % hi);
    }
    return d;
  }
};

// the reference's gain context (gain.h:19-72), per stream, on the host: now in dB as a double
struct GainClock {
  double target_db = 0.0, now_db = 0.0;
  static double db_to_amp(double db) { return std::pow(10.0, db * 0.05); }
  // this call's segment + the state after n samples (the same loop the host layer's GainRamp::Apply runs, without data)
  GainSeg advance(int n, double rate) {
    const double goal = db_to_amp(target_db);
    double amp = db_to_amp(now_db);
    GainSeg g{amp, goal, 1.0};
    const double per_sample_db = 2.0 / (rate * 0.001);
    int i = 0;
    if (amp < goal) {
      g.step = db_to_amp(per_sample_db);
      while (i < n && amp < goal) { amp = std::fmin(amp * g.step, goal); ++i; }
    } else if (amp > goal) {
      g.step = db_to_amp(-per_sample_db);
      while (i < n && amp > goal) { amp = std::fmax(amp * g.step, goal); ++i; }
    }
    now_db = 20.0 * std::log10(amp);
    return g;
  }
};

}  // namespace wrapn
