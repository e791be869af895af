// oracle/ref_drivers/ref_wrapper.cc -- TEST INFRASTRUCTURE: a thin extern "C" driver around the
// REFERENCE's own header-only DSP sources, compiled where they lie (-I/root/reference/src), into
// oracle/_ref/libref_wrapper.so.  No reference source is copied: this file only instantiates
//   beatrice::resampler::AnyFreqInOut        (reference src/common/resample.h:401-438)
//   beatrice::resampler::ComputeSimpleFraction                  (resample.h:25-46)
//   beatrice::common::Gain / Gain::Context   (reference src/common/gain.h:19-72)
//   beatrice::common::SphericalAverage       (reference src/common/spherical_average.h:80-444)
// with a caller-supplied C callback standing in for the per-hop model call, so that the oracle's
// restatement (oracle/wrapper_oracle.c) can be compared with the real thing and golden vectors can
// be minted (tools/make_golden.py).  reference src/common/processor_core_2.cc is NOT built: it
// needs lib/toml11 (an empty submodule here), and writing a stand-in for it is not allowed.
#include <array>  // resample.h uses std::array without including it (reference resample.h:333)
#include <cstring>
#include <vector>

#include "common/gain.h"
#include "common/resample.h"
#include "common/spherical_average.h"

extern "C" {
typedef void (*ref_hop_fn)(const float* in160, float* out240, void* user);
}

namespace {
struct HopCtx { ref_hop_fn fn; void* user; };
struct HopCall {
  void operator()(const float* in, float* out, HopCtx& c) const { c.fn(in, out, c.user); }
};
struct RefProcessor {
  beatrice::resampler::AnyFreqInOut<HopCall> chain;
  beatrice::common::Gain gain;
  beatrice::common::Gain::Context gin, gout;
  HopCtx ctx;
  RefProcessor(double sr, ref_hop_fn fn, void* user) : chain(sr), gin(sr), gout(sr), ctx{fn, user} {}
};
struct RefGain {
  beatrice::common::Gain gain;
  beatrice::common::Gain::Context ctx;
  RefGain(double sr, double db) : ctx(sr, db) {}
};
}  // namespace

extern "C" {

void ref_fraction(double ratio, int* numer, int* denom) {
  const auto f = beatrice::resampler::ComputeSimpleFraction(ratio);
  *numer = f.numer;
  *denom = f.denom;
}

void* ref_create(double sample_rate, ref_hop_fn fn, void* user) { return new RefProcessor(sample_rate, fn, user); }
void ref_destroy(void* p) { delete static_cast<RefProcessor*>(p); }
void ref_set_input_gain(void* p, double db) { static_cast<RefProcessor*>(p)->gin.SetTargetGain(db); }
void ref_set_output_gain(void* p, double db) { static_cast<RefProcessor*>(p)->gout.SetTargetGain(db); }
// same call sequence as ProcessorCore2::Process (reference src/common/processor_core_2.cc:44-46)
int ref_process(void* vp, const float* in, float* out, int n) {
  auto* p = static_cast<RefProcessor*>(vp);
  if (!p->chain.IsReady()) { std::memset(out, 0, sizeof(float) * n); return 10; }
  p->gain.Process(in, out, n, p->gin);
  p->chain(out, out, n, p->ctx);
  p->gain.Process(out, out, n, p->gout);
  return 0;
}

void* ref_gain_create(double sample_rate, double db) { return new RefGain(sample_rate, db); }
void ref_gain_set_target(void* g, double db) { static_cast<RefGain*>(g)->ctx.SetTargetGain(db); }
void ref_gain_process(void* g, const float* in, float* out, int n) {
  auto* r = static_cast<RefGain*>(g);
  r->gain.Process(in, out, n, r->ctx);
}
void ref_gain_destroy(void* g) { delete static_cast<RefGain*>(g); }

// Spherical average of n_points unit-norm-scaled rows of dimension dim (128 or 256) with the given
// weights; mirrors the call sequence of reference src/common/processor_core_2.cc:127-136,385-388.
int ref_spherical_average(int dim, int n_points, const float* points, const float* weights, const int* argsort,
                          int max_speakers, int max_updates, float* out) {
  if (dim == 128) {
    beatrice::common::SphericalAverage<float, 128> s;
    s.Initialize(n_points, dim, points, max_speakers);
    s.SetWeights(n_points, weights, argsort);
    int it = 0;
    for (; it < max_updates; ++it) if (s.Update()) break;
    alignas(64) float tmp[128];
    s.GetResult(dim, tmp);
    std::memcpy(out, tmp, sizeof(tmp));
    return it;
  }
  if (dim == 256) {
    beatrice::common::SphericalAverage<float, 256> s;
    s.Initialize(n_points, dim, points, max_speakers);
    s.SetWeights(n_points, weights, argsort);
    int it = 0;
    for (; it < max_updates; ++it) if (s.Update()) break;
    alignas(64) float tmp[256];
    s.GetResult(dim, tmp);
    std::memcpy(out, tmp, sizeof(tmp));
    return it;
  }
  return -1;
}

}  // extern "C"
