// morph.hip -- speaker morphing on the device: the weighted spherical means that turn the embeddings
// of up to eight real speakers into one synthetic speaker.
//
// What it replaces: the morph branch of the reference host (reference src/common/processor_core_2.cc:
// 124-172), which runs SphericalAverage<float, M> (reference src/common/spherical_average.h:80-444) on
// the audio thread -- one solve for the additive embedding (M = 256) and 384 solves for the key/value
// tokens (M = 128), spread over four hops because they cost ~1.6 ms each hop on a CPU.  The solves are
// independent, so here each one is ONE wavefront: the M dimensions live across the 64 lanes, every dot
// product is a lane-partial followed by an xor-butterfly wave sum, and all 385 run in one launch.
//
// Same algorithm, same constants, same quirks as host/spherical_mean.h (which is bit-exact to the
// reference): start from the normalised weighted mean, at most four Buss-Fillmore steps preconditioned
// by a two-slot L-BFGS memory, result = combination of the un-normalised points.  Not the same rounding:
// the host sums dot products sequentially and uses glibc's acos/sin, the device sums across lanes and
// uses the ROCm device library, so results agree to float rounding (tests: <= 2e-6 relative), not bit
// for bit.  Nothing downstream of these embeddings takes a discrete decision, so PCM stays within 1e-4.
#include <hip/hip_runtime.h>

#include "engine.h"
#include "spec_math.hip.h"

namespace bhip {

namespace {

constexpr int kMaxPoints = 8;  // reference processor_core_2.h:26 (kSphAvgMaxNSpeakers)
constexpr int kMaxSteps = 4;   // reference processor_core_2.h:137 (kSphAvgMaxNUpdates)

struct SphMeanArgs {
  const float* table;   // [speaker][rows][dim] raw embeddings
  float* out;           // [rows][dim] of the destination speaker
  size_t speaker_stride;
  int n_active;
  int speaker[kMaxPoints];
  float weight[kMaxPoints];  // already normalised to sum 1, descending
};

// DPL = dimensions per lane (dim = 64 * DPL); lane owns dimensions lane + 64 * i
template <int DPL>
__global__ __launch_bounds__(64) void sph_mean_kernel(const SphMeanArgs a) {
  constexpr int DIM = 64 * DPL;
  const int row = blockIdx.x, lane = threadIdx.x, N = a.n_active;
  const float eps = 1.1920928955078125e-07f;  // FLT_EPSILON

  auto dot = [&](const float* x, const float* y) {
    float p = 0.0f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) p = p + x[i] * y[i];
    return bsp::wsum64(p);
  };
  auto normalize = [&](float* x) -> bool {
    const float norm = sqrtf(dot(x, x));
    if (!(norm > 0.0f)) return false;
    const float inv = 1.0f / norm;
#pragma unroll
    for (int i = 0; i < DPL; ++i) x[i] = x[i] * inv;
    return true;
  };
  // sin(x)/x as the reference evaluates it: the magnitude is truncated to an integer before the
  // small-angle tests (unqualified abs() on a float, see host/spherical_mean.h), so 1 for |x| < 1
  auto sinc = [&](float x) -> float {
    const int ax = abs(static_cast<int>(x));
    return ax >= 1 ? static_cast<float>(sin(static_cast<double>(x)) / static_cast<double>(x)) : 1.0f;
  };

  float raw[kMaxPoints][DPL], unit[kMaxPoints][DPL], coef[kMaxPoints];
#pragma unroll
  for (int n = 0; n < kMaxPoints; ++n) {
    coef[n] = 0.0f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      raw[n][i] = n < N ? a.table[(size_t)a.speaker[n] * a.speaker_stride + (size_t)row * DIM + lane + 64 * i] : 0.0f;
      unit[n][i] = raw[n][i];
    }
    if (n < N) normalize(unit[n]);
  }

  float q[DPL], g[DPL], d[DPL];
  // L-BFGS memory: two (step, gradient change) pairs, selected by a run-time slot -> kept in LDS
  __shared__ float mem_s[2][DIM], mem_t[2][DIM];
  float rho[2] = {0.0f, 0.0f}, alpha[2] = {0.0f, 0.0f};
  int slot = 0;
  float gamma = 1.0f;
#pragma unroll
  for (int i = 0; i < DPL; ++i) {
    q[i] = 0.0f; g[i] = 0.0f; d[i] = 0.0f;
    mem_s[0][lane + 64 * i] = 0.0f; mem_s[1][lane + 64 * i] = 0.0f;
    mem_t[0][lane + 64 * i] = 0.0f; mem_t[1][lane + 64 * i] = 0.0f;
  }

  // coefficients, tangent gradient and preconditioned direction at the current q
  auto gradient = [&]() {
    float denom = 0.0f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) g[i] = 0.0f;
#pragma unroll
    for (int n = 0; n < kMaxPoints; ++n) {
      if (n >= N) continue;
      float c = dot(unit[n], q);
      c = c < -1.0f ? -1.0f : (c > 1.0f ? 1.0f : c);
      const float theta = static_cast<float>(acos(static_cast<double>(c)));
      const float inv_sinc = 1.0f / (sinc(theta) + eps);
      denom = denom + a.weight[n] * c * inv_sinc;
      coef[n] = a.weight[n] * inv_sinc;
      const float k = -2.0f * coef[n];
#pragma unroll
      for (int i = 0; i < DPL; ++i) g[i] = g[i] + k * unit[n][i];
    }
    const float inv_denom = 1.0f / (denom + eps);
#pragma unroll
    for (int n = 0; n < kMaxPoints; ++n) coef[n] = coef[n] * inv_denom;
    {  // project g onto the tangent space at q
      const float k = -dot(q, g);
#pragma unroll
      for (int i = 0; i < DPL; ++i) g[i] = g[i] + k * q[i];
    }
#pragma unroll
    for (int i = 0; i < DPL; ++i) d[i] = g[i];
    for (int k = 0; k < 2; ++k) {  // two-loop recursion, newest pair first
      const int m = (slot - k - 1 + 2) % 2;
      float sv[DPL], tv[DPL];
#pragma unroll
      for (int i = 0; i < DPL; ++i) { sv[i] = mem_s[m][lane + 64 * i]; tv[i] = mem_t[m][lane + 64 * i]; }
      const float al = rho[m] * dot(sv, d);
      if (m == 0) alpha[0] = al; else alpha[1] = al;
#pragma unroll
      for (int i = 0; i < DPL; ++i) d[i] = d[i] + (-al) * tv[i];
    }
#pragma unroll
    for (int i = 0; i < DPL; ++i) d[i] = d[i] * gamma;
    for (int k = 0; k < 2; ++k) {
      const int m = (slot + k) % 2;
      float sv[DPL], tv[DPL];
#pragma unroll
      for (int i = 0; i < DPL; ++i) { sv[i] = mem_s[m][lane + 64 * i]; tv[i] = mem_t[m][lane + 64 * i]; }
      const float beta = rho[m] * dot(tv, d);
      const float al = m == 0 ? alpha[0] : alpha[1];
#pragma unroll
      for (int i = 0; i < DPL; ++i) d[i] = d[i] + (al - beta) * sv[i];
    }
  };

  bool started = false;
  if (N > 0) {
#pragma unroll
    for (int n = 0; n < kMaxPoints; ++n) {
      if (n >= N) continue;
#pragma unroll
      for (int i = 0; i < DPL; ++i) q[i] = n == 0 ? a.weight[0] * unit[0][i] : q[i] + a.weight[n] * unit[n][i];
    }
    started = normalize(q);
  }
  if (started) {
    gradient();
    for (int it = 0; it < kMaxSteps; ++it) {
      const float step = sqrtf(dot(d, d));
      if (!(step >= 8.0f * eps)) break;  // converged
      float sv[DPL], tv[DPL];
#pragma unroll
      for (int i = 0; i < DPL; ++i) { sv[i] = q[i]; q[i] = q[i] - d[i]; }
      normalize(q);
#pragma unroll
      for (int i = 0; i < DPL; ++i) { sv[i] = q[i] - sv[i]; tv[i] = g[i]; }
      // as on the host, the recursion inside this gradient() already sees the slot being replaced
      // with its NEW step and with the OLD GRADIENT parked where the gradient change will go
#pragma unroll
      for (int i = 0; i < DPL; ++i) { mem_s[slot][lane + 64 * i] = sv[i]; mem_t[slot][lane + 64 * i] = tv[i]; }
      gradient();
#pragma unroll
      for (int i = 0; i < DPL; ++i) tv[i] = g[i] - tv[i];
      {
        const float k = -dot(q, tv);
#pragma unroll
        for (int i = 0; i < DPL; ++i) tv[i] = tv[i] + k * q[i];
      }
#pragma unroll
      for (int i = 0; i < DPL; ++i) mem_t[slot][lane + 64 * i] = tv[i];
      gamma = dot(sv, tv);
      const float r = 1.0f / gamma;
      if (slot == 0) rho[0] = r; else rho[1] = r;
      gamma = gamma / dot(tv, tv);
      slot = (slot + 1) % 2;
    }
  }
  // un-normalised combination of the ORIGINAL points with the final coefficients (zeros when the
  // weighted mean degenerated, like the host)
  float* o = a.out + (size_t)row * DIM;
#pragma unroll
  for (int i = 0; i < DPL; ++i) {
    float y = coef[0] * raw[0][i];
#pragma unroll
    for (int n = 1; n < kMaxPoints; ++n) if (n < N) y = y + coef[n] * raw[n][i];
    o[lane + 64 * i] = started ? y : 0.0f;
  }
}

}  // namespace

// rows x dim spherical means: out[row] = mean over points table[speaker[n]][row] with weights w[n]
bool spherical_mean_rows(const float* d_table, size_t speaker_stride, int rows, int dim, int n_active, const int* speakers,
                         const float* weights, float* d_out, hipStream_t stream) {
  if (n_active < 0 || n_active > kMaxPoints || (dim != 128 && dim != 256)) return false;
  SphMeanArgs a{};
  a.table = d_table; a.out = d_out; a.speaker_stride = speaker_stride; a.n_active = n_active;
  for (int n = 0; n < n_active; ++n) { a.speaker[n] = speakers[n]; a.weight[n] = weights[n]; }
  if (dim == 128) hipLaunchKernelGGL(sph_mean_kernel<2>, dim3(rows), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL(sph_mean_kernel<4>, dim3(rows), dim3(64), 0, stream, a);
  return hip_ok(hipGetLastError(), "spherical mean launch");
}

}  // namespace bhip
