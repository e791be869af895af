// spherical_mean.h -- weighted mean on the unit sphere used by speaker morphing: this project's
// counterpart of the reference's SphericalAverage<float, M> (reference
// src/common/spherical_average.h:80-444; Buss-Fillmore iteration preconditioned with a two-slot
// L-BFGS memory).  Same sequence of float operations (sequential dot products, acosf / sinf / sqrtf),
// so results equal the reference's bit for bit (tests/test_morph.py checks against a library built
// from the reference header).  Used by ProcessorCore2's morph branch (processor_core.cc).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <limits>
#include <vector>

namespace beatrice_amd {

class SphericalMean {
 public:
  // points: n_all rows of `dim` floats (any norm); at most `limit` of them (0 = all) take part in one mean
  void Initialize(std::size_t n_all, std::size_t dim, const float* points, std::size_t limit = 0, std::size_t memory = 2) {
    n_all_ = n_all; dim_ = dim; mem_ = memory;
    limit_ = (limit == 0 || limit > n_all) ? n_all : limit;
    active_ = 0;
    index_.assign(limit_, 0);
    w_.assign(limit_, 0.0f);
    coef_.assign(limit_, 0.0f);
    raw_.assign(points, points + n_all * dim);
    unit_ = raw_;
    for (std::size_t n = 0; n < n_all; ++n) Normalize(&unit_[n * dim]);
    q_.assign(dim, 0.0f); g_.assign(dim, 0.0f); d_.assign(dim, 0.0f);
    s_.assign(mem_ * dim, 0.0f); t_.assign(mem_ * dim, 0.0f);
    rho_.assign(mem_, 0.0f); alpha_.assign(mem_, 0.0f);
    converged_ = true;
  }

  // weights indexed by point; `order` (optional) = point indices sorted by descending weight
  void SetWeights(std::size_t n_points, const float* weights, const int* order = nullptr) {
    converged_ = false;
    std::fill(coef_.begin(), coef_.end(), 0.0f);
    std::fill(w_.begin(), w_.end(), 0.0f);
    if (order) {
      active_ = std::min(n_points, limit_);
      for (std::size_t i = 0; i < active_; ++i) {
        index_[i] = static_cast<std::size_t>(order[i]);
        w_[i] = weights[index_[i]];
        if (w_[i] == 0.0f) { active_ = i; break; }
      }
    } else {
      active_ = 0;
      for (std::size_t i = 0; i < n_points; ++i) {
        if (weights[i] > 0.0f) {
          index_[active_] = i; w_[active_] = weights[i];
          if (++active_ >= limit_) break;
        }
      }
    }
    bool started = false;
    if (active_ > 0) {
      float sum = 0.0f;
      for (std::size_t i = 0; i < active_; ++i) sum += w_[i];
      if (sum > 0.0f) {
        const float inv = 1.0f / sum;
        for (std::size_t i = 0; i < active_; ++i) w_[i] *= inv;
        const float* p0 = &unit_[index_[0] * dim_];
        for (std::size_t l = 0; l < dim_; ++l) q_[l] = w_[0] * p0[l];
        for (std::size_t n = 1; n < active_; ++n) Axpy(w_[n], &unit_[index_[n] * dim_], q_.data());
        started = Normalize(q_.data());
      }
    }
    if (!started) { converged_ = true; return; }
    slot_ = 0;
    gamma_ = 1.0f;
    std::fill(s_.begin(), s_.end(), 0.0f); std::fill(t_.begin(), t_.end(), 0.0f);
    std::fill(rho_.begin(), rho_.end(), 0.0f); std::fill(alpha_.begin(), alpha_.end(), 0.0f);
    Gradient();
  }

  // one iteration; true once converged
  bool Update() {
    if (converged_) return true;
    const float step = std::sqrt(Dot(d_.data(), d_.data()));
    if (step >= 8 * std::numeric_limits<float>::epsilon()) {
      // move q along -d, remember the step
      float* s = &s_[slot_ * dim_];
      std::copy(q_.begin(), q_.end(), s);
      for (std::size_t l = 0; l < dim_; ++l) q_[l] -= d_[l];
      Normalize(q_.data());
      for (std::size_t l = 0; l < dim_; ++l) s[l] = q_[l] - s[l];
      // new gradient, remember its change
      float* t = &t_[slot_ * dim_];
      std::copy(g_.begin(), g_.end(), t);
      Gradient();
      for (std::size_t l = 0; l < dim_; ++l) t[l] = g_[l] - t[l];
      Project(q_.data(), t);
      // curvature estimates
      gamma_ = Dot(s, t);
      rho_[slot_] = 1.0f / gamma_;
      gamma_ /= Dot(t, t);
      if (++slot_ >= mem_) slot_ = 0;
    } else {
      converged_ = true;
    }
    return converged_;
  }

  // un-normalised combination of the ORIGINAL points with the converged coefficients
  void Result(float* out) const {
    if (index_.empty()) { for (std::size_t l = 0; l < dim_; ++l) out[l] = 0.0f; return; }   // no points at all: nothing to combine
    const float* p0 = &raw_[index_[0] * dim_];
    for (std::size_t l = 0; l < dim_; ++l) out[l] = coef_[0] * p0[l];
    for (std::size_t n = 1; n < active_; ++n) Axpy(coef_[n], &raw_[index_[n] * dim_], out);
  }

 private:
  float Dot(const float* a, const float* b) const {
    float y = 0.0f;
    for (std::size_t l = 0; l < dim_; ++l) y += a[l] * b[l];
    return y;
  }
  void Axpy(float a, const float* x, float* y) const {
    for (std::size_t l = 0; l < dim_; ++l) y[l] += a * x[l];
  }
  bool Normalize(float* x) const {
    const float norm = std::sqrt(Dot(x, x));
    if (!(norm > 0.0f)) return false;
    const float inv = 1.0f / norm;
    for (std::size_t l = 0; l < dim_; ++l) x[l] *= inv;
    return true;
  }
  void Project(const float* axis, float* y) const { Axpy(-Dot(axis, y), axis, y); }
  static float Sinc(float x) {
    static const float e0 = std::numeric_limits<float>::epsilon();
    static const float e1 = std::sqrt(e0);
    static const float e2 = std::sqrt(e1);
    // Behavioural note: in the reference, `abs(x)` on a float binds to `int abs(int)` (only <cmath> is
    // included and the call is unqualified), so the magnitude is TRUNCATED to an integer before the
    // threshold tests: Sinc(x) is exactly 1 for |x| < 1 and sin(x)/x (evaluated in double) otherwise.
    // Reproduced as is -- morph results must match the reference's, quirk included.
    const float ax = static_cast<float>(std::abs(static_cast<int>(x)));
    if (ax >= e2) return static_cast<float>(std::sin(static_cast<double>(x)) / static_cast<double>(x));  // ::sin(double)
    float y = 1.0f;
    if (ax >= e0) {
      const float x2 = x * x;
      y -= x2 / 6.0f;
      if (ax >= e1) y += x2 * x2 / 120.0f;
    }
    return y;
  }
  // coefficients v, tangent gradient g and preconditioned direction d at the current q
  void Gradient() {
    const float eps = std::numeric_limits<float>::epsilon();
    float denom = 0.0f;
    std::fill(g_.begin(), g_.end(), 0.0f);
    for (std::size_t n = 0; n < active_; ++n) {
      const float* p = &unit_[index_[n] * dim_];
      const float c = std::clamp(Dot(p, q_.data()), -1.0f, 1.0f);
      const float theta = static_cast<float>(std::acos(static_cast<double>(c)));  // the reference calls ::acos(double)
      const float inv_sinc = 1.0f / (Sinc(theta) + eps);
      denom += w_[n] * c * inv_sinc;
      coef_[n] = w_[n] * inv_sinc;
      Axpy(-2.0f * coef_[n], p, g_.data());
    }
    const float inv_denom = 1.0f / (denom + eps);
    for (std::size_t n = 0; n < active_; ++n) coef_[n] *= inv_denom;
    Project(q_.data(), g_.data());
    d_ = g_;
    for (std::size_t k = 0; k < mem_; ++k) {  // two-loop recursion over the memory slots
      const std::size_t i = (slot_ - k - 1 + mem_) % mem_;
      alpha_[i] = rho_[i] * Dot(&s_[i * dim_], d_.data());
      Axpy(-alpha_[i], &t_[i * dim_], d_.data());
    }
    for (std::size_t l = 0; l < dim_; ++l) d_[l] *= gamma_;
    for (std::size_t k = 0; k < mem_; ++k) {
      const std::size_t i = (slot_ + k) % mem_;
      const float beta = rho_[i] * Dot(&t_[i * dim_], d_.data());
      Axpy(alpha_[i] - beta, &s_[i * dim_], d_.data());
    }
  }

  std::size_t n_all_ = 0, dim_ = 0, limit_ = 0, active_ = 0, mem_ = 2, slot_ = 0;
  bool converged_ = true;
  float gamma_ = 1.0f;
  std::vector<std::size_t> index_;
  std::vector<float> w_, coef_, raw_, unit_, q_, g_, d_, s_, t_, rho_, alpha_;
};

}  // namespace beatrice_amd
