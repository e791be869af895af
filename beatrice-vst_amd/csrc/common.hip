// common.hip -- device memory plumbing shared by the modules: parameter blobs, ring arenas,
// weight-pointer binding (tensor order of MODEL_SPEC section 5) and the set-time launchers.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "conv_gemm.hip.h"
#include "engine.h"

namespace bhip {

bool hip_ok(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  static const bool verbose = std::getenv("BEATRICE_HIP_DEBUG") != nullptr;
  if (verbose) std::fprintf(stderr, "[beatrice_hip] %s -> %s\n", what, hipGetErrorString(e));
  (void)hipGetLastError();
  return false;
}

LaunchHook*& launch_hook() {
  static thread_local LaunchHook* hook = nullptr;
  return hook;
}

bool DeviceBlob::upload(const float* host, size_t n) {
  release();
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), n * sizeof(float)));
  n_floats = n;
  BHIP_TRY(hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
  BHIP_TRY(hipDeviceSynchronize());
  return true;
}
void DeviceBlob::release() {
  if (d) (void)hipFree(d);
  d = nullptr;
  n_floats = 0;
}

bool RingArena::build(int B_, const std::vector<RingSpec>& specs) {
  release();
  B = B_;
  size_t total = 0;
  for (const RingSpec& s : specs) {
    if (s.m < 1 || B_HOP_WRAP % s.m != 0) return false;  // the step counter wraps at B_HOP_WRAP = lcm(1..17): every slot count must divide it
    if ((size_t)B * s.C * s.n * s.m >= ((size_t)1 << 32)) return false;  // ring_frame indexes a ring with 32 bits (ring.h)
    total += ((size_t)B * s.C * s.n * s.m + 63) / 64 * 64;
  }
  BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&base), total * sizeof(float)));
  floats = total;
  BHIP_TRY(hipMemset(base, 0, total * sizeof(float)));
  size_t off = 0;
  for (const RingSpec& s : specs) {
    s.ring->base = base + off;
    s.ring->C = s.C;
    s.ring->n = s.n;
    s.ring->m = s.m;
    off += ((size_t)B * s.C * s.n * s.m + 63) / 64 * 64;
    rings.push_back(s.ring);
  }
  return true;
}
bool team_capacity_ok(const void* kernel, int n_workgroups, int threads, size_t dynamic_lds_bytes) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop{};
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, dynamic_lds_bytes) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
      hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return (long long)per_cu * prop.multiProcessorCount >= n_workgroups;
}
bool RingArena::zero_all(hipStream_t s) const {
  BHIP_TRY(hipMemsetAsync(base, 0, floats * sizeof(float), s));
  return true;
}
bool RingArena::zero_stream(int b, hipStream_t s) const {
  for (const Ring* r : rings) {
    const size_t per = ring_stream_floats(*r);
    BHIP_TRY(hipMemsetAsync(r->base + (size_t)b * per, 0, per * sizeof(float), s));
  }
  return true;
}
void RingArena::release() {
  if (base) (void)hipFree(base);
  base = nullptr;
  floats = 0;
  rings.clear();
}

// ---- weight binding: the tensor order of MODEL_SPEC section 5 --------------------------------
static const int kPhoneF[5][4] = {{1, 64, 10, 5}, {64, 128, 8, 4}, {128, 256, 4, 2}, {256, 256, 4, 2}, {256, 256, 4, 2}};

size_t PhoneWeights::n_floats(int out_ch) {
  size_t n = 0;
  for (auto& f : kPhoneF) n += (size_t)f[0] * f[2] * f[1] + f[1];
  n += 4 * (5 * 256 * 256 + 256);
  n += 2 * 256 * 768 + 2 * 768;
  n += 256 * out_ch + out_ch;
  return n;
}
void PhoneWeights::bind(const float* p, int out_ch) {
  f1_w = p; p += 10 * 64;
  f1_b = p; p += 64;
  for (int i = 1; i < 5; ++i) {
    f_w[i - 1] = p; p += (size_t)kPhoneF[i][0] * kPhoneF[i][2] * kPhoneF[i][1];
    f_b[i - 1] = p; p += kPhoneF[i][1];
  }
  for (int i = 0; i < 4; ++i) { rb_w[i] = p; p += 5 * 256 * 256; rb_b[i] = p; p += 256; }
  gru_wih = p; p += 256 * 768;
  gru_whh = p; p += 256 * 768;
  gru_bih = p; p += 768;
  gru_bhh = p; p += 768;
  out_w = p; p += 256 * out_ch;
  out_b = p; p += out_ch;
}

size_t PitchWeights::n_floats(int bins) {
  size_t n = 2 * B_FFT_N;
  n += 3 * B_SPEC_BINS * 128 + 128 + 2 * (3 * 128 * 128 + 128);
  n += 2 * 128 * 384 + 2 * 384;
  n += 128 * bins + bins + 128 + 1;
  return n;
}
void PitchWeights::bind(const float* p, int bins) {
  window = p; p += B_FFT_N;
  twiddle = p; p += B_FFT_N;
  for (int i = 0; i < 3; ++i) {
    const int cin = i == 0 ? B_SPEC_BINS : 128;
    p_w[i] = p; p += 3 * cin * 128;
    p_b[i] = p; p += 128;
  }
  gru_wih = p; p += 128 * 384;
  gru_whh = p; p += 128 * 384;
  gru_bih = p; p += 384;
  gru_bhh = p; p += 384;
  out_w = p; p += 128 * bins;
  out_b = p; p += bins;
  voi_w = p; p += 128;
  voi_b = p; p += 1;
}

size_t EmbedWeights::n_floats() { return 2 * (B_HID * B_HID + B_HID) + B_NBLOCKS * 2 * (B_KV_CH * B_HID + B_HID); }
void EmbedWeights::bind(const float* p) {
  add_w = p; p += B_HID * B_HID; add_b = p; p += B_HID;
  frm_w = p; p += B_HID * B_HID; frm_b = p; p += B_HID;
  for (int b = 0; b < B_NBLOCKS; ++b) {
    k_w[b] = p; p += B_KV_CH * B_HID; k_b[b] = p; p += B_HID;
    v_w[b] = p; p += B_KV_CH * B_HID; v_b[b] = p; p += B_HID;
  }
}

static const int kUpRate[4] = {5, 4, 4, 3};
static const int kUpCh[5] = {256, 128, 64, 32, 16};
size_t WaveWeights::n_floats(bool legacy) {
  const size_t phone_ch = legacy ? 256 : B_PHONE_CH, bins = legacy ? 384 : B_PITCH_BINS;
  size_t n = phone_ch * B_HID + B_HID + bins * B_HID + 4 * B_HID;
  n += B_NBLOCKS * ((3 * B_HID * B_HID + B_HID) + (legacy ? 1 : 3) * (B_HID * B_HID + B_HID));
  for (int s = 0; s < 4; ++s) {
    const size_t cin = kUpCh[s], cout = kUpCh[s + 1], r = kUpRate[s];
    n += 2 * cin * r * cout + r * cout + 2 * (3 * cout * cout + cout);
  }
  n += 7 * 16 + 1;
  return n;
}
void WaveWeights::bind(const float* p, bool legacy) {
  inp_w = p; p += (legacy ? 256 : B_PHONE_CH) * B_HID;
  inp_b = p; p += B_HID;
  pitch_emb = p; p += (legacy ? 384 : B_PITCH_BINS) * B_HID;
  feat_w = p; p += 4 * B_HID;
  for (int b = 0; b < B_NBLOCKS; ++b) {
    c1_w[b] = p; p += 3 * B_HID * B_HID; c1_b[b] = p; p += B_HID;
    c2_w[b] = p; p += B_HID * B_HID; c2_b[b] = p; p += B_HID;
    if (legacy) { q_w[b] = q_b[b] = o_w[b] = o_b[b] = nullptr; continue; }
    q_w[b] = p; p += B_HID * B_HID; q_b[b] = p; p += B_HID;
    o_w[b] = p; p += B_HID * B_HID; o_b[b] = p; p += B_HID;
  }
  for (int s = 0; s < 4; ++s) {
    const size_t cin = kUpCh[s], cout = kUpCh[s + 1], r = kUpRate[s];
    up_w[s] = p; p += 2 * cin * r * cout; up_b[s] = p; p += r * cout;
    ra_w[s] = p; p += 3 * cout * cout; ra_b[s] = p; p += cout;
    rb_w[s] = p; p += 3 * cout * cout; rb_b[s] = p; p += cout;
  }
  fin_w = p; p += 7 * 16;
  fin_b = p; p += 1;
}

// ---- host-side repack of GEMM weights into MFMA B-fragment order (conv_gemm.hip.h) -----------
static void pack_kn(const float* w_const, int K, int N) {
  float* w = const_cast<float*>(w_const);
  std::vector<float> tmp((size_t)K * N);
  for (int kk = 0; kk < K; ++kk)
    for (int n = 0; n < N; ++n) tmp[packed_w_offset(K, kk, n)] = w[(size_t)kk * N + n];
  std::copy(tmp.begin(), tmp.end(), w);
}
void PhoneWeights::pack_host(float* base, int out_ch) {
  PhoneWeights w{};
  w.bind(base, out_ch);
  for (int i = 1; i < 5; ++i) pack_kn(w.f_w[i - 1], kPhoneF[i][0] * kPhoneF[i][2], kPhoneF[i][1]);
  for (int i = 0; i < 4; ++i) pack_kn(w.rb_w[i], 5 * 256, 256);
  pack_kn(w.gru_wih, 256, 768);
  pack_kn(w.gru_whh, 256, 768);
  pack_kn(w.out_w, 256, out_ch);
}
void PitchWeights::pack_host(float* base, int bins) {
  PitchWeights w{};
  w.bind(base, bins);
  pack_kn(w.p_w[0], 3 * B_SPEC_BINS, 128);
  pack_kn(w.p_w[1], 3 * 128, 128);
  pack_kn(w.p_w[2], 3 * 128, 128);
  pack_kn(w.gru_wih, 128, 384);
  pack_kn(w.gru_whh, 128, 384);
  pack_kn(w.out_w, 128, bins);
}
void WaveWeights::pack_host(float* base, bool legacy) {
  WaveWeights w{};
  w.bind(base, legacy);
  pack_kn(w.inp_w, legacy ? 256 : B_PHONE_CH, B_HID);
  for (int b = 0; b < B_NBLOCKS; ++b) {
    pack_kn(w.c1_w[b], 3 * B_HID, B_HID);
    pack_kn(w.c2_w[b], B_HID, B_HID);
    if (legacy) continue;
    pack_kn(w.q_w[b], B_HID, B_HID);
    pack_kn(w.o_w[b], B_HID, B_HID);
  }
  pack_kn(w.up_w[0], 2 * 256, 5 * 128);
  pack_kn(w.ra_w[0], 3 * 128, 128);
  pack_kn(w.rb_w[0], 3 * 128, 128);
  pack_kn(w.up_w[1], 2 * 128, 4 * 64);
  // fused tail (wave_tail.hip.h): res2a/b, up3, res3a/b, up4, res4a/b
  pack_kn(w.ra_w[1], 3 * 64, 64);
  pack_kn(w.rb_w[1], 3 * 64, 64);
  pack_kn(w.up_w[2], 2 * 64, 4 * 32);
  pack_kn(w.ra_w[2], 3 * 32, 32);
  pack_kn(w.rb_w[2], 3 * 32, 32);
  pack_kn(w.up_w[3], 2 * 32, 3 * 16);
  pack_kn(w.ra_w[3], 3 * 16, 16);
  pack_kn(w.rb_w[3], 3 * 16, 16);
}

// ---- set-time launchers ------------------------------------------------------------------------
void embed_project_rows(const float* w, const float* b, const float* d_x, float* d_y, int rows, hipStream_t stream) {
  const int total = rows * B_HID;
  hipLaunchKernelGGL(dense_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, d_x, w, b, d_y, rows, B_HID, B_HID);
}
void embed_project_kv(const EmbedWeights& w, int block, const float* d_kv_raw, int slots, float* d_kt, float* d_v,
                      hipStream_t stream, float* d_kt_plain, float* d_v_plain) {
  hipLaunchKernelGGL(kv_project_kernel, dim3(B_KV_LEN, slots), dim3(256), 0, stream, d_kv_raw, w.k_w[block], w.k_b[block],
                     w.v_w[block], w.v_b[block], d_kt, d_v, d_kt_plain, d_v_plain);
}
void codebook_prepare(const float* d_cb, int n, float* d_cbT, float* d_cnorm, hipStream_t stream) {
  hipLaunchKernelGGL(codebook_prep_kernel, dim3(n), dim3(512), 0, stream, d_cb, d_cbT, d_cnorm);
}

}  // namespace bhip
