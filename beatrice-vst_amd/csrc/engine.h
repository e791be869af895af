// engine.h -- host-side model/state objects of libbeatrice_hip.
//
// Three per-hop modules mirror the reference's three per-hop ABI calls
// (reference lib/beatricelib/beatrice.h:243-247, 266-271, 301-307).  Each module has
//   *Weights : device pointers into the uploaded parameter blob (immutable, shareable),
//   *State   : per-stream activation rings + per-stream settings for B streams,
//   *_forward: enqueue the module's kernels on a HIP stream (no allocation, no sync -- the
//              sequence is capturable into a hipGraph).
// The 1-stream C-ABI (abi.hip) wraps these with B = 1; the batched ABI (batch.hip) with B = N.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

#include "kernels_misc.hip.h"
#include "meas_env.h"
#include "ring.h"

namespace bhip {

// ---- error handling: never throw across the C-ABI; remember the first failure ---------------
bool hip_ok(hipError_t e, const char* what);  // logs once per site when BEATRICE_HIP_DEBUG is set
#define BHIP_TRY(expr)                                \
  do {                                                \
    if (!::bhip::hip_ok((expr), #expr)) return false; \
  } while (0)

// ---- launch hook: every kernel launch of the per-hop chain goes through launch_site(), so a
// profiler (batch.hip, BeatriceBatch_ProfileKernels) can bracket each launch with HIP events and
// attach the launch's algorithmic FLOPs / bytes.  No hook = plain launch.
struct LaunchInfo { const char* name; double flops; double bytes; };
struct LaunchHook {
  virtual ~LaunchHook() = default;
  virtual void on_launch(const LaunchInfo& info, hipStream_t stream, void (*thunk)(void*), void* ctx) = 0;
};
LaunchHook*& launch_hook();  // thread-local
template <class F>
inline void launch_site(const LaunchInfo& info, hipStream_t stream, F&& fn) {
  LaunchHook* h = launch_hook();
  if (!h) { fn(); return; }
  h->on_launch(info, stream, [](void* c) { (*static_cast<F*>(c))(); }, &fn);
}

// ---- device blob -----------------------------------------------------------------------------
struct DeviceBlob {
  float* d = nullptr;
  size_t n_floats = 0;
  bool upload(const float* host, size_t n);
  void release();
};

// ---- ring arena --------------------------------------------------------------------------------
struct RingSpec { Ring* ring; int C, n, m; };
struct RingArena {
  float* base = nullptr;
  size_t floats = 0;
  int B = 0;
  std::vector<Ring*> rings;
  bool build(int B, const std::vector<RingSpec>& specs);  // one hipMalloc, zero-filled
  bool zero_all(hipStream_t s) const;
  bool zero_stream(int b, hipStream_t s) const;
  void release();
};

// words behind a module-owned input buffer that travel with the 1-stream ABI's input copy (abi.hip):
//   phone: [0] step counter, [1] k-NN k, [2..3] codebook^T pointer, [4..5] norms pointer
//   pitch: [0] step counter, [1] lowest bin, [2] highest bin
constexpr int kMailboxWords = 8;

// ---- team launches (team.hip.h): every workgroup of the launch spin-waits on granules other workgroups of the SAME launch write,
// so all of them must be resident at once.  (a) At create time a context only gets a team launch when the device can hold the whole
// team (occupancy x hipDeviceProp's compute-unit count >= workgroups: this catches a PARTITION of the GPU, whose device reports fewer
// compute units; it does NOT see CU masks -- HSA_CU_MASK, a stream's CU mask -- or compute units held by other processes).
// (b) Residency can therefore still fail at run time -- a masked GPU, other processes' kernels between the team's workgroups --: the
// first call then spins until the bounded waits give up (~0.3-1 s, once), and the bounded spins then give up and raise the context's pinned
// flag: team_recover() clears it, zeroes the granules and the context's rings (the stream restarts from silence, like a new
// context) and switches the context to the per-layer launches for good; the call that hit the timeout returns zeros (the ABI's
// answer to any internal failure), the following ones work.
bool team_capacity_ok(const void* kernel, int n_workgroups, int threads, size_t dynamic_lds_bytes);
template <class State>
bool team_timed_out(const State& st) { return st.d_team_dead != nullptr && *st.d_team_dead != 0; }
template <class State>
bool team_recover(State& st, hipStream_t s) {   // (call with the stream idle)
  if (!team_timed_out(st)) return false;
  *st.d_team_dead = 0;
  st.team_off = true;
  if (st.d_team_xb && st.team_granules) (void)hipMemsetAsync(st.d_team_xb, 0, sizeof(unsigned long long) * st.team_granules, s);
  (void)st.arena.zero_all(s);
  (void)hipStreamSynchronize(s);
  return true;
}

// ---- phone extractor ---------------------------------------------------------------------------
struct PhoneWeights {
  const float *f1_w, *f1_b;
  const float *f_w[4], *f_b[4];    // F2..F5
  const float *rb_w[4], *rb_b[4];  // residual blocks
  const float *gru_wih, *gru_whh, *gru_bih, *gru_bhh, *out_w, *out_b;
  // out_ch: width of the phone vector -- 128 (rc.0) or 256 (the legacy generations, MODEL_SPEC 6.1)
  static size_t n_floats(int out_ch = B_PHONE_CH);
  void bind(const float* base, int out_ch = B_PHONE_CH);
  static void pack_host(float* base, int out_ch = B_PHONE_CH);  // GEMM tensors -> MFMA-fragment order (conv_gemm.hip.h), in place
};
struct PhoneWeightsLegacy : PhoneWeights {   // the same tensors with a 256-wide output layer
  static size_t n_floats() { return PhoneWeights::n_floats(256); }
  void bind(const float* base) { PhoneWeights::bind(base, 256); }
  static void pack_host(float* base) { PhoneWeights::pack_host(base, 256); }
};
struct PhoneState {
  int B = 0;
  int H = 1;  // hops per step (1 = real-time per-hop path; 2/4 = block mode for bulk conversion)
  RingArena arena;
  Ring audio, f[5], rb[4], h, raw;
  float* d_in = nullptr;     // [B][H*160]; owned unless shared
  bool owns_in = false;
  int* hop_mailbox = nullptr;  // owned d_in only: kMailboxWords words right behind the audio (sent with the input copy)
  size_t io_stride = 0;        // batch, resident I/O: d_in holds several steps, this many floats apart (slot = hop[1])
  float* d_phone = nullptr;  // ring [B][out_slots * H][128]: step t writes slot t mod out_slots
  int out_slots = 1;         // 3 in a batch, so that the next steps' front end may run while the waveform generator reads
  const float** d_cbT = nullptr;    // [B] device pointers
  const float** d_cnorm = nullptr;  // [B]
  int* d_vqk = nullptr;             // [B]
  int* d_hop = nullptr;       // owned hop counter
  int* hop = nullptr;         // counter the kernels read (== d_hop unless shared by a batch)
  int* hop_in = nullptr;      // counter the FIRST kernel reads (== hop unless a batch double-buffers the counter)
  int* hop_publish = nullptr; // batch: the first kernel copies *hop_in here (== hop) for the rest of the chain
  int* hop_publish_wave = nullptr;  // batch: ... and to the waveform generator's counter pair [counter & 1]
  bool advance_hop = true;    // this module's forward ends with the counter increment
  bool skip_vq = false;       // no stream uses the codebook: phone.out writes d_phone, no k-NN launch
  void (*after_convs)(void*) = nullptr;   // 1-stream ABI: called once the convolutions (the team launch) are enqueued, before the GRU cell's launch --
  void* after_convs_arg = nullptr;        // where abi.hip enqueues the partner pitch context's hop (its launches go out while the team launch runs)
  int out_ch = B_PHONE_CH;    // width of the phone vector (256: legacy generations, which have no codebook step)
  // pipe_slack: one more step slot on every ring a later layer reads, so that each LAYER may run as its own pipeline
  // stage one step behind its producer (batch.hip, tick mode)
  unsigned long long* d_team_xb = nullptr;   // one stream, one hop per call: the eight convolutions as one team launch (team.hip.h)
  int* d_team_dead = nullptr;   // pinned host word: set by a team launch that gave a wait up (the call then returns zeros)
  size_t team_granules = 0;
  bool team_off = false;        // set by team_recover(): this context runs the per-layer launches from then on
  bool create(int B, int H, float* shared_in, int out_slots = 1, bool pipe_slack = false, int out_ch = B_PHONE_CH);
  void destroy();
};
void phone_forward(const PhoneWeights& w, const PhoneState& s, hipStream_t stream);

// ---- pitch estimator ---------------------------------------------------------------------------
struct PitchWeights {
  const float *window, *twiddle;
  const float *p_w[3], *p_b[3];
  const float *gru_wih, *gru_whh, *gru_bih, *gru_bhh, *out_w, *out_b, *voi_w, *voi_b;
  // bins: pitch classes -- 448 (rc.0) or 384 (the legacy generations, MODEL_SPEC 6.2)
  static size_t n_floats(int bins = B_PITCH_BINS);
  void bind(const float* base, int bins = B_PITCH_BINS);
  static void pack_host(float* base, int bins = B_PITCH_BINS);
};
struct PitchWeightsLegacy : PitchWeights {
  static size_t n_floats() { return PitchWeights::n_floats(384); }
  void bind(const float* base) { PitchWeights::bind(base, 384); }
  static void pack_host(float* base) { PitchWeights::pack_host(base, 384); }
};
struct PitchState {
  int B = 0;
  int H = 1;  // hops per step
  RingArena arena;
  Ring audio, spec, p[3], h, logits;
  float* d_in = nullptr;
  bool owns_in = false;
  int* hop_mailbox = nullptr;  // owned d_in only: kMailboxWords words right behind the audio
  size_t io_stride = 0;        // see PhoneState
  int *d_min_q = nullptr, *d_max_q = nullptr, *d_prev_q = nullptr;  // [B]
  int q_slots = 1;                                                 // step slots of the three outputs below (2 with pipe_slack)
  int *d_q_raw = nullptr, *d_q = nullptr;                          // [q_slots][B][H]
  float* d_feat = nullptr;             // [q_slots][B][H][4]
  PitchParams* d_params = nullptr;     // [B] or nullptr (1-stream ABI: host does the transform)
  int* d_hop = nullptr;       // owned hop counter
  int* hop = nullptr;         // counter the kernels read (== d_hop unless shared by a batch)
  int* hop_in = nullptr;      // counter the first kernel (FFT) reads
  bool advance_hop = true;    // this module's forward ends with the counter increment
  int bins = B_PITCH_BINS;    // pitch classes (384: legacy generations)
  bool q_raw_in_feat = false;   // d_q_raw = d_feat + 4 (create())
  float* h_result = nullptr;    // 1-stream ABI: pinned host block the head kernel writes [4 feat | raw bin | sequence word] into itself (kernels_misc.hip.h PitchHeadArgs::host_out)
  unsigned long long* d_team_xb = nullptr;   // (see PhoneState)
  int* d_team_dead = nullptr;   // pinned host word: set by a team launch that gave a wait up (the call then returns zeros)
  size_t team_granules = 0;
  bool team_off = false;
  bool create(int B, int H, float* shared_in, bool with_params, bool pipe_slack = false, int bins = B_PITCH_BINS);
  void destroy();
};
void pitch_forward(const PitchWeights& w, const PitchState& s, hipStream_t stream);

// ---- embedding setter --------------------------------------------------------------------------
struct EmbedWeights {
  const float *add_w, *add_b, *frm_w, *frm_b;
  const float *k_w[B_NBLOCKS], *k_b[B_NBLOCKS], *v_w[B_NBLOCKS], *v_b[B_NBLOCKS];
  static size_t n_floats();
  void bind(const float* base);
  static void pack_host(float*) {}  // set-time kernels read plain [K][N]
};

// ---- waveform generator ------------------------------------------------------------------------
struct WaveWeights {
  const float *inp_w, *inp_b, *pitch_emb, *feat_w;
  const float *c1_w[B_NBLOCKS], *c1_b[B_NBLOCKS], *c2_w[B_NBLOCKS], *c2_b[B_NBLOCKS];
  const float *q_w[B_NBLOCKS], *q_b[B_NBLOCKS], *o_w[B_NBLOCKS], *o_b[B_NBLOCKS];
  const float *up_w[4], *up_b[4], *ra_w[4], *ra_b[4], *rb_w[4], *rb_b[4];
  const float *fin_w, *fin_b;
  // legacy (MODEL_SPEC 6.3): phone vector 256 wide, 384 pitch embeddings, blocks without the attention half (no q / o tensors)
  static size_t n_floats(bool legacy = false);
  void bind(const float* base, bool legacy = false);
  static void pack_host(float* base, bool legacy = false);  // every MFMA layer (conv_gemm and the fused tail); fin_w stays plain
};
struct WaveWeightsLegacy : WaveWeights {
  static size_t n_floats() { return WaveWeights::n_floats(true); }
  void bind(const float* base) { WaveWeights::bind(base, true); }
  static void pack_host(float* base) { WaveWeights::pack_host(base, true); }
};
struct WaveState {
  int B = 0;
  int H = 1;            // hops per step; attention rows = (stream, hop in step)
  int n_slots = 0;      // K/V slots per block in the tables
  int n_tiles_max = 0;  // attention M-tiles (16 rows each) upper bound
  RingArena arena;
  Ring e, x[B_NBLOCKS + 1];
  // scratch of one conditioned block (reused by every block); one set per pipeline stage so that blocks of
  // different steps can be in flight at once (batch.hip)
  static constexpr int kScratchSets = 4;
  struct Scratch { Ring h1, xa, q, sc, o; } scr[kScratchSets];
  int boundary_slots = 0;   // extra step slots on the rings a pipeline stage boundary may cut (x[], ya2): 0, or 2 in a batch
                            // (rounded up to a slot count that divides the step counter's wrap)
  Ring ya1, yb1, yc1, ya2;  // upsampler stage 1 and the stage-2 transposed conv output
  Ring tail;                // per-stream history block of the fused upsampler tail (wave_tail.hip.h)
  Ring ya3, ya4;            // batch with pipeline slack only: the frames between the three tail stages of the tick launch
                            // (tail_stages.hip.h): 32 ch x 80 and 16 ch x 240 frames per step, two step slots each
  // inputs (device): phone [B][H][128], q [B][H], feat [B][H][4]; owned unless shared with other modules
  float* d_phone = nullptr; int* d_q = nullptr; float* d_feat = nullptr;
  int q_slots = 1;  // step slots of d_q / d_feat (PitchState::q_slots when shared)
  bool owns_inputs = false;
  float* d_out = nullptr;  // [B][H*240]
  int* h_flag = nullptr;         // 1-stream ABI: d_out is a pinned host block and the tail kernel writes *d_seq here once the samples are (wave_tail.hip.h TailArgs::host_flag)
  const int* d_seq = nullptr;
  // conditioning tables and per-stream selectors
  float* d_add_tab = nullptr; int n_add = 0;  // [n_add][256] projected additive embeddings
  float* d_frm_tab = nullptr; int n_frm = 0;  // [n_frm][256] projected formant embeddings
  int *d_add_idx = nullptr, *d_frm_idx = nullptr;  // [B]
  float* d_kt[B_NBLOCKS] = {nullptr, nullptr, nullptr, nullptr};  // [n_slots][256][384]
  float* d_v[B_NBLOCKS] = {nullptr, nullptr, nullptr, nullptr};   // [n_slots][384][256]
  // the same tables in plain row-major order (K^T [256][384], V [384][256]) for the 4x4x1 multi-block MFMAs of rows whose
  // neighbours attend to other slots (rowchain.hip.h, quad path); tick-mode batches only, null elsewhere
  float* d_ktp[B_NBLOCKS] = {nullptr, nullptr, nullptr, nullptr};
  float* d_vp[B_NBLOCKS] = {nullptr, nullptr, nullptr, nullptr};
  int* d_perm[B_NBLOCKS] = {nullptr, nullptr, nullptr, nullptr};       // [n_tiles_max][16]
  int* d_tile_slot[B_NBLOCKS] = {nullptr, nullptr, nullptr, nullptr};  // [n_tiles_max]
  int* d_hop = nullptr;       // owned hop counter
  int* hop = nullptr;         // counter the kernels read (== d_hop unless shared by a batch)
  size_t io_stride = 0;         // batch, resident I/O: d_out holds io_slots steps, this many floats apart
  int io_slots = 0;
  // the conditioning mix (wave.cond) belongs to the front end of a step: it reads the front end's counter and,
  // in a batch, stores the next step's {counter, I/O slot} (no front-end kernel reads that pair after the first launch)
  const int* front_hop = nullptr;  // nullptr: same counter as the rest of the module
  int* front_next_out = nullptr;
  int front_slots = 1;             // step slots of the front end's outputs (phone vector, conditioning e): 1, or 3 in a batch
  bool advance_hop = true;    // this module's forward ends with the counter increment
  // legacy generations (MODEL_SPEC 6.3, 1-stream ABI only): the phone vector is 256 wide, the pitch embedding has 384 rows, the
  // conditioning vector is ONE additive row handed over per hop (n_frm = 0: no formant table), blocks are c1 + c2 only
  bool legacy = false;
  // one stream, one hop per call (the 1-stream C-ABI): the layers from the input mix to the stage-2 transposed conv run as ONE
  // launch of a team of workgroups (team.hip.h); these are its exchange buffers (granules) and its "a wait was given up" flag
  unsigned long long* d_team_xb = nullptr;
  int* d_team_dead = nullptr;   // pinned host word: set by a team launch that gave a wait up (the call then returns zeros)
  size_t team_granules = 0;
  bool team_off = false;
  bool create(int B, int H, int n_slots, int n_add, int n_frm, float* shared_phone, int* shared_q, float* shared_feat,
              int front_slots = 1, bool pipe_slack = false, bool legacy = false);
  void destroy();
};
// parts of the module, for pipelines that cut it into stages: 1 = input mix, 2..5 = conditioned blocks 0..3,
// 6 = upsampler GEMMs, 7 = fused tail
struct WavePart { int first = 1, last = 7, scratch = 0; };
void wave_forward(const WaveWeights& w, const WaveState& s, hipStream_t stream, bool cond_done = false, WavePart part = WavePart());
void wave_cond(const WaveWeights& w, const WaveState& s, hipStream_t stream);

// content encoder + pitch estimator (+ the waveform generator's conditioning mix) with the pitch
// estimator's launches paired into the content encoder's (fuse.hip.h).  Returns false when the
// configuration is outside the paired regime (more than 2048 rows in the paired layers): nothing was enqueued, the
// caller runs phone_forward / pitch_forward / wave_forward(cond_done = false) instead.
void phone_vq(const PhoneWeights& w, const PhoneState& s, hipStream_t stream);
bool front_forward(const PhoneWeights& pw, const PhoneState& ps, const PitchWeights& qw, const PitchState& qs,
                   const WaveWeights& ww, const WaveState& ws, hipStream_t stream);

// set-time projections (embedding setter)
void embed_project_rows(const float* w, const float* b, const float* d_x, float* d_y, int rows, hipStream_t stream);
// (d_kt_plain / d_v_plain: optional second copies in plain row-major order, WaveState::d_ktp / d_vp)
void embed_project_kv(const EmbedWeights& w, int block, const float* d_kv_raw, int slots, float* d_kt, float* d_v,
                      hipStream_t stream, float* d_kt_plain = nullptr, float* d_v_plain = nullptr);
void codebook_prepare(const float* d_cb, int n, float* d_cbT, float* d_cnorm, hipStream_t stream);

// speaker morphing (morph.hip): out[row][:] = weighted spherical mean over table[speakers[n]][row][:],
// n < n_active <= 8, weights normalised and descending; dim = 128 or 256; one wavefront per row
bool spherical_mean_rows(const float* d_table, size_t speaker_stride, int rows, int dim, int n_active, const int* speakers,
                         const float* weights, float* d_out, hipStream_t stream);

}  // namespace bhip
