// batch.hip -- N-stream extension ABI (include/beatrice_batch.h).
//
// One BeatriceBatch = B independent streams advancing one hop per call, sharing the immutable model
// objects.  Host-side it mirrors, per stream, the settings and the call protocol that the reference
// host keeps per plugin instance (reference src/common/processor_core_2.cc; line references at each
// function).  Device-side it is phone_forward -> pitch_forward -> wave_forward on one HIP stream,
// replayed from a hipGraph (the chain has ~70 launches per hop; hop position is read from device
// memory by the kernels, so one captured graph serves every hop).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <deque>
#include <numeric>
#include <random>
#include <string>
#include <type_traits>
#include <chrono>
#include <deque>
#include <vector>

#include "abi_objects.h"
#include "beatrice_batch.h"
#include "tick.hip.h"
#include "wrapper.hip.h"

using namespace bhip;

namespace {

struct StreamCfg {
  int target_speaker = 0;
  int kv_set_count = B_NBLOCKS;  // reference: key_value_speaker_embedding_set_count_ (processor_core_2.h:134)
  int kv_slot[B_NBLOCKS] = {0, 0, 0, 0};
  int kv_delay = 0;               // hops before the pending blocks start to install (staged morph: the reference computes the means first)
  int codebook_speaker = 0;       // after a step: the codebook of its last hop
  int codebook_row[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // codebook of each hop of the step (a morphing stream draws one per hop)
  int additive_speaker = 0;
  int formant_index = 4;
  int vq_k = 0;
  int min_q = 1, max_q = B_PITCH_BINS - 1;
  PitchParams pitch{52.0, 1.0, 0.0, 0.0, 0, 0};  // defaults: processor_core_2.h:105-110
};

// A table entry that is a morph of real speakers (BeatriceBatch_MorphSpeaker): what the per-hop codebook
// lottery needs (reference processor_core_2.cc:94-121)
struct MorphSlot {
  bool active = false;
  int n_speakers = 0;           // real speakers the weights refer to
  int n_odds = 0;               // min(n_speakers, 8)
  int order[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float odds[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // pruned weights in `order`
};

// Pinned host copy of a small per-stream device array.  Double-buffered: push() sends the buffer the
// host has been editing and flips to the other one (brought up to date first), so the host never writes
// into memory an asynchronous copy may still be reading and no step has to wait for the previous one
// just because a setting changed.  The flip only blocks if the copy issued TWO pushes ago is unfinished.
template <class T>
struct Mirror {
  T* h = nullptr;  // the buffer being edited
  T* d = nullptr;
  size_t n = 0;
  T* buf[2] = {nullptr, nullptr};
  hipEvent_t sent[2] = {nullptr, nullptr};
  bool pending[2] = {false, false};
  int cur = 0;
  bool alloc_host(size_t n_) {
    n = n_;
    for (int i = 0; i < 2; ++i) {
      BHIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&buf[i]), sizeof(T) * n, hipHostMallocDefault));
      std::memset(buf[i], 0, sizeof(T) * n);
      BHIP_TRY(hipEventCreateWithFlags(&sent[i], hipEventDisableTiming));
    }
    h = buf[0];
    return true;
  }
  // sends [off, off + len) of the edited buffer to dst (nullptr: d + off) for each part, then flips
  bool push_parts(hipStream_t s, int n_parts, const size_t* off, const size_t* len, T* const* dst) {
    for (int i = 0; i < n_parts; ++i)
      BHIP_TRY(hipMemcpyAsync(dst[i] ? dst[i] : d + off[i], buf[cur] + off[i], sizeof(T) * len[i], hipMemcpyHostToDevice, s));
    return flip(s);
  }
  bool flip(hipStream_t s) {
    BHIP_TRY(hipEventRecord(sent[cur], s));
    pending[cur] = true;
    const int nxt = cur ^ 1;
    if (pending[nxt]) { BHIP_TRY(hipEventSynchronize(sent[nxt])); pending[nxt] = false; }
    std::memcpy(buf[nxt], buf[cur], sizeof(T) * n);
    cur = nxt;
    h = buf[cur];
    return true;
  }
  void release() {
    for (int i = 0; i < 2; ++i) {
      if (pending[i]) (void)hipEventSynchronize(sent[i]);
      if (buf[i]) (void)hipHostFree(buf[i]);
      if (sent[i]) (void)hipEventDestroy(sent[i]);
      buf[i] = nullptr; sent[i] = nullptr; pending[i] = false;
    }
    h = nullptr;
  }
};

}  // namespace

struct BeatriceBatch {
  const Beatrice20rc0_PhoneExtractor* phone_m = nullptr;
  const Beatrice20rc0_PitchEstimator* pitch_m = nullptr;
  const Beatrice20rc0_WaveformGenerator* wave_m = nullptr;
  const Beatrice20rc0_EmbeddingSetter* embed_m = nullptr;
  int B = 0, max_speakers = 0, n_speakers = 0;
  int device = -1;  // the GPU of the model objects the batch was created from; every entry point runs with it current (DeviceScope)
  int H = 1;  // hops per step (block mode when > 1)
  bool ok = false;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  PhoneState phone;
  PitchState pitch;
  WaveState wave;
  float* d_in = nullptr;  // [B][H*160], shared by phone and pitch
  // speaker tables on device
  float *d_cb_raw = nullptr, *d_cbT = nullptr, *d_cnorm = nullptr, *d_add_raw = nullptr, *d_frm_raw = nullptr, *d_kv_raw = nullptr;
  // per-stream settings
  std::vector<StreamCfg> cfg;
  std::vector<MorphSlot> morph;  // [max_speakers]
  int n_morph_slots = 0;
  // the codebook lottery's engine belongs to the stream, as the reference's belongs to the plugin instance
  // (processor_core_2.h:48,145): a stream's draws do not depend on which other streams share its batch
  std::vector<std::mt19937> lottery;  // [B]
  bool lottery_seeded = false;        // a caller's seed has been applied (first BeatriceBatch_MorphSpeaker, or BeatriceBatch_SeedLottery)
  // every per-stream setting array the kernels read lives in ONE device block with one pinned mirror, so a
  // step after any change costs a single small host-to-device copy (a dozen separate copies cost ~50 us of
  // stream time per step with 64 rotating speakers)
  Mirror<unsigned char> settings;
  // layout: [front part: arrays the front end reads][wave part: attention tile lists], and on the device the wave
  // part four times -- the waveform generator of step t reads copy t & 3, so a change can be pushed for step t while
  // the generator stages of steps t-1..t-3 are still running (their copies are refreshed when their slot comes up)
  struct { size_t cbT, cnorm, vqk, min_q, max_q, add_idx, frm_idx, params, perm[B_NBLOCKS], tile_slot[B_NBLOCKS], qperm[B_NBLOCKS], qslot[B_NBLOCKS], front_bytes, wave_bytes; } off{};
  bool front_dirty = true, wave_dirty[4] = {true, true, true, true};  // [kSlots]
  template <class T> T* host_view(size_t o) { return reinterpret_cast<T*>(settings.h + o); }
  template <class T> T* dev_view(size_t o) { return reinterpret_cast<T*>(settings.d + o); }
  void* module_owned[8 + 2 * B_NBLOCKS] = {};  // the modules' own (now unused) setting arrays, handed back before destroy()
  int pending_kv = 0;  // streams with kv_set_count < 4
  std::vector<int> row_slot[B_NBLOCKS];  // [B*H] K/V slot of attention row (stream, hop in step)
  bool kv_transient = false;  // rows of the last step's early hops still hold pre-switch slots (H > 1)
  bool vq_dirty = true;   // a VQ setting changed since the k-NN launch was last (de)selected
  bool inflight = false;  // device-variant steps have been enqueued since the last synchronisation
  // staging for the host variant
  float *h_in = nullptr, *h_out = nullptr;
  // One step = a chain of stages: stage 0 the front end (content encoder + pitch estimator + conditioning mix),
  // stages 1.. consecutive parts of the waveform generator (WavePart).  Each stage is its own launch (graph).
  // With pipelining off all stages go to `stream`, in order.  With a pipeline depth of n = 2..4 there are n
  // stages on n HIP streams, and stage s of step t+1 overlaps stage s+1 of step t whenever the caller enqueues
  // steps ahead of their completion: the chain is a string of ~40 launches that are each latency-bound and leave
  // most of the chip idle, so several steps in flight at different depths of the chain fill it.  Ordering:
  //   stage s of step t   after stage s-1 of step t        (data of the same step)
  //   stage s of step t   after stage s+1 of step t-3      (buffers that cross a stage boundary hold three steps:
  //                                                          three-slot phone / conditioning buffers, two spare slots
  //                                                          on the rings x[], ya2; the stages' scratch is private.
  //                                                          One step of slack keeps the ~20 us cross-stream hand-over
  //                                                          off the critical path: +2 % over two-step buffers)
  //   stage 0 of step t   after the last stage of step t-4 (counter pairs and attention tile lists have 4 copies)
  // Outputs are identical; a step that is waited for before the next is enqueued runs exactly as without pipelining.
  static constexpr int kMaxStages = 4, kSlots = 4;
  int n_stages = 2;      // stages of the current plan (2 when pipelining is off: front end, whole generator)
  bool pipelined = false;
  WavePart part[kMaxStages] = {};                 // [s], s >= 1
  hipStream_t stage_stream_own[kMaxStages] = {};  // [s], s >= 1; stage 0 runs on `stream`
  hipEvent_t ev_done[kMaxStages][kSlots] = {};    // stage s of step t enqueued/done, at [t & 3]
  long long steps_enqueued = 0;
  int hop_host = 0;     // mirror of the device step counter (same increments, same wrap)
  int last_parity = 0;  // slot (step counter mod 3) of the last enqueued step's phone vectors
  int* d_hop_wave = nullptr;  // int[4][2]: {counter, I/O slot} of step t at [t & 3], for the waveform generator's stages
  bool use_graph = true;
  hipGraph_t graph[kMaxStages][kSlots] = {};      // pipelined: stage 0 uses [0][0] only; in order: [0][slot] holds the whole step
  hipGraphExec_t exec[kMaxStages][kSlots] = {};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int* d_hop_next = nullptr;  // {step counter, resident-I/O slot}, double-buffered: first kernels read it, the last one writes it
  float* own_d_out = nullptr; // the waveform module's output buffer while a resident output buffer is bound
  int io_slots = 0;           // > 0: resident I/O bound (BeatriceBatch_BindResidentIO)
  bool io_mapped = false;     // BeatriceBatch_ConvertFrames: the chain reads the pinned input mirror and writes the pinned output mirror itself
  bool want_mapped = false;   // (set around the step ConvertFrames enqueues)
  float* dev_d_out = nullptr; // the waveform module's device output buffer while io_mapped
  int io_host = 0;            // mirror of the device's resident-I/O slot counter
  int last_hop = 0;           // step counter of the last enqueued step (selects the slot of the pitch head's outputs)
  tick::State tk;             // tick pipelining (tick.hip.h)
  // host streaming (BeatriceBatch_EnableHostStreaming): tick pipelining fed from / drained to HOST buffers, the copies
  // on their own streams beside the ticks
  struct HostStream {
    bool on = false;
    int n_slots = 0;
    float *d_in = nullptr, *d_out = nullptr;   // [n_slots][B][160], [n_slots][B][240]: the resident I/O the ticks use
    float *h_in = nullptr, *h_out = nullptr;   // pinned mirrors
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<hipEvent_t> ev_in, ev_out;     // per slot: upload done / download done
    std::vector<hipEvent_t> ev_tick;           // ring over ticks: tick launched (recorded on the batch's stream)
    struct Pending { long long step; int slot; long long done_tick; bool fetched; };
    std::deque<Pending> pending;               // steps fed and not yet handed back, oldest first
    long long fed = 0;
    long long rec[2] = {-1, -1};               // ticks whose events were recorded last and second to last
    bool mapped = false;                       // the ticks read and write the pinned mirrors themselves (no copies, no copy streams)
    std::vector<long long> tick_of_ev;         // which tick the event in ev_tick[i] was recorded behind
  } hs;
  // any-rate device wrapper (wrapper.hip.h): the reference host's gains, resampler pair and 480-sample FIFO for all streams
  wrapn::WrapPlan wrap;
  std::vector<wrapn::GainClock> gain_in, gain_out;  // [B]
  wrapn::StreamState* d_wrap = nullptr;             // [B]
  float *d_wrap_taps = nullptr, *d_wrap_inner = nullptr, *d_wrap_io = nullptr, *h_wrap_io = nullptr;  // taps: down | up; inner [B][kInnerStride]
  Mirror<wrapn::GainSeg> wrap_gains;                // [2][B]: input | output segments of the current call
  bool wrap_gains_constant = false;                 // the device copy holds constant segments that are still right
  // 48 kHz device wrapper (configs[4])
  Wrap48State* d_w48 = nullptr;
  // the same wrapper around the TICK pipeline (BeatriceBatch_BindResidentIO48k): resident 48 kHz slots, own 16 / 24 kHz slots between
  struct Resident48 {
    bool on = false;
    int channels = 0, n_slots = 0;
    const float* d_in48 = nullptr;   // [n_slots][B][channels][480]
    float* d_out48 = nullptr;        // [n_slots][B][channels][480]
    float *d_in16 = nullptr, *d_out24 = nullptr;  // [n_slots][B][160], [n_slots][B][240]: the resident I/O of the ticks
    int deferred_slot = -1;                       // step completed by the last tick, its 48 kHz block not yet produced
  } r48;
  float *d_coef_down = nullptr, *d_coef_up = nullptr, *d_io48 = nullptr, *h_io48 = nullptr;  // io: in [B][2][480] | out [B][2][480]
  // The any-rate wrapper around the tick pipeline (BeatriceBatch_BindResidentBlocks): host-rate blocks resident on the device,
  // the input half of the chain in front of the ticks, the output half `delay` calls later (wrapper.hip.h wrap_post_kernel)
  struct ResidentBlocks {
    bool on = false;
    int channels = 0, n = 0, n_slots = 0, io_slots = 0, delay = 0, ring = 0;
    const float* d_in = nullptr;   // [n_slots][B][channels][n]
    float* d_out = nullptr;        // [n_slots][B][channels][n]
    float *d_in16 = nullptr, *d_out24 = nullptr;   // [io_slots][B][160], [io_slots][B][240]: the resident I/O of the ticks
    wrapn::GainSeg *d_gains = nullptr, *h_gains = nullptr;   // [ring][2][B]: a call's input | output segments, until its output half has run
    hipEvent_t* gain_ev = nullptr;                           // [ring]: upload of ring entry done
    long long calls = 0, t48 = 0;                            // calls so far; 48 kHz samples fed so far
    struct Job { long long call, t0; wrapn::Dir dout; };
    std::deque<Job> jobs;                                    // calls whose output half is still to run, oldest first
  } rb;
};

namespace {

// Waits for the steps enqueued so far (needed before the graph or a speaker table they use is replaced;
// the pinned setting mirrors are double-buffered and do not need it).
hipStream_t stage_stream(const BeatriceBatch* b, int s) { return b->pipelined && s > 0 ? b->stage_stream_own[s] : b->stream; }
hipStream_t wave_stream(const BeatriceBatch* b) { return stage_stream(b, b->n_stages - 1); }  // where a step's output appears
bool tick_drain(BeatriceBatch* b);
void host_stream_free(BeatriceBatch* b);
bool sync_all(BeatriceBatch* b) {
  bool ok = !b->tk.on || tick_drain(b);  // steps still inside the tick pipeline come out first
  ok = hip_ok(hipStreamSynchronize(b->stream), "sync") && ok;
  for (int s = 1; s < BeatriceBatch::kMaxStages; ++s)
    if (b->stage_stream_own[s]) ok = hip_ok(hipStreamSynchronize(b->stage_stream_own[s]), "sync stage") && ok;
  b->inflight = false;
  return ok;
}
// Experiment switch: BEATRICE_HIP_CUMASK="lo-hi;lo-hi;..." gives the stream of stage 0, 1, ... a CU mask (CU index
// ranges), so that concurrently running stages do not land on the same CUs.
bool make_stage_stream(hipStream_t* st, int stage) {
  const char* spec = std::getenv("BEATRICE_HIP_CUMASK");
  if (spec) {
    std::string sp(spec);
    size_t pos = 0;
    for (int i = 0; i < stage && pos != std::string::npos; ++i) { pos = sp.find(';', pos); if (pos != std::string::npos) ++pos; }
    if (pos != std::string::npos && pos < sp.size()) {
      int lo = 0, hi = -1;
      if (std::sscanf(sp.c_str() + pos, "%d-%d", &lo, &hi) == 2 && lo >= 0 && hi >= lo && hi < 512) {
        uint32_t mask[16] = {};
        for (int c = lo; c <= hi; ++c) mask[c >> 5] |= 1u << (c & 31);
        return hip_ok(hipExtStreamCreateWithCUMask(st, 16, mask), "cu mask stream");
      }
    }
  }
  return make_stream(st);
}

// stage plans by pipeline depth (waveform parts: 1 input mix, 2..5 blocks, 6 upsampler GEMMs, 7 tail); the cuts
// balance the measured stage times at 256 streams (front end 83 us, generator 200 us)
void set_plan(BeatriceBatch* b, int depth) {
  b->pipelined = depth >= 2;
  b->n_stages = depth < 2 ? 2 : depth;
  switch (b->n_stages) {
    case 2: b->part[1] = WavePart{1, 7, 0}; break;
    case 3: b->part[1] = WavePart{1, 4, 0}; b->part[2] = WavePart{5, 7, 1}; break;
    default: b->part[1] = WavePart{1, 3, 0}; b->part[2] = WavePart{4, 5, 1}; b->part[3] = WavePart{6, 7, 2}; break;
  }
}
void settle(BeatriceBatch* b) {
  if (b->inflight) (void)sync_all(b);
}

// Tick mode splits the attention rows of a step between two kinds of workgroup (rowchain.hip.h): 16-row tiles that share a
// K/V slot (block_b_body) and quads of <= 4 rows (block_bq_body).  BEATRICE_HIP_TICK_NO_QUADS: A/B switch for measurements.
bool quads_on(const BeatriceBatch* b) {
  static const bool no_quads = std::getenv("BEATRICE_HIP_TICK_NO_QUADS") != nullptr;
  return b->tk.on && b->wave.d_ktp[0] != nullptr && !no_quads;
}
void rebuild_tiles(BeatriceBatch* b, int blk) {
  // attention rows (stream, hop in step) grouped by K/V slot, ascending slot then ascending row, 16 per tile
  const int nt = b->wave.n_tiles_max, rows = b->B * b->H;
  int* perm = b->host_view<int>(b->off.perm[blk]);
  int* slot = b->host_view<int>(b->off.tile_slot[blk]);
  for (bool& d : b->wave_dirty) d = true;
  const std::vector<int>& rs = b->row_slot[blk];
  std::fill(perm, perm + (size_t)nt * 16, -1);
  std::fill(slot, slot + nt, -1);
  std::vector<int> order(rows);
  for (int i = 0; i < rows; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return rs[x] < rs[y]; });
  // In-order modes: every row in a tile of its slot (at most one partial tile per slot).  Tick mode: a slot's rows fill whole
  // tiles first; a remainder of >= 8 rows is one more (padded) tile, a smaller one goes to the quad list -- with 64 speakers
  // on 256 streams that is 64 quads instead of 64 tiles with 12 of 16 rows empty.
  const bool quads = quads_on(b);
  int* qperm = b->host_view<int>(b->off.qperm[blk]);
  int* qslot = b->host_view<int>(b->off.qslot[blk]);
  std::fill(qperm, qperm + (size_t)nt * 16, -1);
  std::fill(qslot, qslot + (size_t)nt * 4, -1);
  int tile = 0, quad = 0;
  for (int i = 0; i < rows;) {
    const int sl = rs[order[i]];
    int n = 1;
    while (i + n < rows && rs[order[i + n]] == sl) ++n;
    int at = 0;
    while (n - at >= (quads ? 8 : 1)) {
      const int take = std::min(16, n - at);
      slot[tile] = sl;
      for (int e = 0; e < take; ++e) perm[tile * 16 + e] = order[i + at + e];
      ++tile; at += take;
    }
    while (at < n) {
      const int take = std::min(4, n - at);
      qslot[quad] = sl;
      for (int e = 0; e < take; ++e) qperm[quad * 4 + e] = order[i + at + e];
      ++quad; at += take;
    }
    i += n;
  }
}

void fill_row_slots(BeatriceBatch* b, int s) {
  for (int blk = 0; blk < B_NBLOCKS; ++blk)
    for (int hh = 0; hh < b->H; ++hh) b->row_slot[blk][(size_t)s * b->H + hh] = b->cfg[s].kv_slot[blk];
}

void sync_stream_arrays(BeatriceBatch* b, int s) {
  const StreamCfg& c = b->cfg[s];
  for (int hh = 0; hh < b->H; ++hh) {  // k-NN rows are (stream, hop in step)
    b->host_view<const float*>(b->off.cbT)[(size_t)s * b->H + hh] = b->d_cbT + (size_t)c.codebook_row[hh] * B_PHONE_CH * B_CODEBOOK;
    b->host_view<const float*>(b->off.cnorm)[(size_t)s * b->H + hh] = b->d_cnorm + (size_t)c.codebook_row[hh] * B_CODEBOOK;
  }
  b->host_view<int>(b->off.vqk)[s] = c.vq_k;
  b->host_view<int>(b->off.min_q)[s] = c.min_q;
  b->host_view<int>(b->off.max_q)[s] = c.max_q;
  b->host_view<int>(b->off.add_idx)[s] = c.additive_speaker;
  b->host_view<int>(b->off.frm_idx)[s] = c.formant_index;
  b->host_view<PitchParams>(b->off.params)[s] = c.pitch;
  b->front_dirty = true;
}

// One K/V block per stream per hop, as the reference host does before its three per-hop calls
// (processor_core_2.cc:179-181, processor_core_2.h:161-169).
void advance_kv(BeatriceBatch* b) {
  if (b->pending_kv == 0 && !b->kv_transient) return;
  bool dirty[B_NBLOCKS] = {false, false, false, false};
  bool advanced = false;
  const int H = b->H;
  for (int s = 0; s < b->B; ++s) {
    StreamCfg& c = b->cfg[s];
    for (int hh = 0; hh < H; ++hh) {  // the hops of this step, each preceded by one block install
      if (c.kv_delay > 0) {
        --c.kv_delay;
      } else if (c.kv_set_count < B_NBLOCKS) {
        c.kv_slot[c.kv_set_count] = c.target_speaker;
        ++c.kv_set_count;
        advanced = true;
      }
      for (int blk = 0; blk < B_NBLOCKS; ++blk) {
        int& rs = b->row_slot[blk][(size_t)s * H + hh];
        if (rs != c.kv_slot[blk]) { rs = c.kv_slot[blk]; dirty[blk] = true; }
      }
    }
  }
  int still = 0;
  for (const StreamCfg& c : b->cfg) if (c.kv_set_count < B_NBLOCKS) ++still;
  b->pending_kv = still;
  b->kv_transient = advanced && H > 1;
  for (int blk = 0; blk < B_NBLOCKS; ++blk) if (dirty[blk]) rebuild_tiles(b, blk);
}

// what changed since the last step goes to the device: the front part as it is, the wave part into the copy the
// waveform generator of THIS step (parity) will read
bool push_settings(BeatriceBatch* b, int parity /* slot: step & 3 */) {
  size_t off[2], len[2];
  unsigned char* dst[2];
  int n = 0;
  if (b->front_dirty) { off[n] = 0; len[n] = b->off.front_bytes; dst[n] = nullptr; ++n; }
  if (b->wave_dirty[parity]) {
    off[n] = b->off.front_bytes; len[n] = b->off.wave_bytes;
    dst[n] = b->settings.d + b->off.front_bytes + (size_t)parity * b->off.wave_bytes; ++n;
  }
  if (n == 0) return true;
  b->front_dirty = false;
  b->wave_dirty[parity] = false;
  return b->settings.push_parts(b->stream, n, off, len, dst);
}

// Front end.  Latency-bound regime (a few hundred rows): the pitch estimator's launches are paired into the
// content encoder's (front.hip); elsewhere the modules run one after the other.  (Running them as parallel
// graph branches was measured and buys nothing on ROCm 7.2 / MI355X, profiles/r01_notes.md.)
void enqueue_front(BeatriceBatch* b, hipStream_t st) {
  static const bool no_pairs = std::getenv("BEATRICE_HIP_NO_PAIRS") != nullptr;  // A/B switch for measurements
  if (!no_pairs && front_forward(b->phone_m->w, b->phone, b->pitch_m->w, b->pitch, b->wave_m->w, b->wave, st)) return;
  phone_forward(b->phone_m->w, b->phone, st);
  pitch_forward(b->pitch_m->w, b->pitch, st);
  wave_cond(b->wave_m->w, b->wave, st);
}
void enqueue_wave(BeatriceBatch* b, int stage, int slot, hipStream_t st) {
  b->wave.hop = b->d_hop_wave + 2 * slot;
  for (int blk = 0; blk < B_NBLOCKS; ++blk) {  // this slot's copy of the attention tile lists
    b->wave.d_perm[blk] = b->dev_view<int>(b->off.perm[blk] + (size_t)slot * b->off.wave_bytes);
    b->wave.d_tile_slot[blk] = b->dev_view<int>(b->off.tile_slot[blk] + (size_t)slot * b->off.wave_bytes);
  }
  wave_forward(b->wave_m->w, b->wave, st, /*cond_done=*/true, b->part[stage]);
}

void drop_graph(BeatriceBatch* b) {
  for (auto& per_stage : b->exec) for (hipGraphExec_t& e : per_stage) { if (e) (void)hipGraphExecDestroy(e); e = nullptr; }
  for (auto& per_stage : b->graph) for (hipGraph_t& g : per_stage) { if (g) (void)hipGraphDestroy(g); g = nullptr; }
}

template <class F>
bool capture(hipStream_t st, hipGraph_t* graph, hipGraphExec_t* exec, F enqueue) {
  BHIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  enqueue();
  BHIP_TRY(hipStreamEndCapture(st, graph));
  BHIP_TRY(hipGraphInstantiate(exec, *graph, nullptr, nullptr, 0));
  return true;
}

// pipelining off: the whole step as ONE launch on the batch's stream (a second graph launch per step costs ~15 us)
bool run_step_in_order(BeatriceBatch* b, int slot) {
  hipStream_t st = b->stream;
  auto enqueue_slot = [b, st](int sl) { enqueue_front(b, st); for (int s = 1; s < b->n_stages; ++s) enqueue_wave(b, s, sl, st); };
  if (!b->use_graph) { if (slot >= 0) enqueue_slot(slot); return hip_ok(hipGetLastError(), "step launch"); }
  if (slot < 0 || !b->exec[0][slot])  // first use: capture the variants of all four slots at once, so that no later step pays for a capture
    for (int sl = 0; sl < BeatriceBatch::kSlots; ++sl)
      if (!b->exec[0][sl] && !capture(st, &b->graph[0][sl], &b->exec[0][sl], [&] { enqueue_slot(sl); })) return false;
  if (slot < 0) return true;  // capture only (BeatriceBatch_Prepare)
  BHIP_TRY(hipGraphLaunch(b->exec[0][slot], st));
  return true;
}

bool run_stage(BeatriceBatch* b, int stage, int slot) {
  hipStream_t st = stage_stream(b, stage);
  auto enqueue_slot = [b, stage, st](int sl) { if (stage == 0) enqueue_front(b, st); else enqueue_wave(b, stage, sl, st); };
  if (!b->use_graph) { if (slot >= 0) enqueue_slot(slot); return hip_ok(hipGetLastError(), "stage launch"); }
  const int g = stage == 0 || slot < 0 ? 0 : slot;  // the front end reads its counter through one fixed pointer
  if (slot < 0 || !b->exec[stage][g])  // first use: every slot's variant of this stage at once (no capture inside a later step)
    for (int sl = 0; sl < (stage == 0 ? 1 : BeatriceBatch::kSlots); ++sl)
      if (!b->exec[stage][sl] && !capture(st, &b->graph[stage][sl], &b->exec[stage][sl], [&] { enqueue_slot(sl); })) return false;
  if (slot < 0) return true;  // capture only (BeatriceBatch_Prepare)
  BHIP_TRY(hipGraphLaunch(b->exec[stage][g], st));
  return true;
}

// the k-NN launch is dropped while no stream uses the codebook (phone.out then writes the module output)
void update_vq_mode(BeatriceBatch* b) {
  bool none = true;
  for (const StreamCfg& c : b->cfg) none = none && c.vq_k == 0;
  if (none != b->phone.skip_vq) { settle(b); b->phone.skip_vq = none; drop_graph(b); b->tk.table_dirty = true; }
}

// streams whose target is a morphed entry draw the codebook of ONE real speaker per hop, with the
// morph weights as odds (reference processor_core_2.cc:94-121: same draws, same order of operations), from
// the stream's own engine
void draw_codebooks(BeatriceBatch* b) {
  if (b->n_morph_slots == 0) return;
  for (int s = 0; s < b->B; ++s) {
    StreamCfg& c = b->cfg[s];
    MorphSlot& m = b->morph[c.target_speaker];
    if (!m.active) continue;
    std::mt19937& rng = b->lottery[s];
    float sum = 0.0f;
    for (int i = 0; i < m.n_odds; ++i) sum += m.odds[i];
    for (int hh = 0; hh < b->H; ++hh) {
      int idx = m.order[0];
      if (sum <= std::numeric_limits<float>::epsilon()) {
        idx = std::uniform_int_distribution<int>(0, m.n_speakers - 1)(rng);
      } else {
        float r = std::uniform_real_distribution<float>(0.0f, sum)(rng);
        for (int i = 0; i < m.n_odds; ++i) {
          r -= m.odds[i];
          if (r < 0.0f) { idx = m.order[i]; break; }
        }
      }
      c.codebook_row[hh] = idx;
    }
    c.codebook_speaker = c.codebook_row[b->H - 1];
    sync_stream_arrays(b, s);
  }
}

bool tick_run(BeatriceBatch* b, bool feeding);
// Host-buffer steps (BeatriceBatch_ConvertFrames) let the kernels read the pinned input mirror and write the pinned output
// mirror directly: two copy commands around the chain cost a switch to the copy engine and back each (0.35 -> 0.31 ms per
// step at 256 streams); every other kind of step uses the device buffers.
bool set_io_mapped(BeatriceBatch* b, bool on) {
  if (on == b->io_mapped) return true;
  if (on && (b->io_slots > 0 || b->tk.on || b->pipelined)) return true;   // not applicable: ConvertFrames copies as before
  if (!sync_all(b)) return false;
  drop_graph(b);  // kernel arguments change
  if (on) {
    b->dev_d_out = b->wave.d_out;
    b->phone.d_in = b->pitch.d_in = b->h_in;
    b->wave.d_out = b->h_out;
  } else {
    b->phone.d_in = b->pitch.d_in = b->d_in;
    b->wave.d_out = b->dev_d_out;
    b->dev_d_out = nullptr;
  }
  b->io_mapped = on;
  return true;
}

bool step_device(BeatriceBatch* b, const float* d_in, float* d_out) {
  if (b->io_slots > 0 && (d_in || d_out)) return false;  // resident I/O is bound: the step reads and writes its slots
  if (b->io_mapped != b->want_mapped && !set_io_mapped(b, b->want_mapped)) return false;
  if (b->tk.on) return tick_run(b, true);
  advance_kv(b);
  draw_codebooks(b);
  if (b->vq_dirty) { update_vq_mode(b); b->vq_dirty = false; }
  const long long t = b->steps_enqueued;
  const int slot = b->hop_host & 3, slot2 = (slot + 1) & 3 /* step t-3 */, last = b->n_stages - 1;
  hipStream_t fs = b->stream;
  if (!b->pipelined) {
    if (!push_settings(b, slot)) return false;
    if (d_in && d_in != b->d_in)
      BHIP_TRY(hipMemcpyAsync(b->d_in, d_in, sizeof(float) * b->B * b->H * B_IN_HOP, hipMemcpyDeviceToDevice, fs));
    if (!run_step_in_order(b, slot)) return false;
    if (d_out && d_out != b->wave.d_out)
      BHIP_TRY(hipMemcpyAsync(d_out, b->wave.d_out, sizeof(float) * b->B * b->H * B_OUT_HOP, hipMemcpyDeviceToDevice, fs));
    b->last_parity = b->hop_host % 3;
    b->last_hop = b->hop_host;
    b->hop_host = hop_next(b->hop_host);
    if (b->io_slots > 0) b->io_host = (b->io_host + 1) % b->io_slots;
    b->steps_enqueued = t + 1;
    b->inflight = true;
    return true;
  }
  // ordering rules: see the comment at BeatriceBatch::n_stages
  if (t >= 3) BHIP_TRY(hipStreamWaitEvent(fs, b->ev_done[1][slot2], 0));
  if (t >= 4) BHIP_TRY(hipStreamWaitEvent(fs, b->ev_done[last][slot], 0));
  if (!push_settings(b, slot)) return false;
  if (d_in && d_in != b->d_in)
    BHIP_TRY(hipMemcpyAsync(b->d_in, d_in, sizeof(float) * b->B * b->H * B_IN_HOP, hipMemcpyDeviceToDevice, fs));
  if (!run_stage(b, 0, slot)) return false;
  BHIP_TRY(hipEventRecord(b->ev_done[0][slot], fs));
  for (int s = 1; s <= last; ++s) {
    hipStream_t st = stage_stream(b, s);
    BHIP_TRY(hipStreamWaitEvent(st, b->ev_done[s - 1][slot], 0));
    if (s < last && t >= 3) BHIP_TRY(hipStreamWaitEvent(st, b->ev_done[s + 1][slot2], 0));
    if (!run_stage(b, s, slot)) return false;
    if (s == last && d_out && d_out != b->wave.d_out)
      BHIP_TRY(hipMemcpyAsync(d_out, b->wave.d_out, sizeof(float) * b->B * b->H * B_OUT_HOP, hipMemcpyDeviceToDevice, st));
    BHIP_TRY(hipEventRecord(b->ev_done[s][slot], st));
  }
  b->last_parity = b->hop_host % 3;
  b->last_hop = b->hop_host;
  b->hop_host = hop_next(b->hop_host);
  if (b->io_slots > 0) b->io_host = (b->io_host + 1) % b->io_slots;
  b->steps_enqueued = t + 1;
  b->inflight = true;
  return true;
}

// ---- tick pipelining (tick.hip.h) ---------------------------------------------------------------------------------
bool tick_build_table(BeatriceBatch* b) {
  using namespace tick;
  State& k = b->tk;
  auto tb = std::make_unique<Builder>();
  const PhoneWeights& pw = b->phone_m->w;
  const PitchWeights& qw = b->pitch_m->w;
  const WaveWeights& ww = b->wave_m->w;
  const PhoneState& ps = b->phone;
  const PitchState& qs = b->pitch;
  const WaveState& ws = b->wave;
  const int B = b->B;
  const Plan pl = k.plan;
  // measurement aid, MEASUREMENT BUILDS ONLY (tools/debug/build_variant.sh <name> -DBEATRICE_HIP_MEASUREMENT_BUILD; the
  // product library has no switch that changes results): BEATRICE_HIP_TICK_DROP=<bit mask> leaves groups of bodies out of
  // the launch
#ifdef BEATRICE_HIP_MEASUREMENT_BUILD
  static const int drop = std::getenv("BEATRICE_HIP_TICK_DROP") ? std::atoi(std::getenv("BEATRICE_HIP_TICK_DROP")) : 0;
#else
  constexpr int drop = 0;
#endif
  auto keep = [](int group) { return ((drop >> group) & 1) == 0; };
  // (every body takes its step counter from the launch's StepPairs -- a null counter pointer says so, ring.h stepc)
  auto hp = [&](int) -> const int* { return nullptr; };
  auto conv = [&](const Ring& in, const Ring& out, const float* w, const float* bias, int stage) { return conv_args(in, out, w, bias, hp(stage), B); };
  // (workgroups are dispatched in this order: the longest-running bodies first)
  // ---- longest workgroups first (measured, two per CU): f5 48 us, p1 46, f4 45, rb 42, block halves 41 / 38, tail 36
  { const ConvArgs a = conv(ps.f[3], ps.f[4], pw.f_w[3], pw.f_b[3], Plan::F5); tb->add<T_F5>(OpF5::info("phone.f5", a), a, OpF5::grid(a), Plan::F5, keep(6), 47); }
  { const ConvArgs a = conv(qs.spec, qs.p[0], qw.p_w[0], qw.p_b[0], Plan::P1); tb->add<T_P1>(OpP1::info("pitch.p1", a), a, OpP1::grid(a), Plan::P1, keep(2), 34.5); }
  { const ConvArgs a = conv(ps.f[2], ps.f[3], pw.f_w[2], pw.f_b[2], Plan::F4); tb->add<T_F4>(OpF4::info("phone.f4", a), a, OpF4::grid(a), Plan::F4, keep(6), 37.5); }
  for (int i = 0; i < 4; ++i) {
    const ConvArgs a = conv(i == 0 ? ps.f[4] : ps.rb[i - 1], ps.rb[i], pw.rb_w[i], pw.rb_b[i], Plan::RB0 + i);
    tb->add<T_RB>(OpRB::info("phone.rb", a), a, OpRB::grid(a), Plan::RB0 + i, keep(6), 46);
  }
  // conditioned blocks: two row-local chains each
  for (int blk = 0; blk < B_NBLOCKS; ++blk) {
    const WaveState::Scratch& sc = ws.scr[blk];  // one scratch set per block: all four blocks are in flight at once
    const int s0 = pl.blk(blk);
    const rc::BlockBArgs ba{sc.xa, ws.x[blk + 1], ww.q_w[blk], ww.q_b[blk], ww.o_w[blk], ww.o_b[blk], ws.d_kt[blk], ws.d_v[blk],
                            b->dev_view<int>(b->off.perm[blk]), b->dev_view<int>(b->off.tile_slot[blk]), hp(s0 + 1)};
    tb->add<T_BLKB>(LaunchInfo{"wave.blk.b", 2.0 * B * (256.0 * 256 * 2 + 256.0 * 384 * 2), 4.0 * (2.0 * 256 * 256 + 2.0 * 256 * 384 + B * 3.0 * 256)}, ba,
                    dim3(ws.n_tiles_max, 1), s0 + 1, keep(5), 41.0);
    if (quads_on(b)) {  // rows without 15 neighbours on their K/V slot: one workgroup per quad (rebuild_tiles decides which rows)
      const rc::BlockBqArgs bq{sc.xa, ws.x[blk + 1], ww.q_w[blk], ww.q_b[blk], ww.o_w[blk], ww.o_b[blk], ws.d_ktp[blk], ws.d_vp[blk],
                               b->dev_view<int>(b->off.qperm[blk]), b->dev_view<int>(b->off.qslot[blk]), hp(s0 + 1)};
      // (a slot leaves at most 7 rows = 2 quads to this list: <= n_slots workgroups of two quads)
      tb->add<T_BLKBQ>(LaunchInfo{"wave.blk.bq", 0.0, 0.0}, bq, dim3(std::min(2 * ws.n_tiles_max, ws.n_slots), 1), s0 + 1, keep(5), 42.0);
    }
  }
  for (int blk = 0; blk < B_NBLOCKS; ++blk) {
    const rc::BlockAArgs aa{ws.x[blk], ws.scr[blk].xa, ww.c1_w[blk], ww.c1_b[blk], ww.c2_w[blk], ww.c2_b[blk], hp(pl.blk(blk)), B};
    switch (blk) {
      case 0: tb->add<T_BLKA1>(rc::BlockAOp<1>::info(aa), aa, rc::BlockAOp<1>::grid(aa), pl.blk(blk), keep(5), 41); break;
      case 1: tb->add<T_BLKA2>(rc::BlockAOp<2>::info(aa), aa, rc::BlockAOp<2>::grid(aa), pl.blk(blk), keep(5), 41); break;
      case 2: tb->add<T_BLKA4>(rc::BlockAOp<4>::info(aa), aa, rc::BlockAOp<4>::grid(aa), pl.blk(blk), keep(5), 41); break;
      default: tb->add<T_BLKA8>(rc::BlockAOp<8>::info(aa), aa, rc::BlockAOp<8>::grid(aa), pl.blk(blk), keep(5), 41); break;
    }
  }
  if (!pl.split_tail) {
    TailArgs ta = tail_args(ww, ws); ta.hop = hp(pl.tail()); tb->add<T_TAIL>(tail_info(ws), ta, dim3(B, 1), pl.tail(), keep(4), 41, true);
  } else {
    // the tail as three stages, several streams per workgroup (tail_stages.hip.h); same weights, same state block
    tst::StageArgs t1{}, t2{}, t3{};
    t1.in = ws.ya2; t1.out = ws.ya3; t2.in = ws.ya3; t2.out = ws.ya4; t3.in = ws.ya4;
    for (tst::StageArgs* t : {&t1, &t2, &t3}) { t->state = ws.tail.base; t->hop = nullptr; t->B = B; }
    t1.w[0] = ww.ra_w[1]; t1.b[0] = ww.ra_b[1]; t1.w[1] = ww.rb_w[1]; t1.b[1] = ww.rb_b[1]; t1.w[2] = ww.up_w[2]; t1.b[2] = ww.up_b[2];
    t2.w[0] = ww.ra_w[2]; t2.b[0] = ww.ra_b[2]; t2.w[1] = ww.rb_w[2]; t2.b[1] = ww.rb_b[2]; t2.w[2] = ww.up_w[3]; t2.b[2] = ww.up_b[3];
    t3.w[0] = ww.ra_w[3]; t3.b[0] = ww.ra_b[3]; t3.w[1] = ww.rb_w[3]; t3.b[1] = ww.rb_b[3];
    t3.fin_w = ww.fin_w; t3.fin_b = ww.fin_b; t3.d_out = ws.d_out; t3.io_stride = ws.io_stride;
    tb->add<T_TAIL1>(tst::T1Op::info(t1), t1, tst::T1Op::grid(t1), pl.tail(), keep(4), 30, true);
    tb->add<T_TAIL2>(tst::T2Op::info(t2), t2, tst::T2Op::grid(t2), pl.tail() + 1, keep(4), 28, true);
    tb->add<T_TAIL3>(tst::T3Op::info(t3), t3, tst::T3Op::grid(t3), pl.tail() + 2, keep(4), 16, true);
  }
  // ---- everything else, longest workgroups first (they start when the heavy ones above leave their slots)
  { const ConvArgs a = conv(ps.f[1], ps.f[2], pw.f_w[1], pw.f_b[1], Plan::F3); tb->add<T_F3>(OpF3::info("phone.f3", a), a, OpF3::grid(a), Plan::F3, keep(6), 18); }
  { const ConvArgs a = conv(ws.x[4], ws.ya1, ww.up_w[0], ww.up_b[0], pl.up1()); tb->add<T_UP1>(OpUP1::info("wave.up1", a), a, OpUP1::grid(a), pl.up1(), keep(7), 36); }
  { const ConvArgs a = conv(ws.yb1, ws.yc1, ww.rb_w[0], ww.rb_b[0], pl.up1() + 2); tb->add<T_RES1B>(OpRES1B::info("wave.res1b", a), a, OpRES1B::grid(a), pl.up1() + 2, keep(7), 10); }
  { const ConvArgs a = conv(ws.ya1, ws.yb1, ww.ra_w[0], ww.ra_b[0], pl.up1() + 1); tb->add<T_RES1A>(OpRES1A::info("wave.res1a", a), a, OpRES1A::grid(a), pl.up1() + 1, keep(7), 10); }
  { const ConvArgs a = conv(ws.yc1, ws.ya2, ww.up_w[1], ww.up_b[1], pl.up1() + 3); tb->add<T_UP2>(OpUP2::info("wave.up2", a), a, OpUP2::grid(a), pl.up1() + 3, keep(7), 12); }
  { const ConvArgs a = conv(ps.f[0], ps.f[1], pw.f_w[0], pw.f_b[0], Plan::F2); tb->add<T_F2>(OpF2::info("phone.f2", a), a, OpF2::grid(a), Plan::F2, keep(6), 10); }
  { const GruArgs g{ps.rb[3], ps.h, pw.gru_wih, pw.gru_whh, pw.gru_bih, pw.gru_bhh, hp(Plan::PGRU), B, 0}; tb->add<T_PGRU>(GruOp<256, 256, TICK_GRU_RT>::info("phone.gru", g), g, GruOp<256, 256, TICK_GRU_RT>::grid(g), Plan::PGRU, keep(1), 7.6); }
  { const ConvArgs a = conv(qs.h, qs.logits, qw.out_w, qw.out_b, Plan::POUT); tb->add<T_POUT>(OpPOUT::info("pitch.out", a), a, OpPOUT::grid(a), Plan::POUT, keep(2), 9.4); }
  for (int i = 0; i < 2; ++i) {
    const ConvArgs a = conv(qs.p[i], qs.p[i + 1], qw.p_w[i + 1], qw.p_b[i + 1], Plan::P2 + i);
    tb->add<T_P23>(OpP23::info("pitch.p23", a), a, OpP23::grid(a), Plan::P2 + i, keep(2), 8);
  }
  { const Ring phone_in{ws.d_phone, B_PHONE_CH, 1, ws.front_slots}; ConvArgs a = conv(phone_in, ws.x[0], ww.inp_w, ww.inp_b, Plan::INP); a.res = ws.e; tb->add<T_INP>(OpINP::info("wave.inp", a), a, OpINP::grid(a), Plan::INP, keep(3), 7.7); }
  { const GruArgs g{qs.p[2], qs.h, qw.gru_wih, qw.gru_whh, qw.gru_bih, qw.gru_bhh, hp(Plan::QGRU), B, 0}; tb->add<T_QGRU>(GruOp<128, 128, TICK_GRU_RT>::info("pitch.gru", g), g, GruOp<128, 128, TICK_GRU_RT>::grid(g), Plan::QGRU, keep(1), 4.6); }
  { FftArgs a = fft_args(qw, qs); a.hop = hp(Plan::FFT); tb->add<T_FFT>(fft_info(qs), FftArgs2{a, B}, dim3((B + 1) / 2, 1), Plan::FFT, keep(0), 6); }
  { const ConvArgs a = conv(ps.h, phone_out_ring(ps), pw.out_w, pw.out_b, Plan::OUT); tb->add<T_OUT>(OpOUT::info("phone.out", a), a, OpOUT::grid(a), Plan::OUT, keep(3), 6); }
  { const VqArgs a{1, ps.raw, phone_vector_ring(ps), hp(Plan::VQ), ps.d_cbT, ps.d_cnorm, ps.d_vqk}; tb->add<T_VQ>(LaunchInfo{"phone.vq", 0, 4.0 * B * 256}, a, dim3(B, 1), Plan::VQ, !ps.skip_vq, 6.0, true); }
  { PitchHeadArgs a = head_args(qw, qs); a.hop = hp(Plan::HEAD); tb->add<T_HEAD>(head_info(qs), a, dim3((B + 7) / 8, 1), Plan::HEAD, keep(0), 4.7); }
  { F1Args a = f1_args(pw, ps); a.hop = hp(Plan::F1); a.hop_publish = nullptr; a.hop_publish_wave = nullptr; tb->add<T_F1>(f1_info(ps), F1Args2{a, B}, dim3((B + 1) / 2, 1), Plan::F1, keep(0), 4.5); }
  { CondArgs a = cond_args(ww, ws); a.hop = hp(Plan::COND); a.hop_next_out = nullptr; tb->add<T_COND>(cond_info(ws), a, dim3((B + 1) / 2, 1), Plan::COND, keep(0), 1.3); }
  if (!tb->ok) return false;
  // XCD-aware placement (bodies with many weights pinned to one XCD each, so that the weights stay in that L2) was
  // measured twice: it cuts the launch's memory-side traffic 4x (rocprofv3 FETCH_SIZE 105 -> 26 MB per tick) and the
  // pinned workgroups run ~10-25 % shorter, but confining a body to 32 CUs costs more in makespan than that gains
  // (0.096 vs 0.090 ms per tick): off by default
  static const bool by_xcd = std::getenv("BEATRICE_HIP_TICK_XCD") != nullptr;
  if (by_xcd) tb->place_by_xcd();
  if (std::getenv("BEATRICE_HIP_TICK_TRACE")) {
    if (k.d_trace) (void)hipFree(k.d_trace);
    BHIP_TRY(hipMalloc(reinterpret_cast<void**>(&k.d_trace), sizeof(unsigned long long) * 3 * tb->t.total));
    tb->t.trace = k.d_trace;
  }
  BHIP_TRY(hipMemcpy(k.d_table, &tb->t, sizeof(Tab), hipMemcpyHostToDevice));
  k.table_total = tb->t.total;
  k.table_flops = tb->flops;
  k.table_bytes = tb->bytes;
  // who reads which part of the settings block, and where its private copy lives
  k.consumers.clear();
  unsigned char* d = b->settings.d;
  k.consumers.push_back(Consumer{Plan::VQ, b->off.cbT, b->off.min_q - b->off.cbT, d + b->off.cbT, -1});
  k.consumers.push_back(Consumer{Plan::HEAD, b->off.min_q, b->off.add_idx - b->off.min_q, d + b->off.min_q, -1});
  k.consumers.push_back(Consumer{Plan::COND, b->off.add_idx, b->off.front_bytes - b->off.add_idx, d + b->off.add_idx, -1});
  for (int blk = 0; blk < B_NBLOCKS; ++blk) {
    const size_t lo = b->off.perm[blk], hi = blk + 1 < B_NBLOCKS ? b->off.perm[blk + 1] : b->off.front_bytes + b->off.wave_bytes;
    k.consumers.push_back(Consumer{pl.blk(blk) + 1, lo, hi - lo, d + lo, -1});
  }
  k.table_dirty = false;
  return true;
}

// One tick: every stage advances by one step; `feeding` = a new step enters at stage 0.
// BEATRICE_HIP_TICK_HOSTPROF=1: host time of tick_run by section, printed when the process ends (measurement aid)
struct HostProf {
  static constexpr int N = 5;
  static bool on() { static const bool v = std::getenv("BEATRICE_HIP_TICK_HOSTPROF") != nullptr; return v; }
  struct Totals { double us[N] = {}; long long calls = 0; ~Totals() { if (calls) std::fprintf(stderr, "tick_run host us per call: settings/kv %.1f, table %.1f, snapshot upload %.1f, copies %.1f, launch %.1f (%lld calls)\n", us[0] / calls, us[1] / calls, us[2] / calls, us[3] / calls, us[4] / calls, calls); } };
  static Totals& totals() { static Totals t; return t; }
  std::chrono::steady_clock::time_point t0;
  HostProf() { if (on()) { t0 = std::chrono::steady_clock::now(); totals().calls += 1; } }
  void lap(int i) {
    if (!on()) return;
    const auto t1 = std::chrono::steady_clock::now();
    totals().us[i] += std::chrono::duration<double, std::micro>(t1 - t0).count();
    t0 = t1;
  }
};
bool tick_run(BeatriceBatch* b, bool feeding) {
  using namespace tick;
  State& k = b->tk;
  HostProf prof;
  if (feeding) {
    advance_kv(b);
    draw_codebooks(b);
    if (b->vq_dirty) { update_vq_mode(b); b->vq_dirty = false; }   // (drains the pipeline if the k-NN stage comes or goes)
  }
  prof.lap(0);
  if (k.table_dirty && !tick_build_table(b)) return false;
  prof.lap(1);
  hipStream_t st = b->stream;
  Copy upload{nullptr, nullptr, 0};
  int upload_stage = -1;
  if (feeding) {
    bool dirty = b->front_dirty || k.snap_cur < 0;
    for (bool w : b->wave_dirty) dirty = dirty || w;
    if (dirty) {  // a new version of the settings: one upload into the next slot of the snapshot ring
      const int serial = k.snap_next++;
      const size_t off = 0, len = k.snap_bytes;
      unsigned char* dst = k.d_snap + (size_t)(serial % kRing) * k.snap_bytes;
      // through a ring of pinned staging copies, so that the host may run several settings changes ahead of the device
      // (the batch's two-deep mirror would make every second change wait for the copy of the change before it); the
      // tick's prologue kernel reads the staging copy straight from host memory (tick.hip.h)
      const int si = serial % State::kStaging;
      if (k.stage_pending[si]) { if (!hip_ok(hipEventSynchronize(k.stage_ev[si]), "tick settings staging")) return false; }
      unsigned char* src = k.h_stage + (size_t)si * k.snap_bytes;
      std::memcpy(src, b->settings.h + off, len);
      upload = Copy{dst, src, (int)len};
      upload_stage = si;
      k.snap_cur = serial;
      b->front_dirty = false;
      for (bool& w : b->wave_dirty) w = false;
    }
    const long long u = k.n_fed;
    k.fed_step[k.tick % kRing] = u;
    k.snap_of_step[u % kRing] = k.snap_cur;
    k.hop_of_step[u % kRing] = b->hop_host;
    k.io_of_step[u % kRing] = b->io_host;
  } else {
    k.fed_step[k.tick % kRing] = -1;
  }
  prof.lap(2);
  Prolog p{};
  p.n_stages = k.plan.count();
  auto step_at = [&k](int stage) -> long long {
    const long long t2 = k.tick - stage;
    return t2 >= 0 ? k.fed_step[t2 % kRing] : -1;
  };
  for (int s = 0; s < p.n_stages; ++s) {
    const long long u = step_at(s);
    p.hop[s] = u < 0 ? -1 : k.hop_of_step[u % kRing];
    p.io[s] = u < 0 ? 0 : k.io_of_step[u % kRing];
  }
  for (Consumer& c : k.consumers) {
    const long long u = step_at(c.stage);
    if (u < 0) continue;
    const int want = k.snap_of_step[u % kRing];
    if (want == c.held) continue;
    p.copy[p.n_copies++] = Copy{c.dst, k.d_snap + (size_t)(want % kRing) * k.snap_bytes + c.off, (int)c.bytes};
    c.held = want;
  }
  if (upload.bytes > 0) {  // first, so that entry order = age; (a consumer never needs the snapshot uploaded in its own tick: none sits at stage 0)
    for (int i = p.n_copies; i > 0; --i) p.copy[i] = p.copy[i - 1];
    p.copy[0] = upload;
    p.n_copies += 1;
  }
  if (p.n_copies > 0) {  // (only on ticks where the settings changed or a change arrives at a consumer)
    int chunks = 0;
    for (int i = 0; i < p.n_copies; ++i) { p.first_chunk[i] = chunks; chunks += (p.copy[i].bytes + kCopyChunk - 1) / kCopyChunk; }
    p.first_chunk[p.n_copies] = chunks;
    hipLaunchKernelGGL(prologue_kernel, dim3(chunks), dim3(256), 0, st, p);
    if (upload_stage >= 0) {
      if (!hip_ok(hipEventRecord(k.stage_ev[upload_stage], st), "tick settings event")) return false;
      k.stage_pending[upload_stage] = true;
    }
  }
  if (b->r48.on && (feeding || b->r48.deferred_slot >= 0)) {
    // one launch for both ends of the 48 kHz wrapper: the block entering the pipeline -> its 16 kHz hop, straight into the
    // resident slot; and the step the PREVIOUS tick completed leaves through the up-sampler and the 480-sample FIFO
    // (resample.h:346-361: the block emitted for step j carries the model output of step j - 1) into the 48 kHz slot of step j
    BeatriceBatch::Resident48& r = b->r48;
    Wrap48TickArgs wa{};
    wa.channels = r.channels; wa.st = b->d_w48; wa.coef_down = b->d_coef_down; wa.coef_up = b->d_coef_up;
    if (feeding) {
      wa.n_pre = b->B;
      wa.in48 = r.d_in48 + (size_t)b->io_host * b->B * r.channels * 480;
      wa.in16 = r.d_in16 + (size_t)b->io_host * b->B * B_IN_HOP;
    }
    if (r.deferred_slot >= 0) {
      wa.n_post = b->B;
      wa.out48 = r.d_out48 + (size_t)r.deferred_slot * b->B * r.channels * 480;
      wa.model_out = r.d_out24 + (size_t)r.deferred_slot * b->B * B_OUT_HOP;
      r.deferred_slot = -1;
    }
    hipLaunchKernelGGL(wrap48_tick_kernel, dim3(wa.n_pre + wa.n_post), dim3(256), 0, st, wa);
  }
  prof.lap(3);
  fuse::StepPairs pairs;
  for (int s = 0; s < fuse::kMaxStepPairs; ++s) { pairs.hop[s] = s < p.n_stages ? p.hop[s] : -1; pairs.io[s] = s < p.n_stages ? p.io[s] : 0; }
  fuse::launch_table_w<4>(k.d_table, k.table_total, st, pairs);
  if (b->r48.on) {  // the step this tick completed: its 48 kHz block is produced by the wrapper launch of the next tick (or of the drain)
    const long long u = step_at(k.plan.count() - 1);
    if (u >= 0) b->r48.deferred_slot = k.io_of_step[u % kRing];
  }
  if (feeding) {
    b->last_parity = b->hop_host % 3;
    b->last_hop = b->hop_host;
    b->hop_host = hop_next(b->hop_host);
    b->io_host = (b->io_host + 1) % b->io_slots;
    b->steps_enqueued += 1;
    k.n_fed += 1;
    k.last_feed_tick = k.tick;
  }
  prof.lap(4);
  k.tick += 1;
  b->inflight = true;
  return hip_ok(hipGetLastError(), "tick launch");
}
// ticks without new input until the last step fed has left the last stage
// output half of one call of the any-rate wrapper around the ticks (BeatriceBatch_BindResidentBlocks): its block's inner
// samples gathered from the resident model outputs, second resampling direction, output gain, into the call's slot
bool rb_post(BeatriceBatch* b, const BeatriceBatch::ResidentBlocks::Job& j) {
  BeatriceBatch::ResidentBlocks& r = b->rb;
  const size_t nt = b->wrap.taps_down.size();
  const int slot = (int)(j.call % r.n_slots), ge = (int)(j.call % r.ring);
  hipLaunchKernelGGL(wrapn::wrap_post_kernel, dim3(b->B), dim3(256), 0, b->stream, r.d_out24, r.io_slots, b->B, j.t0, b->d_wrap,
                     r.d_gains + (size_t)ge * 2 * b->B + b->B, b->d_wrap_taps + (j.dout.decimate ? 0 : nt), j.dout,
                     r.d_out + (size_t)slot * b->B * r.channels * r.n, r.channels);
  return hip_ok(hipGetLastError(), "wrapper output half");
}
bool tick_drain(BeatriceBatch* b) {
  bool ok = true;
  if (b->tk.on && b->tk.d_trace && b->tk.last_feed_tick == b->tk.tick - 1 && b->tk.n_fed > b->tk.plan.count()) {
    // measurement aid: the tick just enqueued had every stage busy; dump its per-workgroup timeline (100 MHz wall clock)
    std::vector<unsigned long long> tr((size_t)3 * b->tk.table_total);
    if (hip_ok(hipStreamSynchronize(b->stream), "trace sync") &&
        hip_ok(hipMemcpy(tr.data(), b->tk.d_trace, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost), "trace copy"))
      if (FILE* f = std::fopen(std::getenv("BEATRICE_HIP_TICK_TRACE"), "w")) {
        for (size_t i = 0; i < tr.size(); i += 3) std::fprintf(f, "%llu %llu %llu\n", tr[i], tr[i + 1], tr[i + 2]);
        std::fclose(f);
      }
  }
  while (ok && b->tk.on && b->tk.tick <= b->tk.last_feed_tick + b->tk.plan.count() - 1) ok = tick_run(b, false);
  if (ok && b->r48.on && b->r48.deferred_slot >= 0) {  // the 48 kHz block of the step the last tick completed
    BeatriceBatch::Resident48& r = b->r48;
    Wrap48TickArgs wa{};
    wa.channels = r.channels; wa.st = b->d_w48; wa.coef_down = b->d_coef_down; wa.coef_up = b->d_coef_up;
    wa.n_post = b->B;
    wa.out48 = r.d_out48 + (size_t)r.deferred_slot * b->B * r.channels * 480;
    wa.model_out = r.d_out24 + (size_t)r.deferred_slot * b->B * B_OUT_HOP;
    r.deferred_slot = -1;
    hipLaunchKernelGGL(wrap48_tick_kernel, dim3(wa.n_post), dim3(256), 0, b->stream, wa);
    ok = hip_ok(hipGetLastError(), "wrap48 flush");
  }
  while (ok && b->rb.on && !b->rb.jobs.empty()) {  // every model hop has left the pipeline: the output halves still owed, in order
    ok = rb_post(b, b->rb.jobs.front());
    b->rb.jobs.pop_front();
  }
  return ok;
}
int tick_enable(BeatriceBatch* b, bool on) {
  using namespace tick;
  State& k = b->tk;
  if (on == k.on) return 0;
  if (on) {
    // one 10 ms hop per step, resident I/O with enough slots
    // that a step's input is still there when the pitch head reads it nine ticks on and outputs have somewhere to land
    if (b->H != 1 || b->B > 4096 || b->io_slots < k.plan.count() + 1) return -1;
    if (!sync_all(b)) return -2;
    if (b->pipelined) { drop_graph(b); set_plan(b, 1); }
    k.snap_bytes = b->off.front_bytes + b->off.wave_bytes;
    if (!k.d_table) {
      if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&k.d_table), sizeof(Tab)), "tick table") ||
          !hip_ok(hipMalloc(reinterpret_cast<void**>(&k.d_snap), k.snap_bytes * kRing), "tick snapshots") ||
          !hip_ok(hipHostMalloc(reinterpret_cast<void**>(&k.h_stage), k.snap_bytes * State::kStaging, hipHostMallocDefault), "tick staging"))
        return -2;
      for (hipEvent_t& e : k.stage_ev) if (!hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "tick staging event")) return -2;
    }
    if (!hip_ok(hipDeviceSynchronize(), "tick sync")) return -2;
    k.tick = 0; k.n_fed = 0; k.last_feed_tick = -1000; k.snap_cur = -1; k.snap_next = 0;
    for (long long& f : k.fed_step) f = -1;
    k.table_dirty = true;
    k.on = true;
    for (int blk = 0; blk < B_NBLOCKS; ++blk) rebuild_tiles(b, blk);  // (tick mode cuts the attention rows into tiles AND quads)
    return 0;
  }
  if (!sync_all(b)) return -2;  // drains
  k.on = false;
  for (int blk = 0; blk < B_NBLOCKS; ++blk) rebuild_tiles(b, blk);
  // the in-order chain reads its counters from device memory: hand them the host's values
  const int pair[2] = {b->hop_host, b->io_host};
  if (!hip_ok(hipMemcpy(b->d_hop_next, pair, sizeof(pair), hipMemcpyHostToDevice), "tick leave")) return -2;
  b->front_dirty = true;
  for (bool& w : b->wave_dirty) w = true;
  return 0;
}

template <class F>
int for_streams(BeatriceBatch* b, int stream, F f) {
  if (!b || !b->ok) return -2;
  if (stream < -1 || stream >= b->B) return -1;
  const int lo = stream < 0 ? 0 : stream, hi = stream < 0 ? b->B : stream + 1;
  for (int s = lo; s < hi; ++s) { f(b->cfg[s]); sync_stream_arrays(b, s); }
  return 0;
}

int midi_to_bin(double note) {
  // reference processor_core_2.cc:561-583: clamp note to [0,128], bin = round((note-33)*8), clamp 1..447
  note = std::min(std::max(note, 0.0), 128.0);
  const int q = (int)std::round((note - 33.0) * (BEATRICE_PITCH_BINS_PER_OCTAVE / 12.0));
  return std::min(std::max(q, 1), B_PITCH_BINS - 1);
}

}  // namespace

// ---- device-resident parameter blobs (multi-GPU load) ------------------------
// Rank 0 reads and packs a model file once; the other ranks allocate an empty blob of the same size, the caller
// broadcasts device memory to device memory (RCCL over xGMI) and marks the model ready: what travels is exactly
// what the kernels read (MFMA-fragment order), no host round trip and no repacking on the receivers.
namespace {
template <class Model, class Weights>
int model_blob(Model* m, int allocate, void** d_ptr, size_t* n_bytes) {
  if (!m || !d_ptr || !n_bytes) return -1;
  const DeviceScope dev_(m->device);
  const size_t n = Weights::n_floats();
  if (allocate && !m->loaded && m->blob.n_floats != n) {
    m->blob.release();
    if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&m->blob.d), n * sizeof(float)), "blob alloc")) return -2;
    m->blob.n_floats = n;
  }
  if (!m->blob.d || m->blob.n_floats != n) return -1;
  *d_ptr = m->blob.d;
  *n_bytes = n * sizeof(float);
  return 0;
}
template <class Model>
int model_ready(Model* m) {
  if (!m || !m->blob.d) return -1;
  const DeviceScope dev_(m->device);
  if (!hip_ok(hipDeviceSynchronize(), "blob ready")) return -2;
  m->w.bind(m->blob.d);
  m->loaded = true;
  return 0;
}
}  // namespace

extern "C" {
static bool rb_step(BeatriceBatch* b);
static void rb_release(BeatriceBatch* b);

// ---- memory loaders ---------------------------------------------------------------------------
#define BHIP_MEMORY_LOADER(Name, Obj, KIND, Weights)                                                        \
  Beatrice_ErrorCode BeatriceHip_Load##Name##FromMemory(Obj* m, const void* bytes, size_t size) {           \
    std::vector<float> host;                                                                                \
    const Beatrice_ErrorCode e = parse_model_bytes(static_cast<const unsigned char*>(bytes), size, KIND,   \
                                                   (long)Weights::n_floats(), &host);                       \
    if (e) return e;                                                                                        \
    const DeviceScope dev_(m->device);                                                                      \
    m->loaded = false;                                                                                      \
    Weights::pack_host(host.data());                                                                        \
    if (!m->blob.upload(host.data(), host.size())) return Beatrice_kFileOpenError;                          \
    m->w.bind(m->blob.d);                                                                                   \
    m->loaded = true;                                                                                       \
    return Beatrice_kSuccess;                                                                               \
  }
BHIP_MEMORY_LOADER(PhoneExtractor, Beatrice20rc0_PhoneExtractor, KIND_PHONE, PhoneWeights)
BHIP_MEMORY_LOADER(PitchEstimator, Beatrice20rc0_PitchEstimator, KIND_PITCH, PitchWeights)
BHIP_MEMORY_LOADER(WaveformGenerator, Beatrice20rc0_WaveformGenerator, KIND_WAVE, WaveWeights)
BHIP_MEMORY_LOADER(EmbeddingSetter, Beatrice20rc0_EmbeddingSetter, KIND_EMBED, EmbedWeights)

int BeatriceHip_SetDevice(int ordinal) {
  int n = 0;
  if (ordinal < -1 || (ordinal >= 0 && (hipGetDeviceCount(&n) != hipSuccess || ordinal >= n))) return -1;
  set_target_device(ordinal);
  return 0;
}
int BeatriceHip_GetDevice(void) { return target_device(); }
int BeatriceBatch_Device(const BeatriceBatch* b) { return b ? b->device : -1; }

int BeatriceHip_ModelBlob(int kind, void* model, int allocate, void** d_ptr, size_t* n_bytes) {
  switch (kind) {
    case KIND_PHONE: return model_blob<Beatrice20rc0_PhoneExtractor, PhoneWeights>(static_cast<Beatrice20rc0_PhoneExtractor*>(model), allocate, d_ptr, n_bytes);
    case KIND_PITCH: return model_blob<Beatrice20rc0_PitchEstimator, PitchWeights>(static_cast<Beatrice20rc0_PitchEstimator*>(model), allocate, d_ptr, n_bytes);
    case KIND_WAVE: return model_blob<Beatrice20rc0_WaveformGenerator, WaveWeights>(static_cast<Beatrice20rc0_WaveformGenerator*>(model), allocate, d_ptr, n_bytes);
    case KIND_EMBED: return model_blob<Beatrice20rc0_EmbeddingSetter, EmbedWeights>(static_cast<Beatrice20rc0_EmbeddingSetter*>(model), allocate, d_ptr, n_bytes);
    default: return -1;
  }
}
int BeatriceHip_ModelBlobReady(int kind, void* model) {
  switch (kind) {
    case KIND_PHONE: return model_ready(static_cast<Beatrice20rc0_PhoneExtractor*>(model));
    case KIND_PITCH: return model_ready(static_cast<Beatrice20rc0_PitchEstimator*>(model));
    case KIND_WAVE: return model_ready(static_cast<Beatrice20rc0_WaveformGenerator*>(model));
    case KIND_EMBED: return model_ready(static_cast<Beatrice20rc0_EmbeddingSetter*>(model));
    default: return -1;
  }
}

// ---- lifecycle ----------------------------------------------------------------------------------
BeatriceBatch* BeatriceBatch_Create(const Beatrice20rc0_PhoneExtractor* phone, const Beatrice20rc0_PitchEstimator* pitch,
                                    const Beatrice20rc0_WaveformGenerator* wave, const Beatrice20rc0_EmbeddingSetter* embed,
                                    int n_streams, int max_speakers) {
  return BeatriceBatch_CreateBlock(phone, pitch, wave, embed, n_streams, max_speakers, 1);
}

BeatriceBatch* BeatriceBatch_CreateBlock(const Beatrice20rc0_PhoneExtractor* phone, const Beatrice20rc0_PitchEstimator* pitch,
                                         const Beatrice20rc0_WaveformGenerator* wave, const Beatrice20rc0_EmbeddingSetter* embed,
                                         int n_streams, int max_speakers, int hops_per_step) {
  auto* b = new BeatriceBatch();
  if (!phone || !pitch || !wave || !embed || !phone->loaded || !pitch->loaded || !wave->loaded || !embed->loaded ||
      n_streams < 1 || max_speakers < 1 || (hops_per_step != 1 && hops_per_step != 2 && hops_per_step != 4 && hops_per_step != 8))
    return b;  // unhealthy object; every call on it fails with -2
  if (pitch->device != phone->device || wave->device != phone->device || embed->device != phone->device) return b;  // one GPU per batch
  b->device = phone->device;
  const DeviceScope dev_(b->device);
  b->phone_m = phone; b->pitch_m = pitch; b->wave_m = wave; b->embed_m = embed;
  b->B = n_streams; b->max_speakers = max_speakers; b->H = hops_per_step;
  const int B = n_streams, S = max_speakers, H = hops_per_step;
  bool ok = make_stage_stream(&b->stream, 0);
  b->owns_stream = ok;
  ok = ok && hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_in), sizeof(float) * B * H * B_IN_HOP), "d_in") &&
       hip_ok(hipMemset(b->d_in, 0, sizeof(float) * B * H * B_IN_HOP), "d_in0");
  // the front end's outputs (phone vector, conditioning mix) have three step slots: see `pipelined`
  const bool slack = H == 1;  // rings sized so that every layer can be its own pipeline stage (tick.hip.h)
  ok = ok && b->phone.create(B, H, b->d_in, 3, slack) && b->pitch.create(B, H, b->d_in, true, slack) &&
       b->wave.create(B, H, S, S, 9, b->phone.d_phone, b->pitch.d_q, b->pitch.d_feat, 3, slack);
  b->wave.q_slots = b->pitch.q_slots;
  // The modules advance in lockstep: one step counter, with no launch spent on incrementing it.  The step's
  // first kernels (phone.f1, pitch.fft) read the pair {counter, I/O slot} from d_hop_next; phone.f1 publishes
  // it to phone.d_hop for the rest of the front end and to d_hop_wave[counter & 1] for the waveform generator
  // (which may lag one step behind); the front end's last body (wave.cond) stores the next pair to d_hop_next.
  ok = ok && hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_hop_next), 2 * sizeof(int)), "hop_next") &&
       hip_ok(hipMemset(b->d_hop_next, 0, 2 * sizeof(int)), "hop_next0") &&
       hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_hop_wave), 8 * sizeof(int)), "hop_wave") &&
       hip_ok(hipMemset(b->d_hop_wave, 0, 8 * sizeof(int)), "hop_wave0");
  b->pitch.hop = b->phone.d_hop; b->wave.hop = b->d_hop_wave;
  b->phone.hop_in = b->d_hop_next; b->pitch.hop_in = b->d_hop_next;
  b->phone.hop_publish = b->phone.d_hop; b->phone.hop_publish_wave = b->d_hop_wave;
  b->wave.front_hop = b->phone.d_hop; b->wave.front_next_out = b->d_hop_next;
  b->phone.advance_hop = false; b->pitch.advance_hop = false; b->wave.advance_hop = false;
  for (int s = 0; s < BeatriceBatch::kMaxStages && ok; ++s) {  // (stage streams are created when a pipeline depth asks for them)
    for (int k = 0; k < BeatriceBatch::kSlots && ok; ++k) ok = hip_ok(hipEventCreateWithFlags(&b->ev_done[s][k], hipEventDisableTiming), "ev");
  }
  set_plan(b, 1);
  const size_t cbf = (size_t)S * B_CODEBOOK * B_PHONE_CH, kvf = (size_t)S * B_KV_LEN * B_KV_CH;
  ok = ok && hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_cb_raw), sizeof(float) * cbf), "cb") &&
       hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_cbT), sizeof(float) * cbf), "cbT") &&
       hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_cnorm), sizeof(float) * S * B_CODEBOOK), "cnorm") &&
       hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_add_raw), sizeof(float) * S * B_HID), "add") &&
       hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_frm_raw), sizeof(float) * 9 * B_HID), "frm") &&
       hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_kv_raw), sizeof(float) * kvf), "kv");
  ok = ok && hip_ok(hipMemset(b->d_cbT, 0, sizeof(float) * cbf), "cbT0") && hip_ok(hipMemset(b->d_cnorm, 0, sizeof(float) * S * B_CODEBOOK), "cn0") &&
       // entries the caller never fills (a morph entry's codebook, speakers added later) project to zeros, not to garbage
       hip_ok(hipMemset(b->d_cb_raw, 0, sizeof(float) * cbf), "cb0") && hip_ok(hipMemset(b->d_add_raw, 0, sizeof(float) * S * B_HID), "add0") &&
       hip_ok(hipMemset(b->d_frm_raw, 0, sizeof(float) * 9 * B_HID), "frm0") && hip_ok(hipMemset(b->d_kv_raw, 0, sizeof(float) * kvf), "kv0");
  b->cfg.assign(B, StreamCfg());
  b->morph.assign(S, MorphSlot());
  b->lottery.resize(B);
  for (int s = 0; s < B; ++s) b->lottery[s].seed(5489u + (unsigned)s);
  {  // layout of the settings block (256-byte aligned arrays)
    size_t o = 0;
    auto take = [&o](size_t bytes) { const size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    const size_t nt = ok ? (size_t)b->wave.n_tiles_max : 1;
    b->off.cbT = take(sizeof(float*) * B * H); b->off.cnorm = take(sizeof(float*) * B * H); b->off.vqk = take(sizeof(int) * B);
    // (grouped by consumer: k-NN | pitch head | conditioning mix -- tick mode copies a consumer's range as one piece)
    b->off.min_q = take(sizeof(int) * B); b->off.max_q = take(sizeof(int) * B); b->off.params = take(sizeof(PitchParams) * B);
    b->off.add_idx = take(sizeof(int) * B); b->off.frm_idx = take(sizeof(int) * B);
    b->off.front_bytes = o;
    for (int blk = 0; blk < B_NBLOCKS; ++blk) {
      b->off.perm[blk] = take(sizeof(int) * nt * 16); b->off.tile_slot[blk] = take(sizeof(int) * nt);
      // tick mode: the same rows as quads of <= 4 rows per K/V slot, four quads to a tile (rowchain.hip.h block_b_body)
      b->off.qperm[blk] = take(sizeof(int) * nt * 16); b->off.qslot[blk] = take(sizeof(int) * nt * 4);
    }
    b->off.wave_bytes = o - b->off.front_bytes;
  }
  const size_t dev_bytes = b->off.front_bytes + BeatriceBatch::kSlots * b->off.wave_bytes;
  ok = ok && b->settings.alloc_host(b->off.front_bytes + b->off.wave_bytes) &&
       hip_ok(hipMalloc(reinterpret_cast<void**>(&b->settings.d), dev_bytes), "settings") &&
       hip_ok(hipMemset(b->settings.d, 0, dev_bytes), "settings0");
  if (ok) {  // the kernels read the block instead of the modules' own arrays
    void** keep = b->module_owned;
    auto swap_in = [&keep](auto*& member, auto* view) { *keep++ = (void*)member; member = view; };
    swap_in(b->phone.d_cbT, b->dev_view<const float*>(b->off.cbT));
    swap_in(b->phone.d_cnorm, b->dev_view<const float*>(b->off.cnorm));
    swap_in(b->phone.d_vqk, b->dev_view<int>(b->off.vqk));
    swap_in(b->pitch.d_min_q, b->dev_view<int>(b->off.min_q));
    swap_in(b->pitch.d_max_q, b->dev_view<int>(b->off.max_q));
    swap_in(b->pitch.d_params, b->dev_view<PitchParams>(b->off.params));
    swap_in(b->wave.d_add_idx, b->dev_view<int>(b->off.add_idx));
    swap_in(b->wave.d_frm_idx, b->dev_view<int>(b->off.frm_idx));
    for (int blk = 0; blk < B_NBLOCKS; ++blk) {
      swap_in(b->wave.d_perm[blk], b->dev_view<int>(b->off.perm[blk]));
      swap_in(b->wave.d_tile_slot[blk], b->dev_view<int>(b->off.tile_slot[blk]));
    }
  }
  ok = ok && hip_ok(hipHostMalloc(reinterpret_cast<void**>(&b->h_in), sizeof(float) * B * H * B_IN_HOP, hipHostMallocDefault), "h_in") &&
       hip_ok(hipHostMalloc(reinterpret_cast<void**>(&b->h_out), sizeof(float) * B * H * B_OUT_HOP, hipHostMallocDefault), "h_out") &&
       hip_ok(hipEventCreate(&b->ev0), "ev0") && hip_ok(hipEventCreate(&b->ev1), "ev1");
  {  // 48 kHz wrapper: 33-entry Hann-windowed sinc tables of the ratio-1/1 resampler pair
    //   (reference resample.h:209-230 with cutoffs 0.99*16000/48000 in, 0.99*24000/48000 out, :412-417)
    float cd[33], cu[33];
    const double pi = 3.14159265358979323846, cut_d = 0.99 * 16000.0 / 48000.0, cut_u = 0.99 * 24000.0 / 48000.0;
    auto sinc = [&](double x) { return std::abs(x) < 1e-8 ? 1.0 : std::sin(x * pi) / (x * pi); };
    for (int i = 0; i < 33; ++i) {
      const double x = static_cast<double>(i - 16) / 1.0;
      const double hann = 0.5 - 0.5 * std::cos(pi * 2.0 / 32.0 * static_cast<double>(i));
      cd[i] = static_cast<float>(cut_d * sinc(x * cut_d) * hann);
      cu[i] = static_cast<float>(cut_u * sinc(x * cut_u) * hann);
    }
    ok = ok && hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_w48), sizeof(Wrap48State) * B), "w48") &&
         hip_ok(hipMemset(b->d_w48, 0, sizeof(Wrap48State) * B), "w48 0") &&
         hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_coef_down), sizeof(cd)), "cd") &&
         hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_coef_up), sizeof(cu)), "cu") &&
         hip_ok(hipMemcpy(b->d_coef_down, cd, sizeof(cd), hipMemcpyHostToDevice), "cd up") &&
         hip_ok(hipMemcpy(b->d_coef_up, cu, sizeof(cu), hipMemcpyHostToDevice), "cu up") &&
         hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_io48), sizeof(float) * B * 4 * 480), "io48") &&
         hip_ok(hipHostMalloc(reinterpret_cast<void**>(&b->h_io48), sizeof(float) * B * 4 * 480, hipHostMallocDefault), "hio48");
  }
  ok = ok && hip_ok(hipDeviceSynchronize(), "create sync");  // NULL-stream memsets vs the non-blocking stream
  b->ok = ok;
  if (ok) {
    for (int blk = 0; blk < B_NBLOCKS; ++blk) b->row_slot[blk].assign((size_t)B * H, 0);
    for (int s = 0; s < B; ++s) { sync_stream_arrays(b, s); fill_row_slots(b, s); }
    for (int blk = 0; blk < B_NBLOCKS; ++blk) rebuild_tiles(b, blk);
  }
  return b;
}

void BeatriceBatch_Destroy(BeatriceBatch* b) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b) return;
  if (b->stream) (void)sync_all(b);
  drop_graph(b);
  { void* wr[] = {b->d_wrap, b->d_wrap_taps, b->d_wrap_inner, b->d_wrap_io, b->wrap_gains.d}; for (void* p : wr) if (p) (void)hipFree(p); }
  if (b->h_wrap_io) (void)hipHostFree(b->h_wrap_io);
  b->wrap_gains.release();
  { void* tk[] = {b->tk.d_table, b->tk.d_snap, b->tk.d_trace}; for (void* p : tk) if (p) (void)hipFree(p); }
  if (b->tk.h_stage) (void)hipHostFree(b->tk.h_stage);
  for (hipEvent_t e : b->tk.stage_ev) if (e) (void)hipEventDestroy(e);
  host_stream_free(b);
  if (b->r48.d_in16) (void)hipFree(b->r48.d_in16);
  if (b->r48.d_out24) (void)hipFree(b->r48.d_out24);
  rb_release(b);
  if (b->own_d_out) { b->wave.d_out = b->own_d_out; b->own_d_out = nullptr; }
  if (b->io_mapped) { b->wave.d_out = b->dev_d_out; b->phone.d_in = b->pitch.d_in = b->d_in; b->io_mapped = false; }  // (the modules free what they allocated)
  if (b->module_owned[0]) {  // hand the modules their own arrays back so that destroy() frees what it allocated
    void** keep = b->module_owned;
    auto swap_out = [&keep](auto*& member) { member = static_cast<std::remove_reference_t<decltype(member)>>(*keep++); };
    swap_out(b->phone.d_cbT); swap_out(b->phone.d_cnorm); swap_out(b->phone.d_vqk);
    swap_out(b->pitch.d_min_q); swap_out(b->pitch.d_max_q); swap_out(b->pitch.d_params);
    swap_out(b->wave.d_add_idx); swap_out(b->wave.d_frm_idx);
    for (int blk = 0; blk < B_NBLOCKS; ++blk) { swap_out(b->wave.d_perm[blk]); swap_out(b->wave.d_tile_slot[blk]); }
  }
  b->phone.destroy(); b->pitch.destroy(); b->wave.destroy();
  void* dev[] = {b->d_in, b->d_cb_raw, b->d_cbT, b->d_cnorm, b->d_add_raw, b->d_frm_raw, b->d_kv_raw,
                 b->d_w48, b->d_coef_down, b->d_coef_up, b->d_io48, b->d_hop_next, b->d_hop_wave};
  if (b->h_io48) (void)hipHostFree(b->h_io48);
  for (void* p : dev) if (p) (void)hipFree(p);
  b->settings.release();
  if (b->settings.d) (void)hipFree(b->settings.d);
  if (b->h_in) (void)hipHostFree(b->h_in);
  if (b->h_out) (void)hipHostFree(b->h_out);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  for (auto& per_stage : b->ev_done) for (hipEvent_t e : per_stage) if (e) (void)hipEventDestroy(e);
  for (hipStream_t st : b->stage_stream_own) if (st) (void)hipStreamDestroy(st);
  if (b->owns_stream && b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

int BeatriceBatch_IsHealthy(const BeatriceBatch* b) { return b && b->ok ? 1 : 0; }
int BeatriceBatch_NumStreams(const BeatriceBatch* b) { return b ? b->B : 0; }
int BeatriceBatch_HopsPerStep(const BeatriceBatch* b) { return b ? b->H : 0; }
size_t BeatriceBatch_StateBytes(const BeatriceBatch* b) {
  const DeviceScope dev_(b ? b->device : -1);
  return b && b->ok ? sizeof(float) * (b->phone.arena.floats + b->pitch.arena.floats + b->wave.arena.floats) : 0;
}

// ---- speaker tables -----------------------------------------------------------------------------
static bool project_speakers(BeatriceBatch* b, int first, int count) {
  hipStream_t s = b->stream;
  const EmbedWeights& w = b->embed_m->w;
  codebook_prepare(b->d_cb_raw + (size_t)first * B_CODEBOOK * B_PHONE_CH, count,
                   b->d_cbT + (size_t)first * B_PHONE_CH * B_CODEBOOK, b->d_cnorm + (size_t)first * B_CODEBOOK, s);
  embed_project_rows(w.add_w, w.add_b, b->d_add_raw + (size_t)first * B_HID, b->wave.d_add_tab + (size_t)first * B_HID, count, s);
  for (int blk = 0; blk < B_NBLOCKS; ++blk)
    embed_project_kv(w, blk, b->d_kv_raw + (size_t)first * B_KV_LEN * B_KV_CH, count,
                     b->wave.d_kt[blk] + (size_t)first * B_HID * B_KV_LEN, b->wave.d_v[blk] + (size_t)first * B_KV_LEN * B_HID, s,
                     b->wave.d_ktp[blk] ? b->wave.d_ktp[blk] + (size_t)first * B_HID * B_KV_LEN : nullptr,
                     b->wave.d_vp[blk] ? b->wave.d_vp[blk] + (size_t)first * B_KV_LEN * B_HID : nullptr);
  return hip_ok(hipStreamSynchronize(s), "project speakers");
}

int BeatriceBatch_SetSpeakerTables(BeatriceBatch* b, int n, const float* codebooks, const float* additive, const float* formant,
                                   const float* kv) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (n < 1 || n > b->max_speakers || !codebooks || !additive || !formant || !kv) return -1;
  bool ok = sync_all(b) &&
            hip_ok(hipMemcpy(b->d_cb_raw, codebooks, sizeof(float) * n * B_CODEBOOK * B_PHONE_CH, hipMemcpyHostToDevice), "cb") &&
            hip_ok(hipMemcpy(b->d_add_raw, additive, sizeof(float) * n * B_HID, hipMemcpyHostToDevice), "add") &&
            hip_ok(hipMemcpy(b->d_frm_raw, formant, sizeof(float) * 9 * B_HID, hipMemcpyHostToDevice), "frm") &&
            hip_ok(hipMemcpy(b->d_kv_raw, kv, sizeof(float) * n * B_KV_LEN * B_KV_CH, hipMemcpyHostToDevice), "kv");
  if (!ok || !hip_ok(hipDeviceSynchronize(), "tables sync")) return -2;
  b->n_speakers = n;
  for (MorphSlot& m : b->morph) m.active = false;
  b->n_morph_slots = 0;
  const EmbedWeights& w = b->embed_m->w;
  embed_project_rows(w.frm_w, w.frm_b, b->d_frm_raw, b->wave.d_frm_tab, 9, b->stream);
  return project_speakers(b, 0, n) ? 0 : -2;
}

// The four raw tables as they sit on the device, for callers that fill them device-to-device (a broadcast from the
// rank that read the file): [0] codebooks [S][512][128], [1] additive [S][256], [2] formant [9][256], [3] key/value
// [S][384][128]; then BeatriceBatch_ProjectSpeakerTables(b, n) does what SetSpeakerTables does after its upload.
int BeatriceBatch_SpeakerTablesDevice(BeatriceBatch* b, void** d_ptrs, size_t* n_bytes) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (!d_ptrs || !n_bytes) return -1;
  const size_t S = (size_t)b->max_speakers;
  d_ptrs[0] = b->d_cb_raw; n_bytes[0] = sizeof(float) * S * B_CODEBOOK * B_PHONE_CH;
  d_ptrs[1] = b->d_add_raw; n_bytes[1] = sizeof(float) * S * B_HID;
  d_ptrs[2] = b->d_frm_raw; n_bytes[2] = sizeof(float) * 9 * B_HID;
  d_ptrs[3] = b->d_kv_raw; n_bytes[3] = sizeof(float) * S * B_KV_LEN * B_KV_CH;
  return 0;
}
int BeatriceBatch_ProjectSpeakerTables(BeatriceBatch* b, int n) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (n < 1 || n > b->max_speakers) return -1;
  if (!sync_all(b) || !hip_ok(hipDeviceSynchronize(), "tables sync")) return -2;
  b->n_speakers = n;
  for (MorphSlot& m : b->morph) m.active = false;
  b->n_morph_slots = 0;
  const EmbedWeights& w = b->embed_m->w;
  embed_project_rows(w.frm_w, w.frm_b, b->d_frm_raw, b->wave.d_frm_tab, 9, b->stream);
  return project_speakers(b, 0, n) ? 0 : -2;
}

int BeatriceBatch_UpdateSpeaker(BeatriceBatch* b, int spk, const float* codebook, const float* additive, const float* kv) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (spk < 0 || spk >= b->max_speakers) return -1;
  bool ok = sync_all(b);
  if (codebook) ok = ok && hip_ok(hipMemcpy(b->d_cb_raw + (size_t)spk * B_CODEBOOK * B_PHONE_CH, codebook, sizeof(float) * B_CODEBOOK * B_PHONE_CH, hipMemcpyHostToDevice), "cb1");
  if (additive) ok = ok && hip_ok(hipMemcpy(b->d_add_raw + (size_t)spk * B_HID, additive, sizeof(float) * B_HID, hipMemcpyHostToDevice), "add1");
  if (kv) ok = ok && hip_ok(hipMemcpy(b->d_kv_raw + (size_t)spk * B_KV_LEN * B_KV_CH, kv, sizeof(float) * B_KV_LEN * B_KV_CH, hipMemcpyHostToDevice), "kv1");
  if (!ok || !hip_ok(hipDeviceSynchronize(), "speaker sync")) return -2;
  if (spk >= b->n_speakers) b->n_speakers = spk + 1;
  if (b->morph[spk].active) { b->morph[spk].active = false; --b->n_morph_slots; }  // the caller's data replaces a morph
  return project_speakers(b, spk, 1) ? 0 : -2;
}

// ---- speaker morphing ----------------------------------------------------------------------------
// Weight preparation as the reference host does it (voice_morph_state.h:87-104: entries below 0.01
// dropped; processor_core_2.cc:507-532: the eight largest kept, in descending order), then the additive
// and the 384 key/value embeddings of entry `slot` become weighted spherical means computed on the
// device (morph.hip), and their projections are refreshed.
static int morph_into(BeatriceBatch* b, int slot, const float* weights, int n_weights, unsigned seed) {
  if (!weights || n_weights < 1 || n_weights > 256 || slot < n_weights || slot >= b->max_speakers || n_weights > b->n_speakers) return -1;
  std::vector<float> w(weights, weights + n_weights);
  for (float& v : w) if (v < 0.01f) v = 0.0f;
  std::vector<int> order(n_weights);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&w](const int x, const int y) -> bool { return w[x] > w[y]; });
  const int keep = std::min(n_weights, 8);
  // SphericalAverage::SetWeights (spherical_average.h:142-199): points in `order` until the first zero weight
  int n_active = 0, spk[8];
  float wn[8], sum = 0.0f;
  for (int i = 0; i < keep; ++i) {
    if (w[order[i]] == 0.0f) break;
    spk[n_active] = order[i]; wn[n_active] = w[order[i]]; ++n_active;
  }
  for (int i = 0; i < n_active; ++i) sum += wn[i];
  if (n_active > 0 && sum > 0.0f) { const float inv = 1.0f / sum; for (int i = 0; i < n_active; ++i) wn[i] *= inv; }
  else n_active = 0;
  bool ok = sync_all(b);
  ok = ok && spherical_mean_rows(b->d_add_raw, B_HID, 1, B_HID, n_active, spk, wn, b->d_add_raw + (size_t)slot * B_HID, b->stream);
  ok = ok && spherical_mean_rows(b->d_kv_raw, (size_t)B_KV_LEN * B_KV_CH, B_KV_LEN, B_KV_CH, n_active, spk, wn,
                                 b->d_kv_raw + (size_t)slot * B_KV_LEN * B_KV_CH, b->stream);
  if (!ok) return -2;
  if (slot >= b->n_speakers) b->n_speakers = slot + 1;
  if (!project_speakers(b, slot, 1)) return -2;
  MorphSlot& m = b->morph[slot];
  if (!m.active) ++b->n_morph_slots;
  m.active = true;
  m.n_speakers = n_weights;
  m.n_odds = keep;
  for (int i = 0; i < 8; ++i) { m.order[i] = i < keep ? order[i] : 0; m.odds[i] = i < keep ? w[order[i]] : 0.0f; }
  // The engines are seeded ONCE per batch, as the reference seeds its engine once per instance (processor_core_2.h:48,145) and
  // never again when morph weights move (:94-121): only the first morph of the batch's life applies `seed`; a caller that
  // moves weights every step keeps each stream's draw SEQUENCE running, and a morph on one entry leaves the draws of the
  // streams on other entries alone.  BeatriceBatch_SeedLottery re-seeds explicitly.
  if (!b->lottery_seeded) {
    for (int st = 0; st < b->B; ++st) b->lottery[st].seed(seed + (unsigned)st);
    b->lottery_seeded = true;
  }
  return 0;
}
int BeatriceBatch_MorphSpeaker(BeatriceBatch* b, int slot, const float* weights, int n_weights, unsigned seed) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (const int rc = morph_into(b, slot, weights, n_weights, seed)) return rc;
  // streams already on this entry re-install its key/value blocks, one per hop, like after a speaker switch
  for (StreamCfg& c : b->cfg) if (c.target_speaker == slot) { c.kv_set_count = 0; c.kv_delay = 0; }
  b->pending_kv = 0;
  for (const StreamCfg& c : b->cfg) if (c.kv_set_count < B_NBLOCKS) ++b->pending_kv;
  return 0;
}
// The reference's timeline of a weight change for streams that are ALREADY morphing (processor_core_2.cc:51-177): on the next
// hop h0 the new additive embedding is in force and the codebook lottery draws with the new odds; the key/value means are
// computed a quarter per hop over h0 .. h0+3 while the OLD key/value blocks keep playing; at h0+4 the new embeddings are
// registered and installed one block per hop, h0+4 .. h0+7.  In place that cannot be done (BeatriceBatch_MorphSpeaker
// overwrites the entry every stream on it is reading), so the new morph goes into ANOTHER table entry `slot` and the streams
// on `from_slot` move over: additive embedding and lottery at once, key/value blocks after four hops.  `slot` must not be an
// entry whose key/value blocks some stream still has installed (-3: use a third entry, as a caller that moves the weights
// faster than every eight hops needs anyway -- the reference never installs new blocks while the weights keep moving, and
// neither do streams here: each call restarts their four-hop wait).
int BeatriceBatch_MorphSpeakerStaged(BeatriceBatch* b, int slot, int from_slot, const float* weights, int n_weights, unsigned seed) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (from_slot < 0 || from_slot >= b->max_speakers || from_slot == slot) return -1;
  if (slot >= 0 && slot < b->max_speakers)
    for (const StreamCfg& c : b->cfg)
      for (int blk = 0; blk < B_NBLOCKS; ++blk) if (c.kv_slot[blk] == slot) return -3;
  if (const int rc = morph_into(b, slot, weights, n_weights, seed)) return rc;
  for (int s = 0; s < b->B; ++s) {
    StreamCfg& c = b->cfg[s];
    if (c.target_speaker != from_slot) continue;
    c.target_speaker = slot;
    c.additive_speaker = slot;
    c.kv_set_count = 0;
    c.kv_delay = 4;
    sync_stream_arrays(b, s);
  }
  b->pending_kv = 0;
  for (const StreamCfg& c : b->cfg) if (c.kv_set_count < B_NBLOCKS) ++b->pending_kv;
  return 0;
}
// the codebook lottery's engine of one stream (or of all, -1): std::mt19937(seed), e.g. a value derived from the
// stream's global identity when streams are sharded over several batches / GPUs
int BeatriceBatch_SeedLottery(BeatriceBatch* b, int stream, unsigned seed) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (stream < -1 || stream >= b->B) return -1;
  for (int s = (stream < 0 ? 0 : stream); s < (stream < 0 ? b->B : stream + 1); ++s) b->lottery[s].seed(seed);
  b->lottery_seeded = true;  // a later BeatriceBatch_MorphSpeaker keeps these engines
  return 0;
}
// copies the morphed entry's raw embeddings back (test / inspection hook; any pointer may be NULL)
int BeatriceBatch_GetSpeakerEmbeddings(BeatriceBatch* b, int speaker, float* additive, float* key_value) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (speaker < 0 || speaker >= b->max_speakers) return -1;
  bool ok = sync_all(b);
  if (additive) ok = ok && hip_ok(hipMemcpy(additive, b->d_add_raw + (size_t)speaker * B_HID, sizeof(float) * B_HID, hipMemcpyDeviceToHost), "add");
  if (key_value) ok = ok && hip_ok(hipMemcpy(key_value, b->d_kv_raw + (size_t)speaker * B_KV_LEN * B_KV_CH, sizeof(float) * B_KV_LEN * B_KV_CH, hipMemcpyDeviceToHost), "kv");
  return ok ? 0 : -2;
}

// ---- per-stream settings (reference ProcessorCore2 setters) ------------------------------------
// processor_core_2.cc:431-466: codebook + additive switch at once, K/V re-registered and installed
// one block per following hop.
int BeatriceBatch_SetTargetSpeaker(BeatriceBatch* b, int stream, int speaker) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (speaker < 0 || speaker >= b->max_speakers) return -1;
  const int r = for_streams(b, stream, [&](StreamCfg& c) {
    c.target_speaker = speaker; c.codebook_speaker = speaker; c.additive_speaker = speaker; c.kv_set_count = 0; c.kv_delay = 0;
    for (int& cr : c.codebook_row) cr = speaker;
  });
  if (r == 0) { b->pending_kv = 0; for (const StreamCfg& c : b->cfg) if (c.kv_set_count < B_NBLOCKS) ++b->pending_kv; }
  return r;
}
// processor_core_2.cc:270,414: `while (SetKeyValueSpeakerEmbedding());`
int BeatriceBatch_FlushSpeaker(BeatriceBatch* b, int stream) {
  const DeviceScope dev_(b ? b->device : -1);
  const int r = for_streams(b, stream, [&](StreamCfg& c) {
    c.kv_delay = 0;
    for (; c.kv_set_count < B_NBLOCKS; ++c.kv_set_count) c.kv_slot[c.kv_set_count] = c.target_speaker;
  });
  if (r == 0) {
    for (int s = (stream < 0 ? 0 : stream); s < (stream < 0 ? b->B : stream + 1); ++s) fill_row_slots(b, s);
    b->pending_kv = 0;
    for (const StreamCfg& c : b->cfg) if (c.kv_set_count < B_NBLOCKS) ++b->pending_kv;
    for (int blk = 0; blk < B_NBLOCKS; ++blk) rebuild_tiles(b, blk);
  }
  return r;
}
// processor_core_2.cc:468-481
int BeatriceBatch_SetFormantShift(BeatriceBatch* b, int stream, double shift) {
  const DeviceScope dev_(b ? b->device : -1);
  shift = std::min(std::max(shift, -2.0), 2.0);
  const int idx = (int)std::round(shift * 2.0 + 4.0);
  return for_streams(b, stream, [&](StreamCfg& c) { c.formant_index = idx; });
}
// processor_core_2.cc:585-590
int BeatriceBatch_SetVQNumNeighbors(BeatriceBatch* b, int stream, int k) {
  const DeviceScope dev_(b ? b->device : -1);
  k = std::min(std::max(k, 0), 8);
  if (b) b->vq_dirty = true;
  return for_streams(b, stream, [&](StreamCfg& c) { c.vq_k = k; });
}
int BeatriceBatch_SetMinSourcePitch(BeatriceBatch* b, int stream, double note) {
  const DeviceScope dev_(b ? b->device : -1);
  const int q = midi_to_bin(note);
  return for_streams(b, stream, [&](StreamCfg& c) { c.min_q = q; });
}
int BeatriceBatch_SetMaxSourcePitch(BeatriceBatch* b, int stream, double note) {
  const DeviceScope dev_(b ? b->device : -1);
  const int q = midi_to_bin(note);
  return for_streams(b, stream, [&](StreamCfg& c) { c.max_q = q; });
}
// processor_core_2.cc:483-486, 534-559
int BeatriceBatch_SetPitchShift(BeatriceBatch* b, int stream, double v) {
  const DeviceScope dev_(b ? b->device : -1);
  v = std::min(std::max(v, -24.0), 24.0);
  return for_streams(b, stream, [&](StreamCfg& c) { c.pitch.pitch_shift = v; });
}
int BeatriceBatch_SetAverageSourcePitch(BeatriceBatch* b, int stream, double v) {
  const DeviceScope dev_(b ? b->device : -1);
  v = std::min(std::max(v, 0.0), 128.0);
  return for_streams(b, stream, [&](StreamCfg& c) { c.pitch.average_source_pitch = v; });
}
int BeatriceBatch_SetIntonationIntensity(BeatriceBatch* b, int stream, double v) {
  const DeviceScope dev_(b ? b->device : -1);
  return for_streams(b, stream, [&](StreamCfg& c) { c.pitch.intonation_intensity = v; });
}
int BeatriceBatch_SetPitchCorrection(BeatriceBatch* b, int stream, double v) {
  const DeviceScope dev_(b ? b->device : -1);
  v = std::min(std::max(v, 0.0), 1.0);
  return for_streams(b, stream, [&](StreamCfg& c) { c.pitch.pitch_correction = v; });
}
int BeatriceBatch_SetPitchCorrectionType(BeatriceBatch* b, int stream, int type) {
  const DeviceScope dev_(b ? b->device : -1);
  if (type < 0 || type > 1) return -1;
  return for_streams(b, stream, [&](StreamCfg& c) { c.pitch.pitch_correction_type = type; });
}
// processor_core_2.cc:258-291: fresh contexts, then speaker (all four blocks at once) and the other
// settings re-applied -- here the settings persist per stream, only the state is zeroed.
int BeatriceBatch_ResetStream(BeatriceBatch* b, int stream) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (stream < -1 || stream >= b->B) return -1;
  const int lo = stream < 0 ? 0 : stream, hi = stream < 0 ? b->B : stream + 1;
  bool ok = !(b->pipelined || b->tk.on) || sync_all(b);  // later stages of earlier steps may still be running (own streams / later ticks)
  for (int s = lo; s < hi && ok; ++s) {
    ok = b->phone.arena.zero_stream(s, b->stream) && b->pitch.arena.zero_stream(s, b->stream) &&
         b->wave.arena.zero_stream(s, b->stream) &&
         hip_ok(hipMemsetAsync(b->pitch.d_prev_q + s, 0, sizeof(int), b->stream), "prev_q") &&
         hip_ok(hipMemsetAsync(b->d_w48 + s, 0, sizeof(Wrap48State), b->stream), "w48 reset");
  }
  if (!ok) return -2;
  return BeatriceBatch_FlushSpeaker(b, stream);
}

// ---- per-hop ------------------------------------------------------------------------------------
int BeatriceBatch_ConvertFramesDevice(BeatriceBatch* b, const float* d_in, float* d_out) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  return step_device(b, d_in, d_out) ? 0 : -2;
}
// Resident I/O: the caller keeps n_slots steps of input and output on the device,
//   d_in [n_slots][B][H*160], d_out [n_slots][B][H*240];
// step k (BeatriceBatch_ConvertFramesDevice(b, NULL, NULL)) reads slot k mod n_slots and writes the same
// slot of d_out, with no copy: the slot index lives next to the step counter in device memory and is
// advanced by the last kernel, so the captured graph stays valid.  NULL pointers unbind.
int BeatriceBatch_BindResidentIO(BeatriceBatch* b, const float* d_in, float* d_out, int n_slots) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  const bool bind = d_in != nullptr || d_out != nullptr;
  if (bind && (!d_in || !d_out || n_slots < 1)) return -1;
  if (b->tk.on) return -1;  // leave tick mode first
  if (!set_io_mapped(b, false) || !sync_all(b)) return -2;
  b->io_host = 0;
  drop_graph(b);  // kernel arguments change
  if (b->own_d_out) { b->wave.d_out = b->own_d_out; b->own_d_out = nullptr; }
  b->phone.d_in = b->pitch.d_in = b->d_in;
  b->phone.io_stride = b->pitch.io_stride = b->wave.io_stride = 0;
  b->wave.io_slots = b->io_slots = 0;
  if (bind) {
    b->own_d_out = b->wave.d_out;
    b->wave.d_out = d_out;
    b->phone.d_in = b->pitch.d_in = const_cast<float*>(d_in);
    b->phone.io_stride = b->pitch.io_stride = (size_t)b->B * b->H * B_IN_HOP;
    b->wave.io_stride = (size_t)b->B * b->H * B_OUT_HOP;
    b->wave.io_slots = b->io_slots = n_slots;
  }
  const int zero = 0;  // the next step starts at slot 0
  return hip_ok(hipMemcpy(b->d_hop_next + 1, &zero, sizeof(int), hipMemcpyHostToDevice), "slot0") ? 0 : -2;
}

int BeatriceBatch_ConvertFrames(BeatriceBatch* b, const float* in, float* out) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) { if (b && out) std::memset(out, 0, sizeof(float) * b->B * b->H * B_OUT_HOP); return -2; }
  if (b->io_slots > 0 || b->tk.on) return -1;  // resident I/O is bound
  const size_t n_in = (size_t)b->B * b->H * B_IN_HOP, n_out = (size_t)b->B * b->H * B_OUT_HOP;
  b->want_mapped = true;
  bool ok = set_io_mapped(b, true);
  std::memcpy(b->h_in, in, sizeof(float) * n_in);
  if (!b->io_mapped) ok = ok && hip_ok(hipMemcpyAsync(b->d_in, b->h_in, sizeof(float) * n_in, hipMemcpyHostToDevice, b->stream), "in");
  ok = ok && step_device(b, nullptr, nullptr);
  if (!b->io_mapped) ok = ok && hip_ok(hipMemcpyAsync(b->h_out, b->wave.d_out, sizeof(float) * n_out, hipMemcpyDeviceToHost, wave_stream(b)), "out");
  b->want_mapped = false;
  ok = sync_all(b) && ok;
  if (ok) std::memcpy(out, b->h_out, sizeof(float) * n_out);
  else std::memset(out, 0, sizeof(float) * n_out);
  return ok ? 0 : -2;
}
// ---- 48 kHz blocks with the wrapper on the device ----------------------------------------------
static bool step_48k(BeatriceBatch* b, const float* d_in48, float* d_out48, int channels) {
  // the FIFO of the reference emits the PREVIOUS block's model output first (resample.h:346-361)
  hipLaunchKernelGGL(wrap48_post_kernel, dim3(b->B), dim3(256), 0, b->stream, b->d_w48, b->d_coef_up, d_out48, channels);
  hipLaunchKernelGGL(wrap48_pre_kernel, dim3(b->B), dim3(256), 0, b->stream, d_in48, channels, b->d_w48, b->d_coef_down, b->d_in);
  if (!step_device(b, nullptr, nullptr)) return false;
  hipLaunchKernelGGL(wrap48_latch_kernel, dim3((b->B * 240 + 255) / 256), dim3(256), 0, b->stream, b->d_w48, b->wave.d_out, b->B);
  return hip_ok(hipGetLastError(), "wrap48");
}
// Throughput form of the 48 kHz wrapper: n_slots resident 48 kHz blocks per direction, the tick pipeline between them.
// Block k (BeatriceBatch_ConvertBlocks48kDevice(b, NULL, NULL, channels)) is read from slot k mod n_slots; its converted
// block lands in the same slot of d_out48 BeatriceBatch_TickStages() - 1 calls later (or after BeatriceBatch_Synchronize).
// Same samples as the in-order BeatriceBatch_ConvertBlocks48kDevice.  NULL pointers unbind.
int BeatriceBatch_BindResidentIO48k(BeatriceBatch* b, const float* d_in48, float* d_out48, int channels, int n_slots) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::Resident48& r = b->r48;
  if (r.on) {
    if (!sync_all(b)) return -2;
    const int rc = tick_enable(b, false);
    if (rc) return rc;
    (void)BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0);
    if (r.d_in16) (void)hipFree(r.d_in16);
    if (r.d_out24) (void)hipFree(r.d_out24);
    r = BeatriceBatch::Resident48{};
  }
  if (!d_in48 && !d_out48) return 0;
  if (!d_in48 || !d_out48 || channels < 1 || channels > 2 || n_slots < b->tk.plan.count() + 1 || b->H != 1 || b->io_slots > 0 || b->pipelined ||
      b->tk.on || b->hs.on)
    return -1;
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_in16), sizeof(float) * n_slots * b->B * B_IN_HOP), "r48 in16") &&
            hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_out24), sizeof(float) * n_slots * b->B * B_OUT_HOP), "r48 out24") &&
            hip_ok(hipMemset(r.d_in16, 0, sizeof(float) * n_slots * b->B * B_IN_HOP), "r48 zero");
  ok = ok && BeatriceBatch_BindResidentIO(b, r.d_in16, r.d_out24, n_slots) == 0 && tick_enable(b, true) == 0;
  if (!ok) {
    (void)tick_enable(b, false);
    (void)BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0);
    if (r.d_in16) (void)hipFree(r.d_in16);
    if (r.d_out24) (void)hipFree(r.d_out24);
    r = BeatriceBatch::Resident48{};
    return -2;
  }
  r.d_in48 = d_in48; r.d_out48 = d_out48; r.channels = channels; r.n_slots = n_slots; r.on = true;
  return 0;
}
int BeatriceBatch_ConvertBlocks48kDevice(BeatriceBatch* b, const float* d_in, float* d_out, int channels) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (b->r48.on) return (!d_in && !d_out && channels == b->r48.channels) ? (tick_run(b, true) ? 0 : -2) : -1;
  if (channels < 1 || channels > 2 || !d_in || !d_out || b->H != 1 || b->io_slots > 0 || b->pipelined || b->tk.on) return -1;  // per 10 ms block, in order
  return step_48k(b, d_in, d_out, channels) ? 0 : -2;
}
int BeatriceBatch_ConvertBlocks48k(BeatriceBatch* b, const float* in, float* out, int channels) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (channels < 1 || channels > 2 || !in || !out || b->H != 1 || b->io_slots > 0 || b->pipelined || b->tk.on) return -1;
  const size_t n = (size_t)b->B * channels * 480;
  float* h_in = b->h_io48;
  float* h_out = b->h_io48 + (size_t)b->B * 2 * 480;
  float* d_in = b->d_io48;
  float* d_out = b->d_io48 + (size_t)b->B * 2 * 480;
  std::memcpy(h_in, in, sizeof(float) * n);
  bool ok = hip_ok(hipMemcpyAsync(d_in, h_in, sizeof(float) * n, hipMemcpyHostToDevice, b->stream), "in48");
  ok = ok && step_48k(b, d_in, d_out, channels);
  ok = ok && hip_ok(hipMemcpyAsync(h_out, d_out, sizeof(float) * n, hipMemcpyDeviceToHost, b->stream), "out48");
  ok = hip_ok(hipStreamSynchronize(b->stream), "sync") && ok;
  b->inflight = false;
  if (ok) std::memcpy(out, h_out, sizeof(float) * n);
  else std::memset(out, 0, sizeof(float) * n);
  return ok ? 0 : -2;
}

// ---- host-rate blocks with the whole wrapper on the device (wrapper.hip.h) -------------------------------------------
namespace { constexpr int kInnerStride = wrapn::kMaxSamples + 64; }
int BeatriceBatch_ConfigureWrapper(BeatriceBatch* b, double sample_rate) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (b->H != 1) return -1;
  if (!sync_all(b)) return -2;
  if (!b->wrap.configure(sample_rate)) return -1;  // rate <= 0, or a ratio whose filter history exceeds the state block
  const int B = b->B;
  const size_t nt = b->wrap.taps_down.size();
  bool ok = true;
  if (!b->d_wrap) {
    ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_wrap), sizeof(wrapn::StreamState) * B), "wrap state") &&
         hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_wrap_inner), sizeof(float) * B * kInnerStride), "wrap inner") &&
         hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_wrap_io), sizeof(float) * B * 4 * wrapn::kMaxSamples), "wrap io") &&
         hip_ok(hipHostMalloc(reinterpret_cast<void**>(&b->h_wrap_io), sizeof(float) * B * 4 * wrapn::kMaxSamples, hipHostMallocDefault), "wrap io host") &&
         b->wrap_gains.alloc_host(2 * (size_t)B) &&
         hip_ok(hipMalloc(reinterpret_cast<void**>(&b->wrap_gains.d), sizeof(wrapn::GainSeg) * 2 * B), "wrap gains");
    b->gain_in.assign(B, wrapn::GainClock());
    b->gain_out.assign(B, wrapn::GainClock());
  }
  if (b->d_wrap_taps) { (void)hipFree(b->d_wrap_taps); b->d_wrap_taps = nullptr; }
  ok = ok && hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_wrap_taps), sizeof(float) * 2 * nt), "wrap taps") &&
       hip_ok(hipMemcpy(b->d_wrap_taps, b->wrap.taps_down.data(), sizeof(float) * nt, hipMemcpyHostToDevice), "taps down") &&
       hip_ok(hipMemcpy(b->d_wrap_taps + nt, b->wrap.taps_up.data(), sizeof(float) * nt, hipMemcpyHostToDevice), "taps up") &&
       hip_ok(hipMemset(b->d_wrap, 0, sizeof(wrapn::StreamState) * B), "wrap state0") && hip_ok(hipDeviceSynchronize(), "wrap sync");
  b->wrap_gains_constant = false;
  // (a new rate restarts the resampler and the FIFO as the reference's SetSampleRate does; the gains keep their state,
  //  now ramping at the new rate: reference processor_core_2.cc:421-429)
  return ok ? 0 : -2;
}
// reference ProcessorCore2::SetInputGain / SetOutputGain (processor_core_2.cc:488-496): the target; the ramp follows at 2 dB/ms
int BeatriceBatch_SetInputGain(BeatriceBatch* b, int stream, double db) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (stream < -1 || stream >= b->B || b->gain_in.empty()) return -1;
  for (int s = (stream < 0 ? 0 : stream); s < (stream < 0 ? b->B : stream + 1); ++s) b->gain_in[s].target_db = db;
  return 0;
}
int BeatriceBatch_SetOutputGain(BeatriceBatch* b, int stream, double db) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (stream < -1 || stream >= b->B || b->gain_out.empty()) return -1;
  for (int s = (stream < 0 ? 0 : stream); s < (stream < 0 ? b->B : stream + 1); ++s) b->gain_out[s].target_db = db;
  return 0;
}
static bool wrap_chunk(BeatriceBatch* b, const float* d_in, float* d_out, int channels, int n) {
  using namespace wrapn;
  const int B = b->B;
  hipStream_t st = b->stream;
  WrapPlan& w = b->wrap;
  // gains: this call's segment per stream; the device copy is refreshed unless it already holds the same constants
  bool all_constant = true;
  GainSeg* seg = b->wrap_gains.h;
  for (int s = 0; s < B; ++s) {
    const GainSeg gi = b->gain_in[s].advance(n, w.rate), go = b->gain_out[s].advance(n, w.rate);
    all_constant = all_constant && gi.step == 1.0 && go.step == 1.0 && seg[s].step == 1.0 && seg[B + s].step == 1.0 &&
                   seg[s].amp0 == gi.amp0 && seg[B + s].amp0 == go.amp0;
    seg[s] = gi;
    seg[B + s] = go;
  }
  if (!(all_constant && b->wrap_gains_constant)) {
    const size_t off = 0, len = 2 * (size_t)B;
    GainSeg* dst = nullptr;
    if (!b->wrap_gains.push_parts(st, 1, &off, &len, &dst)) return false;
    b->wrap_gains_constant = all_constant;
  }
  const size_t nt = w.taps_down.size();
  const Dir din = w.to_inner(n);
  const int m = din.n_out;
  if (m < 0 || m > kMaxSamples) return false;
  hipLaunchKernelGGL(wrap_in_kernel, dim3(B), dim3(256), 0, st, d_in, channels, n, b->d_wrap, b->wrap_gains.d, b->d_wrap_taps + (din.decimate ? 0 : nt), din,
                     b->d_wrap_inner, kInnerStride);
  for (int at = 0; at < m;) {  // the exact-480 FIFO; a model hop every time it fills
    const int take = std::min(kBlock - w.fill, m - at);
    const int fires = w.fill + take == kBlock ? 1 : 0;
    hipLaunchKernelGGL(wrap_fifo_kernel, dim3(B), dim3(256), 0, st, b->d_wrap_inner, kInnerStride, b->d_wrap, at, w.fill, take, fires, b->d_in);
    if (fires) {
      if (!step_device(b, nullptr, nullptr)) return false;
      hipLaunchKernelGGL(wrap_refill_kernel, dim3((B * kBlock + 255) / 256), dim3(256), 0, st, b->d_wrap, b->wave.d_out, B);
      w.fill = 0;
    } else {
      w.fill += take;
    }
    at += take;
  }
  const Dir dout = w.to_outer(m);
  if (dout.n_out != n) return false;  // the two clocks are coupled so that a block comes back with its own length
  hipLaunchKernelGGL(wrap_out_kernel, dim3(B), dim3(256), 0, st, b->d_wrap_inner, kInnerStride, b->d_wrap, b->wrap_gains.d + B,
                     b->d_wrap_taps + (dout.decimate ? 0 : nt), dout, d_out, channels);
  return hip_ok(hipGetLastError(), "wrapper launch");
}
static int wrap_max_chunk(const BeatriceBatch* b) {  // host samples per launch so that neither side exceeds the kernels' LDS buffers
  const double r = b->wrap.rate / 48000.0;
  return std::max(1, (int)std::floor((wrapn::kMaxSamples - 8) * std::min(1.0, r)));
}
// in / out: [B][channels][n] planar at the configured host rate; any n >= 1 (long blocks are processed in pieces)
int BeatriceBatch_ProcessBlocksDevice(BeatriceBatch* b, const float* d_in, float* d_out, int channels, int n) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (b->rb.on) return (!d_in && !d_out && channels == b->rb.channels && n == b->rb.n) ? (rb_step(b) ? 0 : -2) : -1;
  if (!b->wrap.ready || channels < 1 || channels > 2 || !d_in || !d_out || n < 1 || b->H != 1 || b->io_slots > 0 || b->pipelined || b->tk.on) return -1;
  const int piece = wrap_max_chunk(b);
  if (n <= piece) return wrap_chunk(b, d_in, d_out, channels, n) ? 0 : -2;
  return -1;  // the planar layout [B][channels][n] cannot be cut without copies: callers pass blocks of at most `piece` samples
}
int BeatriceBatch_MaxWrapperBlock(const BeatriceBatch* b) { return b && b->ok && b->wrap.ready ? wrap_max_chunk(b) : 0; }

// ---- the same wrapper around the TICK pipeline (throughput form, resident blocks) ------------------------------------------------
// One call = one host-rate block per stream from slot `call mod n_slots` of d_in: gains and the first resampling direction, the
// 480-sample accumulation, a model hop into the tick pipeline every time it fills (one tick per hop, at least one tick per
// call so that a hop is out of the pipeline TickStages() - 1 calls after it went in); then the output half of the call made
// `delay` = TickStages() - 1 calls ago, into ITS slot of d_out.  Everything that is control is on the host, as in wrap_chunk.
static void rb_release(BeatriceBatch* b) {
  BeatriceBatch::ResidentBlocks& r = b->rb;
  if (r.d_in16) (void)hipFree(r.d_in16);
  if (r.d_out24) (void)hipFree(r.d_out24);
  if (r.d_gains) (void)hipFree(r.d_gains);
  if (r.h_gains) (void)hipHostFree(r.h_gains);
  if (r.gain_ev) { for (int i = 0; i < r.ring; ++i) if (r.gain_ev[i]) (void)hipEventDestroy(r.gain_ev[i]); delete[] r.gain_ev; }
  r = BeatriceBatch::ResidentBlocks{};
}
static bool rb_step(BeatriceBatch* b) {
  using namespace wrapn;
  BeatriceBatch::ResidentBlocks& r = b->rb;
  const int B = b->B, n = r.n;
  hipStream_t st = b->stream;
  WrapPlan& w = b->wrap;
  const long long call = r.calls;
  const int ge = (int)(call % r.ring);
  // this call's gain segments: input half now, output half when its job runs
  if (call >= r.ring && !hip_ok(hipEventSynchronize(r.gain_ev[ge]), "wrapper gain ring")) return false;
  GainSeg* seg = r.h_gains + (size_t)ge * 2 * B;
  for (int s = 0; s < B; ++s) { seg[s] = b->gain_in[s].advance(n, w.rate); seg[B + s] = b->gain_out[s].advance(n, w.rate); }
  GainSeg* dseg = r.d_gains + (size_t)ge * 2 * B;
  BHIP_TRY(hipMemcpyAsync(dseg, seg, sizeof(GainSeg) * 2 * B, hipMemcpyHostToDevice, st));
  BHIP_TRY(hipEventRecord(r.gain_ev[ge], st));
  const size_t nt = w.taps_down.size();
  const Dir din = w.to_inner(n);
  const int m = din.n_out;
  if (m < 0 || m > kMaxSamples) return false;
  const Dir dout = w.to_outer(m);
  if (dout.n_out != n) return false;
  const float* src = r.d_in + (size_t)(call % r.n_slots) * B * r.channels * n;
  hipLaunchKernelGGL(wrap_in_kernel, dim3(B), dim3(256), 0, st, src, r.channels, n, b->d_wrap, dseg, b->d_wrap_taps + (din.decimate ? 0 : nt), din,
                     b->d_wrap_inner, kInnerStride);
  int ticks = 0;
  for (int at = 0; at < m;) {  // the 480-sample accumulation; a model hop every time it fills (the per-stream FIFO array holds it)
    const int take = std::min(kBlock - w.fill, m - at);
    const int fires = w.fill + take == kBlock ? 1 : 0;
    hipLaunchKernelGGL(wrap_fifo_kernel, dim3(B), dim3(256), 0, st, b->d_wrap_inner, kInnerStride, b->d_wrap, at, w.fill, take, fires,
                       r.d_in16 + (size_t)b->io_host * B * B_IN_HOP);
    if (fires) {
      if (!tick_run(b, true)) return false;
      ++ticks;
      w.fill = 0;
    } else {
      w.fill += take;
    }
    at += take;
  }
  if (ticks == 0 && !tick_run(b, false)) return false;   // the pipeline advances with every call
  r.jobs.push_back(BeatriceBatch::ResidentBlocks::Job{call, r.t48, dout});
  r.t48 += m;
  r.calls = call + 1;
  bool ok = true;
  while (ok && !r.jobs.empty() && r.jobs.front().call + r.delay <= call) {
    ok = rb_post(b, r.jobs.front());
    r.jobs.pop_front();
  }
  b->inflight = true;
  return ok;
}
// d_in / d_out: [n_slots][B][channels][n] planar blocks at the configured host rate (BeatriceBatch_ConfigureWrapper first).
// Call k (BeatriceBatch_ProcessBlocksDevice(b, NULL, NULL, channels, n)) reads slot k mod n_slots; its output block is in the
// same slot of d_out BeatriceBatch_ResidentBlocksDelay() calls later (or after BeatriceBatch_Synchronize).  Same samples as
// the in-order BeatriceBatch_ProcessBlocksDevice.  n_slots > delay + 1.  NULL pointers unbind.
int BeatriceBatch_BindResidentBlocks(BeatriceBatch* b, const float* d_in, float* d_out, int channels, int n, int n_slots) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::ResidentBlocks& r = b->rb;
  if (r.on) {
    if (!sync_all(b)) return -2;
    const int rc = tick_enable(b, false);
    if (rc) return rc;
    (void)BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0);
    rb_release(b);
  }
  if (!d_in && !d_out) return 0;
  const int stages = b->tk.plan.count();
  if (!b->wrap.ready || !d_in || !d_out || channels < 1 || channels > 2 || n < 1 || n > wrap_max_chunk(b) || n_slots < stages + 1 || b->H != 1 ||
      b->io_slots > 0 || b->pipelined || b->tk.on || b->hs.on || b->r48.on)
    return -1;
  if (!sync_all(b)) return -2;
  // model hops a call can fire: ceil(inner samples / 480) + 1; a hop's resident output is read until `delay` calls after the
  // call in which the NEXT hop fired
  const int m_max = (int)std::ceil(n * 48000.0 / b->wrap.rate) + 2, hops_per_call = (m_max + wrapn::kBlock - 1) / wrapn::kBlock + 1;
  r.delay = stages - 1;
  r.ring = r.delay + 3;
  r.io_slots = std::max(stages + 1, (r.delay + 2) * hops_per_call + 2);
  const int B = b->B;
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_in16), sizeof(float) * r.io_slots * B * B_IN_HOP), "rb in16") &&
            hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_out24), sizeof(float) * r.io_slots * B * B_OUT_HOP), "rb out24") &&
            hip_ok(hipMemset(r.d_in16, 0, sizeof(float) * r.io_slots * B * B_IN_HOP), "rb zero") &&
            hip_ok(hipMemset(r.d_out24, 0, sizeof(float) * r.io_slots * B * B_OUT_HOP), "rb zero") &&
            hip_ok(hipMalloc(reinterpret_cast<void**>(&r.d_gains), sizeof(wrapn::GainSeg) * r.ring * 2 * B), "rb gains") &&
            hip_ok(hipHostMalloc(reinterpret_cast<void**>(&r.h_gains), sizeof(wrapn::GainSeg) * r.ring * 2 * B, hipHostMallocDefault), "rb gains host");
  if (ok) {
    r.gain_ev = new hipEvent_t[r.ring]();
    for (int i = 0; i < r.ring && ok; ++i) ok = hip_ok(hipEventCreateWithFlags(&r.gain_ev[i], hipEventDisableTiming), "rb event");
  }
  ok = ok && hip_ok(hipDeviceSynchronize(), "rb sync") && BeatriceBatch_BindResidentIO(b, r.d_in16, r.d_out24, r.io_slots) == 0 && tick_enable(b, true) == 0;
  if (!ok) {
    (void)tick_enable(b, false);
    (void)BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0);
    rb_release(b);
    return -2;
  }
  r.d_in = d_in; r.d_out = d_out; r.channels = channels; r.n = n; r.n_slots = n_slots; r.on = true;
  return 0;
}
int BeatriceBatch_ResidentBlocksDelay(const BeatriceBatch* b) { return b && b->ok && b->rb.on ? b->rb.delay : -1; }
int BeatriceBatch_ProcessBlocks(BeatriceBatch* b, const float* in, float* out, int channels, int n) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (!b->wrap.ready || channels < 1 || channels > 2 || !in || !out || n < 1 || n > wrap_max_chunk(b) || b->H != 1 || b->io_slots > 0 || b->pipelined || b->tk.on) return -1;
  const size_t cnt = (size_t)b->B * channels * n;
  float* h_in = b->h_wrap_io;
  float* h_out = b->h_wrap_io + (size_t)b->B * 2 * wrapn::kMaxSamples;
  float* d_in = b->d_wrap_io;
  float* d_out = b->d_wrap_io + (size_t)b->B * 2 * wrapn::kMaxSamples;
  std::memcpy(h_in, in, sizeof(float) * cnt);
  bool ok = hip_ok(hipMemcpyAsync(d_in, h_in, sizeof(float) * cnt, hipMemcpyHostToDevice, b->stream), "wrap in");
  ok = ok && wrap_chunk(b, d_in, d_out, channels, n);
  ok = ok && hip_ok(hipMemcpyAsync(h_out, d_out, sizeof(float) * cnt, hipMemcpyDeviceToHost, b->stream), "wrap out");
  ok = hip_ok(hipStreamSynchronize(b->stream), "wrap sync") && ok;
  b->inflight = false;
  if (ok) std::memcpy(out, h_out, sizeof(float) * cnt);
  else std::memset(out, 0, sizeof(float) * cnt);
  return ok ? 0 : -2;
}

int BeatriceBatch_Synchronize(BeatriceBatch* b) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  return sync_all(b) ? 0 : -2;
}

int BeatriceBatch_SetStream(BeatriceBatch* b, void* hip_stream) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  (void)sync_all(b);
  drop_graph(b);
  if (b->owns_stream) (void)hipStreamDestroy(b->stream);
  b->stream = static_cast<hipStream_t>(hip_stream);
  b->owns_stream = false;
  return 0;
}
void* BeatriceBatch_GetStream(const BeatriceBatch* b) { return b ? b->stream : nullptr; }
int BeatriceBatch_EnableGraph(BeatriceBatch* b, int enable) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  (void)sync_all(b);
  b->use_graph = enable != 0;
  if (!b->use_graph) drop_graph(b);
  return 0;
}
// ---- host streaming: the tick pipeline with HOST buffers on either side -----------------------------------------------------
// The resident I/O slots of the ticks are the batch's PINNED HOST mirrors: the stages that read a hop (f1, fft, pitch head)
// and the one that writes samples (the tail) go over PCIe themselves -- 160 + 240 KB per tick at 256 streams, spread over
// hundreds of workgroups that have plenty to overlap it with -- so a call is: memcpy the hop into its slot, launch the
// tick, record an event, and hand back the step whose tick finished at least two ticks ago (the host then never waits
// for the device's current work, and two ticks stay queued).  3.07-3.16 M frames/s from and to host memory at 256 streams
// against 3.2-3.55 M with resident device buffers (before / after the last changes of the tick bodies).  BEATRICE_HIP_HS_COPIES=1 (A/B): device slots with an upload and a
// download stream beside the ticks instead -- 2.36 M: copy commands and cross-stream waits cost more than PCIe loads.
static void host_stream_fetch(BeatriceBatch* b) {  // enqueue the download of every step the ticks run so far have completed
  BeatriceBatch::HostStream& h = b->hs;
  const long long last_tick = b->tk.tick - 1;
  const size_t n_out = (size_t)b->B * B_OUT_HOP;
  for (auto& p : h.pending) {
    if (p.fetched || p.done_tick > last_tick) continue;
    if (h.mapped) { p.fetched = true; continue; }  // nothing to download: the last stage wrote host memory
    // (the event recorded behind the tick just launched: it is at or after the tick that completed this step, also when
    //  ticks were run by a drain in between, which records none)
    (void)hipStreamWaitEvent(h.s_out, h.ev_tick[h.rec[0] % h.ev_tick.size()], 0);
    (void)hipMemcpyAsync(h.h_out + p.slot * n_out, h.d_out + p.slot * n_out, sizeof(float) * n_out, hipMemcpyDeviceToHost, h.s_out);
    (void)hipEventRecord(h.ev_out[p.slot], h.s_out);
    p.fetched = true;
  }
}
static bool host_stream_tick(BeatriceBatch* b, bool feeding) {
  BeatriceBatch::HostStream& h = b->hs;
  if (!tick_run(b, feeding)) return false;   // (may run a whole drain first: a stage that comes or goes)
  const long long t = b->tk.tick - 1;        // the tick just launched
  (void)hipEventRecord(h.ev_tick[t % h.ev_tick.size()], b->stream);
  h.tick_of_ev[t % h.ev_tick.size()] = t;
  h.rec[1] = h.rec[0]; h.rec[0] = t;
  host_stream_fetch(b);
  return true;
}
// the samples of pending step f are in the pinned output mirror
static bool host_stream_wait(BeatriceBatch* b, const BeatriceBatch::HostStream::Pending& f) {
  BeatriceBatch::HostStream& h = b->hs;
  if (!h.mapped) return hip_ok(hipEventSynchronize(h.ev_out[f.slot]), "hs download");
  const size_t n = h.ev_tick.size();
  for (long long t = f.done_tick; t <= h.rec[0]; ++t)   // the first event recorded at or behind the tick that completed it
    if (h.tick_of_ev[t % n] == t) return hip_ok(hipEventSynchronize(h.ev_tick[t % n]), "hs tick done");
  return false;
}
}  // extern "C"
namespace {
void host_stream_free(BeatriceBatch* b) {
  BeatriceBatch::HostStream& h = b->hs;
  for (hipEvent_t e : h.ev_in) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h.ev_out) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h.ev_tick) if (e) (void)hipEventDestroy(e);
  h.ev_in.clear(); h.ev_out.clear(); h.ev_tick.clear();
  if (h.s_in) (void)hipStreamDestroy(h.s_in);
  if (h.s_out) (void)hipStreamDestroy(h.s_out);
  if (h.d_in) (void)hipFree(h.d_in);
  if (h.d_out) (void)hipFree(h.d_out);
  if (h.h_in) (void)hipHostFree(h.h_in);
  if (h.h_out) (void)hipHostFree(h.h_out);
  h = BeatriceBatch::HostStream{};
}
}  // namespace
extern "C" {
int BeatriceBatch_EnableHostStreaming(BeatriceBatch* b, int enable) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::HostStream& h = b->hs;
  if ((enable != 0) == h.on) return 0;
  if (!enable) {
    if (!sync_all(b)) return -2;
    (void)hipStreamSynchronize(h.s_in); (void)hipStreamSynchronize(h.s_out);
    const int rc = tick_enable(b, false);
    if (rc) return rc;
    const int rb = BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0);
    host_stream_free(b);
    return rb;
  }
  if (b->H != 1 || b->io_slots > 0 || b->tk.on || b->pipelined) return -1;  // one hop per step; no other binding or pipelining
  h.n_slots = b->tk.plan.count() + 8;
  const size_t n_in = (size_t)b->B * B_IN_HOP, n_out = (size_t)b->B * B_OUT_HOP;
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void**>(&h.d_in), sizeof(float) * n_in * h.n_slots), "hs d_in") &&
            hip_ok(hipMalloc(reinterpret_cast<void**>(&h.d_out), sizeof(float) * n_out * h.n_slots), "hs d_out") &&
            hip_ok(hipHostMalloc(reinterpret_cast<void**>(&h.h_in), sizeof(float) * n_in * h.n_slots, hipHostMallocDefault), "hs h_in") &&
            hip_ok(hipHostMalloc(reinterpret_cast<void**>(&h.h_out), sizeof(float) * n_out * h.n_slots, hipHostMallocDefault), "hs h_out") &&
            hip_ok(hipMemset(h.d_in, 0, sizeof(float) * n_in * h.n_slots), "hs zero") &&
            hip_ok(hipStreamCreateWithFlags(&h.s_in, hipStreamNonBlocking), "hs s_in") &&
            hip_ok(hipStreamCreateWithFlags(&h.s_out, hipStreamNonBlocking), "hs s_out");
  h.ev_in.assign(h.n_slots, nullptr); h.ev_out.assign(h.n_slots, nullptr); h.ev_tick.assign(tick::kRing, nullptr);
  for (auto* v : {&h.ev_in, &h.ev_out, &h.ev_tick})
    for (hipEvent_t& e : *v) ok = ok && hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hs event");
  h.tick_of_ev.assign(tick::kRing, -1);
  h.mapped = std::getenv("BEATRICE_HIP_HS_COPIES") == nullptr;   // A/B switch: copies on two more streams instead
  if (ok && h.mapped) std::memset(h.h_in, 0, sizeof(float) * n_in * h.n_slots);
  ok = ok && BeatriceBatch_BindResidentIO(b, h.mapped ? h.h_in : h.d_in, h.mapped ? h.h_out : h.d_out, h.n_slots) == 0 && tick_enable(b, true) == 0;
  if (!ok) { (void)tick_enable(b, false); (void)BeatriceBatch_BindResidentIO(b, nullptr, nullptr, 0); host_stream_free(b); return -2; }
  h.pending.clear();
  h.fed = 0;
  h.rec[0] = h.rec[1] = -1;
  h.on = true;
  return 0;
}
int BeatriceBatch_HostStreamDelay(const BeatriceBatch* b) { return b ? b->tk.plan.count() + 1 : 0; }
// in: [B][160] host; out: [B][240] host.  Returns 1 when `out` received the samples of the step fed
// BeatriceBatch_HostStreamDelay() calls ago, 0 while the pipeline is still filling (out untouched), < 0 on error.
int BeatriceBatch_StreamFrames(BeatriceBatch* b, const float* in, float* out) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::HostStream& h = b->hs;
  if (!h.on || !in || !out) return -1;
  const size_t n_in = (size_t)b->B * B_IN_HOP, n_out = (size_t)b->B * B_OUT_HOP;
  const int slot = b->io_host;  // the slot the tick about to be fed reads and, pipeline depth later, writes
  if (h.mapped) {
    // the slot's last readers (stage 9 of the step fed n_slots calls ago) are done: every call since the pipeline filled
    // has waited for a tick later than theirs before handing back its output
    std::memcpy(h.h_in + slot * n_in, in, sizeof(float) * n_in);
    if (!host_stream_tick(b, true)) return -2;
    h.pending.push_back({h.fed, slot, b->tk.last_feed_tick + b->tk.plan.count() - 1, false});
    h.fed += 1;
    const BeatriceBatch::HostStream::Pending& f = h.pending.front();
    if (!f.fetched || f.done_tick > b->tk.last_feed_tick - 2) return 0;   // keep two ticks queued on the device while the host waits
    if (!host_stream_wait(b, f)) return -2;
    std::memcpy(out, h.h_out + f.slot * n_out, sizeof(float) * n_out);
    h.pending.pop_front();
    return 1;
  }
  if (!hip_ok(hipEventSynchronize(h.ev_in[slot]), "hs in reuse")) return -2;  // the upload that last used this pinned slot (long done)
  std::memcpy(h.h_in + slot * n_in, in, sizeof(float) * n_in);
  // every reader of the device slot's old contents is done once the tick before the previous one is (the slot ring is
  // longer than the deepest reader's stage by more than that)
  if (h.rec[1] >= 0) (void)hipStreamWaitEvent(h.s_in, h.ev_tick[h.rec[1] % h.ev_tick.size()], 0);
  bool ok = hip_ok(hipMemcpyAsync(h.d_in + slot * n_in, h.h_in + slot * n_in, sizeof(float) * n_in, hipMemcpyHostToDevice, h.s_in), "hs upload");
  (void)hipEventRecord(h.ev_in[slot], h.s_in);
  (void)hipStreamWaitEvent(b->stream, h.ev_in[slot], 0);
  (void)hipStreamWaitEvent(b->stream, h.ev_out[slot], 0);  // the output slot this step will overwrite has been downloaded
  ok = ok && host_stream_tick(b, true);
  if (!ok) return -2;
  h.pending.push_back({h.fed, slot, b->tk.last_feed_tick + b->tk.plan.count() - 1, false});  // leaves the last stage that many ticks on
  h.fed += 1;
  const BeatriceBatch::HostStream::Pending& f = h.pending.front();
  if (!f.fetched || f.done_tick > b->tk.last_feed_tick - 2) return 0;   // hand back only what was enqueued for download two ticks ago
  if (!hip_ok(hipEventSynchronize(h.ev_out[f.slot]), "hs download")) return -2;
  std::memcpy(out, h.h_out + f.slot * n_out, sizeof(float) * n_out);
  h.pending.pop_front();
  return 1;
}
// After the last StreamFrames: hands back the next step still inside the pipeline (running ticks without input as
// needed); returns 1 with `out` filled, 0 when nothing is pending.
int BeatriceBatch_StreamFlush(BeatriceBatch* b, float* out) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  BeatriceBatch::HostStream& h = b->hs;
  if (!h.on || !out) return -1;
  if (h.pending.empty()) return 0;
  const size_t n_out = (size_t)b->B * B_OUT_HOP;
  while (!h.pending.front().fetched)
    if (!host_stream_tick(b, false)) return -2;
  const BeatriceBatch::HostStream::Pending f = h.pending.front();
  if (!host_stream_wait(b, f)) return -2;
  std::memcpy(out, h.h_out + f.slot * n_out, sizeof(float) * n_out);
  h.pending.pop_front();
  return 1;
}

// Throughput mode for callers that enqueue steps ahead (BeatriceBatch_ConvertFramesDevice without waiting,
// resident I/O): the front end of step t+1 runs on the batch's stream while the waveform generator of step t
// runs on a second stream.  Same results; a step's output is complete when BeatriceBatch_Synchronize returns
// (or, stream-ordered, on BeatriceBatch_GetWaveStream).  Off by default: everything in order on one stream.
int BeatriceBatch_EnableTickPipeline(BeatriceBatch* b, int enable) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (b->hs.on || b->r48.on) return -1;  // those modes own the tick pipeline: leave them instead
  return tick_enable(b, enable != 0);
}
int BeatriceBatch_TickStages(const BeatriceBatch* b) { return b ? b->tk.plan.count() : 0; }
// Measurement hook: `ticks` more ticks (each feeding a step from the resident slots), every tick's pipeline launch
// between one pair of HIP events on the batch's stream; returns the mean duration per launch and its algorithmic work.
// Call with the pipeline full (at least BeatriceBatch_TickStages steps fed) for the steady-state figure.
int BeatriceBatch_TimeTickLaunch(BeatriceBatch* b, int ticks, float* us_per_launch, double* flops, double* bytes) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (!b->tk.on || ticks < 1 || ticks > 64 || !us_per_launch) return -1;
  // ONE pair of events around `ticks` back-to-back launches (an event pair per launch adds two commands between
  // consecutive launches and reads ~5 us long against rocprofv3's kernel durations); the figure includes the boundary
  // between two ticks, which belongs to the launch's cost
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool ok = hip_ok(hipEventCreate(&ev[0]), "tick ev") && hip_ok(hipEventCreate(&ev[1]), "tick ev");
  ok = ok && hip_ok(hipEventRecord(ev[0], b->stream), "tick ev0");
  for (int i = 0; i < ticks && ok; ++i) ok = tick_run(b, true);
  ok = ok && hip_ok(hipEventRecord(ev[1], b->stream), "tick ev1") && hip_ok(hipStreamSynchronize(b->stream), "tick time sync");
  float ms = 0;
  ok = ok && hip_ok(hipEventElapsedTime(&ms, ev[0], ev[1]), "tick elapsed");
  for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
  if (!ok) return -2;
  *us_per_launch = (float)(1000.0 * ms / ticks);
  if (flops) *flops = b->tk.table_flops;
  if (bytes) *bytes = b->tk.table_bytes;
  return 0;
}
int BeatriceBatch_EnablePipelining(BeatriceBatch* b, int enable) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (b->tk.on) return -1;
  if (!set_io_mapped(b, false) || !sync_all(b)) return -2;
  if (enable < 0 || enable > BeatriceBatch::kMaxStages) return -1;
  drop_graph(b);  // stages are captured on the streams they will run on
  set_plan(b, enable == 1 ? 2 : enable);  // 1 = the default depth
  for (int s = 1; s < b->n_stages && b->pipelined; ++s)  // every stream takes a hardware queue: only those in use exist
    if (!b->stage_stream_own[s] && !make_stage_stream(&b->stage_stream_own[s], s)) return -2;
  return 0;
}
void* BeatriceBatch_GetWaveStream(const BeatriceBatch* b) { return b ? wave_stream(b) : nullptr; }
// Captures the hipGraphs of the current mode now (nothing is executed), so that the first steps do not pay for it.
int BeatriceBatch_Prepare(BeatriceBatch* b) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (b->vq_dirty) { update_vq_mode(b); b->vq_dirty = false; }
  if (!b->use_graph) return 0;
  bool ok = true;
  if (!b->pipelined) ok = run_step_in_order(b, -1);
  else for (int s = 0; s < b->n_stages && ok; ++s) ok = run_stage(b, s, -1);
  return ok ? 0 : -2;
}
float* BeatriceBatch_DeviceInput(BeatriceBatch* b) { return b && b->ok ? b->d_in : nullptr; }
float* BeatriceBatch_DeviceOutput(BeatriceBatch* b) {
  const DeviceScope dev_(b ? b->device : -1);
  return b && b->ok && set_io_mapped(b, false) ? b->wave.d_out : nullptr;
}

int BeatriceBatch_GetIntermediates(BeatriceBatch* b, float* phone, int* q_raw, int* q, float* feat) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  bool ok = sync_all(b);
  if (phone) {  // the phone vectors of the last step sit in one of the three step slots of a per-stream ring
    const size_t row = sizeof(float) * b->H * B_PHONE_CH;
    ok = ok && hip_ok(hipMemcpy2D(phone, row, b->phone.d_phone + (size_t)b->last_parity * b->H * B_PHONE_CH, row * b->phone.out_slots, row,
                                  b->B, hipMemcpyDeviceToHost), "phone");
  }
  const size_t qo = (size_t)(b->last_hop % b->pitch.q_slots) * b->B * b->H;  // the pitch head's outputs are double-buffered by step
  if (q_raw) ok = ok && hip_ok(hipMemcpy(q_raw, b->pitch.d_q_raw + qo, sizeof(int) * b->B * b->H, hipMemcpyDeviceToHost), "q_raw");
  if (q) ok = ok && hip_ok(hipMemcpy(q, b->pitch.d_q + qo, sizeof(int) * b->B * b->H, hipMemcpyDeviceToHost), "q");
  if (feat) ok = ok && hip_ok(hipMemcpy(feat, b->pitch.d_feat + qo * 4, sizeof(float) * b->B * b->H * 4, hipMemcpyDeviceToHost), "feat");
  return ok ? 0 : -2;
}

namespace {
struct KernelRow { std::string name; int launches = 0; double us = 0, flops = 0, bytes = 0; };
struct ProfileHook : LaunchHook {
  std::vector<KernelRow> rows;
  int repeats = 1;
  hipEvent_t e0, e1;
  bool ok = true;
  void on_launch(const LaunchInfo& info, hipStream_t stream, void (*thunk)(void*), void* ctx) override {
    // launches that update state in place run once (the tail rewrites its history block, the pitch head its previous
    // bin, the GRUs their state, hop_advance the counter); everything else only writes this step's ring slots
    const bool once = std::strstr(info.name, "hop_advance") || std::strstr(info.name, "wave.tail") || std::strstr(info.name, "pitch.head") ||
                      std::strstr(info.name, "gru");
    const int reps = once ? 1 : repeats;
    ok = ok && hip_ok(hipEventRecord(e0, stream), "p0");
    for (int i = 0; i < reps; ++i) thunk(ctx);
    float ms = 0.f;
    ok = ok && hip_ok(hipEventRecord(e1, stream), "p1") && hip_ok(hipEventSynchronize(e1), "ps") &&
         hip_ok(hipEventElapsedTime(&ms, e0, e1), "pe");
    KernelRow* row = nullptr;
    for (auto& r : rows) if (r.name == info.name) row = &r;
    if (!row) { rows.push_back(KernelRow{info.name}); row = &rows.back(); }
    row->launches += 1;
    row->us += 1000.0 * ms / reps;
    row->flops = info.flops;
    row->bytes = info.bytes;
  }
};
}  // namespace

int BeatriceBatch_ProfileKernels(BeatriceBatch* b, int repeats, int max_entries, char* names, int* launches, double* mean_us,
                                 double* flops, double* bytes) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (repeats < 1 || max_entries < 1 || !names || !launches || !mean_us || !flops || !bytes || b->tk.on) return -1;
  if (!sync_all(b)) return -2;
  advance_kv(b);
  if (b->vq_dirty) { update_vq_mode(b); b->vq_dirty = false; }
  const int slot = b->hop_host & 3;
  if (!push_settings(b, slot)) return -2;
  ProfileHook hook;
  hook.repeats = repeats;
  hook.e0 = b->ev0;
  hook.e1 = b->ev1;
  launch_hook() = &hook;
  enqueue_front(b, b->stream);  // one step, eagerly, every stage on the batch's stream
  for (int st = 1; st < b->n_stages; ++st) enqueue_wave(b, st, slot, b->stream);
  launch_hook() = nullptr;
  b->last_parity = b->hop_host % 3;
  b->last_hop = b->hop_host;
  b->hop_host = hop_next(b->hop_host);
  if (b->io_slots > 0) b->io_host = (b->io_host + 1) % b->io_slots;
  b->steps_enqueued += 1;
  if (!hook.ok || !sync_all(b)) return -2;
  const int n = std::min<int>((int)hook.rows.size(), max_entries);
  for (int i = 0; i < n; ++i) {
    std::memset(names + 64 * i, 0, 64);
    std::strncpy(names + 64 * i, hook.rows[i].name.c_str(), 63);
    launches[i] = hook.rows[i].launches;
    mean_us[i] = hook.rows[i].us / hook.rows[i].launches;
    flops[i] = hook.rows[i].flops;
    bytes[i] = hook.rows[i].bytes;
  }
  return n;
}

int BeatriceBatch_TimeSteps(BeatriceBatch* b, int steps, float* ms) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok || steps < 1 || !ms) return b && b->ok ? -1 : -2;
  bool ok = sync_all(b) && hip_ok(hipEventRecord(b->ev0, b->stream), "ev0");
  for (int i = 0; i < steps && ok; ++i) ok = step_device(b, nullptr, nullptr);
  ok = ok && hip_ok(hipEventRecord(b->ev1, wave_stream(b)), "ev1") && hip_ok(hipEventSynchronize(b->ev1), "evsync") &&
       hip_ok(hipEventElapsedTime(ms, b->ev0, b->ev1), "elapsed");
  return (sync_all(b) && ok) ? 0 : -2;
}

}  // extern "C"
