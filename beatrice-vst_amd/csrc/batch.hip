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
    long long deferred_step = -1;                 // ... and which step that was (its per-stream counters name its silent streams)
  } r48;
  float *d_coef_down = nullptr, *d_coef_up = nullptr, *d_io48 = nullptr, *h_io48 = nullptr;  // io: in [B][2][480] | out [B][2][480]
  // The shell's silent-block rule per stream (BeatriceBatch_EnableSilentBlockRule; kernels_misc.hip.h freeze_*): streams
  // flagged for the next 48 kHz block stand still -- model state, wrapper state, key/value installs, codebook lottery
  struct SilentRule {
    bool on = false;
    std::vector<unsigned char> next;    // [B] flags of the step to come (cleared by the step)
    bool any_next = false;
    bool in_block_step = false;         // the step being enqueued belongs to a 48 kHz block (only those honour `next`)
    static constexpr int kDepth = 4;
    unsigned char *d_flags = nullptr, *h_flags = nullptr;   // [kDepth][B]: a step's flags on the device / their pinned staging
    hipEvent_t flag_ev[kDepth] = {};                        // slot's upload AND the step that reads the device copy are done
    bool flag_pending[kDepth] = {};
    FreezeRing* d_rings = nullptr;
    int n_rings = 0;
    float* d_keep = nullptr;            // copies of the single-slot rings, all streams
    int* d_keep_prev_q = nullptr;       // [B]
    long long steps = 0;
  } silent;
  // The any-rate wrapper with clocks PER STREAM (BeatriceBatch_ConfigureWrapperRates / ProcessBlocksRagged; wrapper.hip.h
  // RagStream): every stream its own host rate, block length and FIFO phase; a stream may sit a call out
  struct RaggedWrap {
    bool ready = false;
    std::vector<wrapn::WrapPlan> classes;          // one per distinct host rate: ratio and tap tables
    std::vector<int> cls;                          // [B] class of each stream
    struct Clock { int phase_down, phase_up, fill; };
    std::vector<Clock> clk;                        // [B] the stream's two resampler clocks and its FIFO fill
    std::vector<Clock> clk_undo;                   // what ProcessBlocksRagged restores when a call is refused half-way through its plan
    std::vector<wrapn::GainClock> gain_undo_in, gain_undo_out;
    std::vector<int> taps_down_off, taps_up_off;   // per class: float offsets into d_taps
    float* d_taps = nullptr;
    static constexpr int kStage = 4;
    wrapn::RagStream *d_rs = nullptr, *h_rs = nullptr;   // [kStage][B]: a call's per-stream records, pinned staging and device copy
    hipEvent_t ev[kStage] = {};
    bool pending[kStage] = {};
    long long calls = 0;
    unsigned char* d_frozen = nullptr;             // [kMaxChunks][B]: per FIFO chunk, the streams that do not fire a hop in it
  } rw;
  // The any-rate wrapper around the tick pipeline (BeatriceBatch_BindResidentBlocks): host-rate blocks resident on the device,
  // the input half of the chain in front of the ticks, the output half `delay` calls later (wrapper.hip.h wrap_post_kernel)
  struct ResidentBlocks {
    bool on = false;
    bool dead = false;   // a call of the uniform form failed after the host's clocks had advanced (rb_step): host and device wrapper state no
                         // longer agree, every further call is refused with -2 until the binding is released
    int channels = 0, n = 0, n_slots = 0, io_slots = 0, delay = 0, ring = 0;
    const float* d_in = nullptr;   // [n_slots][B][channels][n]
    float* d_out = nullptr;        // [n_slots][B][channels][n]
    float *d_in16 = nullptr, *d_out24 = nullptr;   // [io_slots][B][H][160], [io_slots][B][H][240]: the resident I/O of the ticks
    wrapn::GainSeg* h_gains = nullptr;                       // [ring][2][B] pinned: a call's input | output segments, read by the kernels in place
    hipEvent_t* gain_ev = nullptr;                           // [ring]: the output half that read ring entry i has run
    long long calls = 0, t48 = 0;                            // calls so far; 48 kHz samples fed so far
    long long hops_fired = 0;                                // model hops the FIFO has fired; hop k = hop k % H of step k / H
    long long hops_done = 0;                                 // hops of the steps fed by the end of the previous call
    std::vector<char> ev_recorded;                           // [ring] gain_ev[i] marks the output half that read ring entry i last
    int H = 1;                                               // hops per step of the batch
    long long hops_fed() const { return hops_fired / H * H; }   // ... of which the hops of full steps are inside (or through) the ticks
    struct Job {
      long long call, t0; wrapn::Dir dout;
      long long last_hop() const { return (t0 + dout.n_in - 1) / wrapn::kBlock - 1; }   // the newest model hop its samples come from (wrap_post_kernel)
    };
    std::deque<Job> jobs;                                    // calls whose output half is still to run, oldest first
    float* d_zero = nullptr;                                 // [B][channels][n] zeros: the input of the calls BeatriceBatch_FlushResidentBlocks makes up
    // the form with clocks PER STREAM (BeatriceBatch_BindResidentBlocksRagged; the rates, clocks and tap tables are `rw`'s): a slot
    // holds one cell of channels x max_samples floats per stream; a call's per-stream records stay on the device until its output
    // half has run; slot_map[b][g mod map_ring] = the resident slot of the step that stream b's hop g rode in
    bool ragged = false;
    int max_samples = 0, cell = 0, map_ring = 0;
    wrapn::RagStream* h_rs = nullptr;                        // [ring][B] pinned, read by the kernels in place
    int* d_map = nullptr;                                    // [B][map_ring]
    std::vector<long long> t48_s;                            // [B] 48 kHz samples of the stream fed so far
    std::vector<int> hops_s;                                 // [B] model hops the stream has fired
  } rb;
};

namespace {

// Waits for the steps enqueued so far (needed before the graph or a speaker table they use is replaced;
// the pinned setting mirrors are double-buffered and do not need it).
hipStream_t stage_stream(const BeatriceBatch* b, int s) { return b->pipelined && s > 0 ? b->stage_stream_own[s] : b->stream; }
hipStream_t wave_stream(const BeatriceBatch* b) { return stage_stream(b, b->n_stages - 1); }  // where a step's output appears
bool tick_drain(BeatriceBatch* b);
void host_stream_free(BeatriceBatch* b);
void drop_graph(BeatriceBatch* b);
bool sync_all(BeatriceBatch* b) {
  bool ok = !b->tk.on || tick_drain(b);  // steps still inside the tick pipeline come out first
  ok = hip_ok(hipStreamSynchronize(b->stream), "sync") && ok;
  for (int s = 1; s < BeatriceBatch::kMaxStages; ++s)
    if (b->stage_stream_own[s]) ok = hip_ok(hipStreamSynchronize(b->stage_stream_own[s]), "sync stage") && ok;
  b->inflight = false;
  // a one-stream batch runs the modules' team launches (team.hip.h): a timeout voids the steps since the last synchronisation; the
  // stream restarts from silence on the per-layer launches (engine.h team_recover)
  if (team_timed_out(b->phone) || team_timed_out(b->pitch) || team_timed_out(b->wave)) {
    std::fprintf(stderr, "beatrice_hip: a team launch timed out; the batch's stream was reset and continues on the per-layer launches\n");
    // (each module recovers for itself: the one whose team gave a wait up restarts from silence, the others' state is valid and stays)
    const bool pitch_too = team_timed_out(b->pitch);
    team_recover(b->phone, b->stream); team_recover(b->pitch, b->stream); team_recover(b->wave, b->stream);
    if (pitch_too) (void)hipMemsetAsync(b->pitch.d_prev_q, 0, sizeof(int) * b->B, b->stream);   // (the one piece of the pitch estimator's state outside its rings)
    drop_graph(b);
    ok = false;
  }
  if (b->tk.h_link_dead && *b->tk.h_link_dead) {   // a GRU cell of a tick gave up waiting for the cell of the hop before (tick.hip.h): results are void
    std::fprintf(stderr, "beatrice_hip: tick launch: a linked GRU cell timed out\n");
    b->ok = false;
    ok = false;
  }
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
  static const bool no_quads = bhip::meas_env("BEATRICE_HIP_TICK_NO_QUADS") != nullptr;
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
  bool advanced = false, mixed_left = false;
  const int H = b->H;
  for (int s = 0; s < b->B; ++s) {
    StreamCfg& c = b->cfg[s];
    if (b->silent.any_next && b->silent.next[s]) {   // a silent block: the reference does not reach its per-hop protocol
      // (several hops per step: a switch that completed INSIDE the stream's last step left its rows on mixed key/value entries -- hop 0 had
      //  one new block, hop 1 two ... --; they are brought to the settled entries in the stream's next step, so the pass must come back for it)
      for (int hh = 0; hh < H && H > 1 && !mixed_left; ++hh)
        for (int blk = 0; blk < B_NBLOCKS; ++blk) mixed_left = mixed_left || b->row_slot[blk][(size_t)s * H + hh] != c.kv_slot[blk];
      continue;
    }
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
  b->kv_transient = (advanced && H > 1) || mixed_left;
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
  static const bool no_pairs = bhip::meas_env("BEATRICE_HIP_NO_PAIRS") != nullptr;  // A/B switch for measurements
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
    if (b->silent.any_next && b->silent.next[s]) continue;   // no hop, no draw
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
bool project_speakers(BeatriceBatch* b, int first, int count);
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
  if (b->silent.any_next && !b->silent.in_block_step) {   // flags name streams of the next 48 kHz BLOCK: any other kind
    std::fill(b->silent.next.begin(), b->silent.next.end(), 0);           // of step runs for every stream
    b->silent.any_next = false;
  }
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

#include "batch_tick.hip.h"      // tick mode: table, per-tick host work, drain, enable

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
static bool rb_step(BeatriceBatch* b, bool synthetic = false);
static void rb_release(BeatriceBatch* b);
static void silent_release(BeatriceBatch* b);

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
  const bool slack = H <= tick::kMaxHops;  // rings sized so that every layer can be its own pipeline stage (tick.hip.h)
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
  if (b->tk.d_hopv) (void)hipFree(b->tk.d_hopv);
  if (b->tk.h_hopv) (void)hipHostFree(b->tk.h_hopv);
  for (hipEvent_t e : b->tk.hv_ev) if (e) (void)hipEventDestroy(e);
  { void* wr[] = {b->d_wrap, b->d_wrap_taps, b->d_wrap_inner, b->d_wrap_io, b->wrap_gains.d, b->rw.d_taps, b->rw.d_rs, b->rw.d_frozen}; for (void* p : wr) if (p) (void)hipFree(p); }
  if (b->rw.h_rs) (void)hipHostFree(b->rw.h_rs);
  for (hipEvent_t e : b->rw.ev) if (e) (void)hipEventDestroy(e);
  if (b->h_wrap_io) (void)hipHostFree(b->h_wrap_io);
  b->wrap_gains.release();
  { void* tk[] = {b->tk.d_table, b->tk.d_table_sparse, b->tk.d_snap, b->tk.d_trace, b->tk.d_link_q, b->tk.d_link_p, b->tk.d_desc, b->tk.d_desc_sparse, b->tk.d_desc_plain, b->tk.d_desc_ranges, b->tk.d_ring_table, b->tk.d_shift}; for (void* p : tk) if (p) (void)hipFree(p); }
  if (b->tk.h_link_dead) (void)hipHostFree(b->tk.h_link_dead);
  if (b->tk.h_stage) (void)hipHostFree(b->tk.h_stage);
  for (hipEvent_t e : b->tk.stage_ev) if (e) (void)hipEventDestroy(e);
  host_stream_free(b);
  if (b->r48.d_in16) (void)hipFree(b->r48.d_in16);
  if (b->r48.d_out24) (void)hipFree(b->r48.d_out24);
  rb_release(b);
  silent_release(b);
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
namespace {
bool project_speakers(BeatriceBatch* b, int first, int count) {
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
}  // namespace

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
    for (const StreamCfg& c : b->cfg) {
      for (int blk = 0; blk < B_NBLOCKS; ++blk) if (c.kv_slot[blk] == slot) return -3;
      // ... or is on its way there: installs still pending from an earlier staged call, or a stream that took `slot` by
      // another route (its additive row and key/value tables would be overwritten under it)
      if (c.target_speaker == slot || c.additive_speaker == slot) return -3;
    }
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
// The same for n (stream, speaker) pairs in one call -- a server that moves many streams before a step (BASELINE.json configs[3]:
// 64 rotating speakers) pays the bookkeeping of the pending key/value installs once, not once per stream.  All or nothing: an
// invalid pair changes nothing.
int BeatriceBatch_SetTargetSpeakers(BeatriceBatch* b, int n, const int* streams, const int* speakers) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (n < 0 || (n > 0 && (!streams || !speakers))) return -1;
  for (int i = 0; i < n; ++i)
    if (streams[i] < 0 || streams[i] >= b->B || speakers[i] < 0 || speakers[i] >= b->max_speakers) return -1;
  for (int i = 0; i < n; ++i) {
    StreamCfg& c = b->cfg[streams[i]];
    c.target_speaker = speakers[i]; c.codebook_speaker = speakers[i]; c.additive_speaker = speakers[i]; c.kv_set_count = 0; c.kv_delay = 0;
    for (int& cr : c.codebook_row) cr = speakers[i];
    sync_stream_arrays(b, streams[i]);
  }
  b->pending_kv = 0;
  for (const StreamCfg& c : b->cfg) if (c.kv_set_count < B_NBLOCKS) ++b->pending_kv;
  return 0;
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
  if (b->io_slots > 0 && (d_in || d_out)) return -1;   // resident I/O is bound: NULL, NULL (the step reads and writes its slot)
  if (b->hs.on || b->r48.on || b->rb.on) return -1;     // host streaming and the wrappers around the ticks feed the pipeline through their own entry points
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
  if (bind && b->silent.on) return -1;  // the silent-block rule is an in-order mode of the 48 kHz blocks: switch it off first
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
#include "batch_wrappers.hip.h"  // 48 kHz and any-rate wrappers, in order and around the ticks

// Test hook (beatrice_batch.h): the wave module's team launch of a one-stream batch "times out" at the next synchronisation (sync_all's recovery path)
int BeatriceBatch_InjectTeamTimeout(BeatriceBatch* b) {
  if (!b || !b->ok || !b->wave.d_team_dead || b->wave.team_off) return -1;
  *b->wave.d_team_dead = 1;
  return 0;
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
#include "batch_host_stream.hip.h"  // BeatriceBatch_StreamFrames

// Throughput mode for callers that enqueue steps ahead (BeatriceBatch_ConvertFramesDevice without waiting,
// resident I/O): the front end of step t+1 runs on the batch's stream while the waveform generator of step t
// runs on a second stream.  Same results; a step's output is complete when BeatriceBatch_Synchronize returns
// (or, stream-ordered, on BeatriceBatch_GetWaveStream).  Off by default: everything in order on one stream.
int BeatriceBatch_EnableTickPipeline(BeatriceBatch* b, int enable) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (b->hs.on || b->r48.on || b->rb.on) return -1;  // those modes own the tick pipeline: leave them instead
  if (enable && b->silent.on) return -1;              // (see BeatriceBatch_EnableSilentBlockRule)
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
  if (b->hs.on || b->r48.on || b->rb.on) return -1;   // plain tick mode only (those modes feed the pipeline through their own entry points)
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
#ifdef BEATRICE_HIP_MEASUREMENT_BUILD
// MEASUREMENT BUILDS ONLY: from the next tick on the launch holds only the bodies whose tick::BodyType bit is set in `mask`
// (the other stages' outputs go stale: instruction counters and timings per body type, never results)
extern "C" int BeatriceBatchMeas_TickOnlyTypes(BeatriceBatch* b, unsigned long long mask) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok || !b->tk.on) return -1;
  if (!hip_ok(hipStreamSynchronize(b->stream), "meas sync")) return -2;
  g_tick_only_types = mask;
  b->tk.table_dirty = true;
  return 0;
}
#endif
int BeatriceBatch_EnablePipelining(BeatriceBatch* b, int enable) {
  const DeviceScope dev_(b ? b->device : -1);
  if (!b || !b->ok) return -2;
  if (b->tk.on) return -1;
  if (!set_io_mapped(b, false) || !sync_all(b)) return -2;
  if (enable < 0 || enable > BeatriceBatch::kMaxStages) return -1;
  if (enable >= 1 && b->silent.on) return -1;  // (see BeatriceBatch_EnableSilentBlockRule)
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
  if (b->hs.on || b->r48.on || b->rb.on) return -1;   // (as BeatriceBatch_ConvertFramesDevice: those modes feed the pipeline through their own entry points)
  bool ok = sync_all(b) && hip_ok(hipEventRecord(b->ev0, b->stream), "ev0");
  for (int i = 0; i < steps && ok; ++i) ok = step_device(b, nullptr, nullptr);
  ok = ok && hip_ok(hipEventRecord(b->ev1, wave_stream(b)), "ev1") && hip_ok(hipEventSynchronize(b->ev1), "evsync") &&
       hip_ok(hipEventElapsedTime(ms, b->ev0, b->ev1), "elapsed");
  return (sync_all(b) && ok) ? 0 : -2;
}

}  // extern "C"
