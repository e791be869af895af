// abi_objects.h -- definitions of the opaque C-ABI objects (include/beatrice_abi.h) for the HIP
// implementation.  Shared by abi.hip (1-stream ABI) and batch.hip (batched extension).
#pragma once
#include <cstdint>
#include <mutex>
#include <vector>

#include "beatrice_abi.h"
#include "engine.h"

namespace bhip {
enum : uint32_t { KIND_PHONE = 1, KIND_PITCH = 2, KIND_WAVE = 3, KIND_EMBED = 4, KIND_SPEAKERS = 5 };
Beatrice_ErrorCode parse_model_bytes(const unsigned char* bytes, size_t size, uint32_t kind, long expect_floats,
                                     std::vector<float>* out);
Beatrice_ErrorCode read_model_file(const char* path, uint32_t kind, long expect_floats, std::vector<float>* out);
bool make_stream(hipStream_t* s);

// ---- devices.  One process may drive several GPUs (a C++ host with one thread per GPU, examples/node_convert.cc; the
// reference runs many plugin instances per process, src/vst/factory.cc:21): every object remembers the device it was created
// on -- the calling thread's target, BeatriceHip_SetDevice, else its current HIP device -- and every entry point makes that
// device current for its own duration (DeviceScope), whatever the calling thread had selected.
int target_device();              // device for objects this thread creates next
void set_target_device(int d);    // -1: follow the thread's current HIP device again
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  explicit DeviceScope(int device) {
    int cur = -1;
    if (device >= 0 && hipGetDevice(&cur) == hipSuccess && cur != device) { prev = cur; switched = hipSetDevice(device) == hipSuccess; }
  }
  ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};
// One slot of a phone context's codebook pool (abi.hip, Beatrice20rc0_SetCodebook): the device form (transposed +
// norms) of the caller's table at `host`; `print` fingerprints the caller's bytes so that a table rewritten in
// place (a host that reloads a model into the same storage) is re-uploaded instead of served stale.
struct CodebookEntry { const float* host = nullptr; uint64_t print = 0; unsigned long long last_use = 0; float* d_cbT = nullptr; float* d_cnorm = nullptr; };
constexpr int kCodebookPool = 12;  // >= the 8 speakers a morph can draw from (reference processor_core_2.cc:515-525) + slack
}  // namespace bhip

// model objects: immutable after Read*Parameters, shareable between contexts and threads
struct Beatrice20rc0_PhoneExtractor { int device = bhip::target_device(); bhip::DeviceBlob blob; bhip::PhoneWeights w{}; bool loaded = false; };
struct Beatrice20rc0_PitchEstimator { int device = bhip::target_device(); bhip::DeviceBlob blob; bhip::PitchWeights w{}; bool loaded = false;
                                      unsigned generation = 0; /* counts Read*Parameters: a pre-executed hop (abi.hip) is only claimed by the parameters it ran with */ };
struct Beatrice20rc0_WaveformGenerator { int device = bhip::target_device(); bhip::DeviceBlob blob; bhip::WaveWeights w{}; bool loaded = false; };
struct Beatrice20rc0_EmbeddingSetter { int device = bhip::target_device(); bhip::DeviceBlob blob; bhip::EmbedWeights w{}; bool loaded = false; };

// One per-hop call = one hipGraph launch (input copy, the module's kernels, output copy), captured at the first hop of a
// context with a given model (and again when the set of launches changes: k-NN on / off).  ~1-2 ms once, on the thread
// that makes the first call; afterwards a call costs one graph launch of host time instead of 7-30 kernel launches.
namespace bhip {
struct HopGraph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  const void* blob = nullptr;  // the device parameter blob the captured kernels read (a reload re-allocates it)
  int variant = -1;
  bool eager = false;          // capture or instantiation failed for (blob, variant): plain launches instead
  void drop() {
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    exec = nullptr; graph = nullptr; blob = nullptr; variant = -1; eager = false;
  }
};
}  // namespace bhip

// per-stream contexts: device state for ONE stream + a private HIP stream + pinned staging
struct Beatrice20rc0_PhoneContext1 {
  int device = bhip::target_device();
  bhip::PhoneState st;
  hipStream_t stream = nullptr;
  float* h_io = nullptr;  // pinned: 160 in | mailbox (step counter, k, codebook pointers) | 128 out
  int hop_count = 0;      // hops done; travels to the device with the input copy (no launch spent on counting)
  int seq = 0;            // calls enqueued: mailbox word 6, written behind the phone vector in the pinned block when it is there (abi.hip flag_wait)
  // k-NN settings travel with the input copy too (mailbox words behind the audio: counter | k | cbT* | cnorm*), so
  // SetVQNumNeighbors / SetCodebook issue no device call of their own
  int vq_k = 0;
  const float* sel_cbT = nullptr;
  const float* sel_cnorm = nullptr;
  // codebook pool: every slot, the raw staging buffer and its pinned host twin are allocated in CreatePhoneContext1;
  // a table seen for the first time is uploaded stream-ordered into the least recently used slot (no hipMalloc, no wait)
  bhip::CodebookEntry pool[bhip::kCodebookPool];
  float* d_pool = nullptr;     // [kCodebookPool][128*512 + 512]
  float* d_cb_stage = nullptr; // [2][512*128] raw upload staging, alternating
  float* h_cb_stage = nullptr; // pinned twin
  hipEvent_t stage_done[2] = {nullptr, nullptr};
  bool stage_busy[2] = {false, false};
  int stage_next = 0;
  unsigned long long use_clock = 0;
  void* own_sel[3] = {nullptr, nullptr, nullptr};  // the state's own (unused) selector arrays, handed back before destroy()
  bhip::HopGraph hop_graph[2];  // [k-NN launch present]: both captured at the first hop with a given parameter blob
  struct Beatrice20rc0_PitchContext1* paired_pitch = nullptr;   // the pitch context whose call follows this context's with the same input (abi.hip: pre-execution)
  bool ok = false;
};
struct Beatrice20rc0_PitchContext1 {
  int device = bhip::target_device();
  bhip::PitchState st;
  hipStream_t stream = nullptr;
  float* h_io = nullptr;  // pinned: 160 in | mailbox (step counter, bin range, sequence) | 4 feat | 1 bin | sequence echo | previous-bin staging
  int hop_count = 0;
  int min_q = 1, max_q = BEATRICE_20RC0_PITCH_BINS - 1;  // travel with the input copy
  int seq = 0;            // calls enqueued so far: mailbox word 3, echoed by the head kernel into the pinned result block when the hop's results are there
  void* own_sel[2] = {nullptr, nullptr};
  bhip::HopGraph hop_graph;
  // Pre-execution (abi.hip, "the pitch call runs beside the phone call"): the hop a paired phone context's call enqueued here for the
  // input it was given; EstimatePitch1 claims it when it arrives with the same 160 samples, bin range and parameters, and runs the hop
  // itself otherwise.  spec_mu: held by EstimatePitch1 for its duration and by the phone call while it enqueues here.
  std::mutex spec_mu;
  struct Beatrice20rc0_PhoneContext1* paired_phone = nullptr;
  const Beatrice20rc0_PitchEstimator* spec_model = nullptr;   // the estimator of the last EstimatePitch1 (what a pre-execution runs with)
  unsigned spec_generation = 0;
  bool spec_pending = false;   // a pre-executed hop is on the stream / in h_io, unclaimed
  bool spec_launch_ok = false;
  int spec_min_q = 0, spec_max_q = 0;
  bool redo_plain = false;     // this hop's granule tags were used by a pre-execution that was not claimed: the hop runs the per-layer launches
  int committed_prev_q = 0;    // raw bin of the last hop that counted (what d_prev_q holds before the next one)
  int spec_strikes = 0;        // unclaimed pre-executions (decays with claimed ones); too many: no more for this context
  bool spec_off = false;
  long long spec_hits = 0, spec_misses = 0;
  bool ok = false;
};
struct Beatrice20rc0_WaveformContext1 {
  int device = bhip::target_device();
  bhip::WaveState st;
  hipStream_t stream = nullptr;
  float* d_inputs = nullptr;  // device: 128 phone | 4 feat | 1 bin | step counter
  float* h_io = nullptr;      // pinned: inputs | 240 out
  float* dev_d_out = nullptr; // the module's own device output buffer (the tail writes the pinned block itself: out_mapped)
  bool out_mapped = false;
  int hop_count = 0;
  int seq = 0;                // calls enqueued: travels with the inputs, written behind the 240 samples by the tail kernel (abi.hip flag_wait)
  bhip::HopGraph hop_graph;
  bool ok = false;
};
struct Beatrice20rc0_EmbeddingContext {
  int device = bhip::target_device();
  hipStream_t stream = nullptr;
  float* d_block = nullptr;
  float *d_kv_raw = nullptr, *d_tmp = nullptr, *d_add = nullptr, *d_frm = nullptr;
  float* h_stage = nullptr;  // pinned: key/value registration [384][128] | 4 vector slots (additive x2, formant x2)
  hipEvent_t vec_sent[4] = {nullptr, nullptr, nullptr, nullptr};
  bool vec_busy[4] = {false, false, false, false};
  int vec_next[2] = {0, 0};
  hipEvent_t kv_uploaded = nullptr, kv_projected = nullptr;
  bool kv_busy = false, kv_proj_pending = false;
  hipStream_t kv_proj_stream = nullptr;  // where kv_projected was recorded last
  bool ok = false;
};
