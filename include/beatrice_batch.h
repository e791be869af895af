/*
 * beatrice_batch.h -- batched extension of the beatrice C-ABI (this project's addition).
 *
 * The reference library is one-stream-per-context: one plugin instance = one stream = one thread
 * (reference src/vst/processor.h:56-57, factory.cc:21 kManyInstances); its per-hop entry points
 * carry the suffix "1" (ExtractPhone1 / EstimatePitch1 / GenerateWaveform1,
 * reference lib/beatricelib/beatrice.h:243-247, 266-271, 301-307).  A GPU wants many streams per
 * launch, so this header adds an N-stream object that runs the SAME three modules, in the same
 * order, for B independent streams per call, and keeps the reference host's per-stream settings
 * and call protocol (reference src/common/processor_core_2.cc):
 *
 *   reference ProcessorCore2 member            ->  batched entry point (per stream)
 *   SetTargetSpeaker        (:431-466)             BeatriceBatch_SetTargetSpeaker
 *     + one K/V block per hop (:179-181, .h:161)   (applied inside BeatriceBatch_ConvertFrames)
 *   SetFormantShift         (:468-481)             BeatriceBatch_SetFormantShift
 *   SetPitchShift / SetAverageSourcePitch /
 *   SetIntonationIntensity / SetPitchCorrection /
 *   SetPitchCorrectionType  (:483-559)             BeatriceBatch_SetPitch*   (math of :190-252 on device)
 *   SetMinSourcePitch / SetMaxSourcePitch (:561-583) BeatriceBatch_SetMin/MaxSourcePitch
 *   SetVQNumNeighbors       (:585-590)             BeatriceBatch_SetVQNumNeighbors
 *   ResetContext            (:258-291)             BeatriceBatch_ResetStream
 *   Process1                (:50-256)              BeatriceBatch_ConvertFrames (one hop, all streams)
 *
 * All functions return 0 on success and a negative value on error (bad argument -1, HIP failure -2);
 * they never throw.  `stream` = -1 addresses every stream.  Plain C types only.
 */
#ifndef BEATRICE_BATCH_H_
#define BEATRICE_BATCH_H_

#include <stddef.h>

#include "beatrice_abi.h"

BEATRICE_ABI_BEGIN

typedef struct BeatriceBatch BeatriceBatch;

/* Parameter blobs that arrive over the wire (e.g. broadcast with RCCL from rank 0) instead of from
 * a file: same validation and error codes as the Read*Parameters functions. */
Beatrice_ErrorCode BeatriceHip_LoadPhoneExtractorFromMemory(Beatrice20rc0_PhoneExtractor* m, const void* bytes, size_t size);
Beatrice_ErrorCode BeatriceHip_LoadPitchEstimatorFromMemory(Beatrice20rc0_PitchEstimator* m, const void* bytes, size_t size);
Beatrice_ErrorCode BeatriceHip_LoadWaveformGeneratorFromMemory(Beatrice20rc0_WaveformGenerator* m, const void* bytes, size_t size);
Beatrice_ErrorCode BeatriceHip_LoadEmbeddingSetterFromMemory(Beatrice20rc0_EmbeddingSetter* m, const void* bytes, size_t size);

/* Beatrice20rc0_SetCodebook (beatrice.h:318-322) is called per hop on the audio thread in morph mode and therefore recognises
 * a caller-owned table by address + a 96-word sample of its contents.  After rewriting a table IN PLACE (the same address)
 * call this, off the audio thread, so that the next SetCodebook of it uploads the new contents; NULL forgets every table. */
void BeatriceHip_InvalidateCodebook(Beatrice20rc0_PhoneContext1* ctx, const float* codebook);
/* Test hook for the recovery path of the 1-stream team launches (a launch whose workgroups could not all become resident gives its
 * waits up after ~0.3 s instead of hanging the GPU): the context's next GenerateWaveform1 returns zeros, its state restarts from
 * silence, and from then on the context runs one launch per layer.  -1: the context has no team launch. */
int BeatriceHip_InjectTeamTimeout(Beatrice20rc0_WaveformContext1* ctx);
/* ... and for the content encoder's and the pitch estimator's contexts (the next ExtractPhone1 / EstimatePitch1 returns zeros, the context restarts from
 * silence like a new one and runs one launch per layer from then on), and for a ONE-STREAM batch, whose in-order chain uses the same team launches (the
 * steps since the last BeatriceBatch_Synchronize are void: it returns -2 once, the stream restarts from silence, the batch stays usable).  -1: no team launch. */
int BeatriceHip_InjectTeamTimeoutPhone(Beatrice20rc0_PhoneContext1* ctx);
int BeatriceHip_InjectTeamTimeoutPitch(Beatrice20rc0_PitchContext1* ctx);
/* The pitch call beside the phone call.  The reference's hop calls ExtractPhone1 and EstimatePitch1 one after the other on the SAME 160 samples
 * (src/common/processor_core_2.cc:184,188); the two modules are independent.  Once a pitch context has been called right after a phone context with
 * the same samples (same thread), that phone context's calls also start the pitch context's hop for their input on the pitch context's own stream;
 * EstimatePitch1 claims that hop when it arrives with the same samples, bin range, estimator and parameters, and otherwise drops it and runs as it
 * always did -- results and state are the same to the bit either way (csrc/abi.hip, tests/test_gpu_pitch_beside_phone.py).  This entry reads the
 * counters: hops claimed / dropped so far (either pointer may be NULL); returns 1 while the context has a partner and gets such hops, 0 if not, -1 on
 * a bad context.  BEATRICE_HIP_NO_SPECULATION=1 in the environment turns the mechanism off. */
int BeatriceHip_PitchSpeculation(Beatrice20rc0_PitchContext1* ctx, long long* claimed, long long* dropped);

/* Several GPUs in one process (a C++ host with one thread per GPU, examples/node_convert.cc; the reference runs many plugin
 * instances per process, src/vst/factory.cc:21).  Every object of this library -- model objects, contexts, batches --
 * lives on ONE device, fixed when it is created: the calling thread's target device, BeatriceHip_SetDevice(ordinal)
 * (thread-local; -1 = follow the thread's current HIP device, the default), and every entry point makes its object's
 * device current for its own duration and restores the caller's, so a thread may own objects on several GPUs and a host
 * framework's own device selection is left alone.  Objects of different devices cannot be combined (a context with a
 * model object of another GPU emits zeros; BeatriceBatch_Create returns an unhealthy batch).  0, or -1 for an ordinal the
 * runtime does not have. */
int BeatriceHip_SetDevice(int ordinal);
int BeatriceHip_GetDevice(void);
int BeatriceBatch_Device(const BeatriceBatch* b);

/* Test support: the packed (two results per instruction) forms of MODEL_SPEC's scalar functions that the kernels use
 * (csrc/spec_math.hip.h) against their scalar definitions, on the device, for ALL 2^32 float32 bit patterns (NaN excluded).
 * which: 0 exp, 1 tanh, 2 gelu, 3 sigmoid.  Returns the number of inputs whose results differ in any bit (0 = identical),
 * *first_bad_bits = the lowest such input (0xffffffff when none); -1 on a HIP failure or an unknown `which`. */
long long BeatriceHip_MathSelfTest(int which, unsigned* first_bad_bits);

/* Device-resident parameter blobs, for loading a model on several GPUs from ONE file read.
 * kind: 1 phone extractor, 2 pitch estimator, 3 waveform generator, 4 embedding setter; `model` the matching object.
 * BeatriceHip_ModelBlob returns the object's parameter blob as it sits on the device (already in the kernels' packed
 * layout); with allocate != 0 an object that has not been loaded gets an empty blob of the right size.  The caller
 * moves bytes device to device (e.g. a RCCL broadcast from the rank that called Read*Parameters) and then calls
 * BeatriceHip_ModelBlobReady on the receiving objects.  0 on success, -1 bad argument / nothing to share, -2 HIP failure. */
int BeatriceHip_ModelBlob(int kind, void* model, int allocate, void** d_ptr, size_t* n_bytes);
int BeatriceHip_ModelBlobReady(int kind, void* model);

/* ---------------------------------------------------------------------------------------------------------------------------------------
 * MODES.  A batch is in exactly one of the modes below; the table says which entry point works in which (ok), which is refused with -1
 * (-), and at which hops per step H (BeatriceBatch_CreateBlock) the mode exists.  A refused call changes NOTHING: no drain, no clock, no
 * flag -- tests/test_gpu_mode_matrix.py asks every "-" cell of this table between steps and compares the batch, bit for bit, with one that
 * never asked.  The "ok" cells are the -m gpu parity tests against the oracle.  Settings (BeatriceBatch_Set*, ResetStream, Morph*,
 * SetInputGain / SetOutputGain once a wrapper is configured) work in every mode and apply to the step / call that follows them.
 *
 *   mode (how it is entered)                                        H           its entry point per step / call
 *   A  in order            (a new batch)                            1 2 4 8     ConvertFrames, ConvertFramesDevice
 *   B  stage pipelining    (EnablePipelining(2..4))                 1 2 4 8     ConvertFrames, ConvertFramesDevice
 *   C  resident I/O        (BindResidentIO)                         1 2 4 8     ConvertFramesDevice(NULL, NULL)
 *   D  tick mode           (C + EnableTickPipeline(1))              1 2 4       ConvertFramesDevice(NULL, NULL)
 *   E  host streaming      (EnableHostStreaming(1))                 1 2 4       StreamFrames / StreamFlush
 *   F  48 kHz blocks around the ticks (BindResidentIO48k)           1 2 4       ConvertBlocks48kDevice(NULL, NULL)
 *   G  resident blocks around the ticks, one set of clocks
 *                          (ConfigureWrapper + BindResidentBlocks)  1 2 4       ProcessBlocksDevice(NULL, NULL, channels, n)
 *   P  ... with clocks per stream
 *                          (ConfigureWrapperRates + BindResidentBlocksRagged)  1   ProcessBlocksRaggedDevice
 *   S  A with the silent-block rule (EnableSilentBlockRule(1))      1           ConvertBlocks48k[Device]
 *
 *   entry point                          A        B     C     D        E     F        G     P     S
 *   ConvertFrames (host buffers)         ok       ok    -     -        -     -        -     -     ok
 *   ConvertFramesDevice(d_in, d_out)     ok       ok    -     -        -     -        -     -     ok
 *   ConvertFramesDevice(NULL, NULL)      ok       ok    ok    ok       -     -        -     -     ok
 *   ConvertBlocks48k, ..Device(ptrs)     ok, H=1  -     -     -        -     -        -     -     ok
 *   ConvertBlocks48kDevice(NULL, NULL)   -        -     -     -        -     ok       -     -     -
 *   ProcessBlocks, ..Device(ptrs)        ok (1)   -     -     -        -     -        -     -     - (1)
 *   ProcessBlocksDevice(NULL, NULL)      -        -     -     -        -     -        ok    -     -
 *   FlushResidentBlocks                  -        -     -     -        -     -        ok    ok    -
 *   ProcessBlocksRagged                  ok (2)   -     -     -        -     -        -     -     - (2)
 *   ProcessBlocksRaggedDevice            -        -     -     -        -     -        -     ok    -
 *   StreamFrames / StreamFlush           -        -     -     -        ok    -        -     -     -
 *   EnableSilentBlockRule(1)             ok, H=1  -     -     ok (6)   -     ok (6)   -     -     ok
 *   EnableSilentBlockRule(0)             ok       ok    ok    ok       ok    ok       -     -     ok
 *   SetSilentStreams                     - (3)    -     -     - (3)    -     - (3)    -     -     ok
 *   EnablePipelining(n)                  ok       ok    ok    -        -     -        -     -     - (n >= 1)
 *   EnableTickPipeline(1)                -        -     ok(4) ok       -     -        -     -     -
 *   EnableTickPipeline(0)                ok       ok    ok    ok       -     -        -     -     ok
 *   EnableHostStreaming(1)               ok       -     -     -        ok    -        -     -     -
 *   BindResidentIO (bind)                ok       ok    ok    -        -     -        -     -     -
 *   BindResidentIO (NULL, NULL)          ok       ok    ok    -        -     -        -     -     ok
 *   BindResidentIO48k (bind)             ok       -     -     -        -     ok (5)   -     -     -
 *   BindResidentBlocks (bind)            ok (1)   -     -     -        -     -        ok(5) ok(5) -
 *   BindResidentBlocksRagged (bind)      ok (2)   -     -     -        -     -        ok(5) ok(5) -
 *   Bind...(NULL, NULL) of F / G / P     ok       ok    ok    ok       ok    leaves F leaves G / P      ok
 *   ConfigureWrapper                     ok       ok    ok    ok       ok    ok       -     -     ok
 *   ConfigureWrapperRates                ok, H=1  -     -     -        -     -        -     -     ok
 *   ProfileKernels                       ok       ok    ok    -        -     -        -     -     ok
 *   TimeSteps                            ok       ok    ok    ok       -     -        -     -     ok
 *   TimeTickLaunch                       -        -     -     ok       -     -        -     -     -
 *
 *   (1) once BeatriceBatch_ConfigureWrapper has been called (-1 before); one hop per step for the in-order calls, H = 1 / 2 / 4 for the binding.
 *   (2) once BeatriceBatch_ConfigureWrapperRates has been called (-1 before); one hop per step.
 *   (3) ok once the rule is enabled in that mode (D, F: any H; the flagged streams sit the NEXT step out).
 *   (4) with more resident slots than BeatriceBatch_TickStages() and at most 4096 of them, B <= 4096, H = 1 / 2 / 4.
 *   (6) at H = 2 / 4 a flagged stream sits a WHOLE step (its H hops / its H 48 kHz blocks) out: the caller flags a stream whose blocks of the step are ALL silent.  One
 *       silent block among sounding ones of the same step cannot be skipped there (it is converted as the sounding ones are) -- the shell's per-block rule needs one hop per step.
 *   (5) a bind call on a batch that is already in F / G / P first LEAVES that mode (drains, restarts the wrapper), then binds anew.
 *
 * Environment variables the library reads: BEATRICE_HIP_DEBUG (print HIP errors to stderr), BEATRICE_HIP_CUMASK ("lo-hi;lo-hi;..": CU masks
 * of the stage-pipelining streams), BEATRICE_HIP_HOP_GRAPH (the 1-stream calls replayed as hipGraphs), BEATRICE_HIP_NO_SPECULATION (no pitch hop beside
 * the phone call, BeatriceHip_PitchSpeculation above).  Nothing else: the A/B switches of
 * profiles/r0*_notes.md exist in measurement builds only (tools/debug/build_variant.sh <name> -DBEATRICE_HIP_MEASUREMENT_BUILD).
 * --------------------------------------------------------------------------------------------------------------------------------------- */
/* n_streams concurrent streams; speaker tables may hold up to max_speakers entries
 * (n_speakers + 1 when the caller keeps the reference's extra "morph" slot). */
BeatriceBatch* BeatriceBatch_Create(const Beatrice20rc0_PhoneExtractor* phone, const Beatrice20rc0_PitchEstimator* pitch,
                                    const Beatrice20rc0_WaveformGenerator* wave, const Beatrice20rc0_EmbeddingSetter* embed,
                                    int n_streams, int max_speakers);
/* Block mode for offline / utterance conversion: every step converts `hops_per_step` (1, 2, 4 or 8)
 * consecutive hops of every stream, so the per-launch cost of the kernel chain is shared by H hops.
 * Results are bit-identical to H single-hop steps (the recurrent layers still advance hop by hop
 * inside the step; a pending speaker switch still installs one K/V block per hop).  Buffers become
 * in [B][H*160], out [B][H*240].  hops_per_step = 1 is BeatriceBatch_Create (10 ms real-time steps). */
BeatriceBatch* BeatriceBatch_CreateBlock(const Beatrice20rc0_PhoneExtractor* phone, const Beatrice20rc0_PitchEstimator* pitch,
                                         const Beatrice20rc0_WaveformGenerator* wave, const Beatrice20rc0_EmbeddingSetter* embed,
                                         int n_streams, int max_speakers, int hops_per_step);
void BeatriceBatch_Destroy(BeatriceBatch* b);
int BeatriceBatch_IsHealthy(const BeatriceBatch* b);
int BeatriceBatch_InjectTeamTimeout(BeatriceBatch* b);   /* test hook, see BeatriceHip_InjectTeamTimeout */
int BeatriceBatch_NumStreams(const BeatriceBatch* b);
int BeatriceBatch_HopsPerStep(const BeatriceBatch* b);
/* bytes of per-stream activation history (all rings of all three modules) held for the n_streams streams */
size_t BeatriceBatch_StateBytes(const BeatriceBatch* b);

/* Upload the four caller-owned tables exactly as Beatrice20rc0_ReadSpeakerEmbeddings fills them
 * ([n][512][128], [n][256], [9][256], [n][384][128]); projects additive/formant vectors and the
 * K/V of every (speaker, block) once, so that later speaker switches are index changes. */
int BeatriceBatch_SetSpeakerTables(BeatriceBatch* b, int n_speakers, const float* codebooks, const float* additive,
                                   const float* formant, const float* key_value);
/* The same tables filled device to device: d_ptrs[4] / n_bytes[4] receive the raw device tables (codebooks, additive,
 * formant, key_value; sized for max_speakers entries); after writing n entries into them (e.g. by a broadcast),
 * BeatriceBatch_ProjectSpeakerTables(b, n) runs the projections that BeatriceBatch_SetSpeakerTables runs after its upload. */
int BeatriceBatch_SpeakerTablesDevice(BeatriceBatch* b, void** d_ptrs, size_t* n_bytes);
int BeatriceBatch_ProjectSpeakerTables(BeatriceBatch* b, int n_speakers);
/* Replace one table entry (e.g. the morph slot after a spherical average on the host). */
int BeatriceBatch_UpdateSpeaker(BeatriceBatch* b, int speaker, const float* codebook, const float* additive,
                                const float* key_value);

/* Speaker morphing on the device (reference processor_core_2.cc:51-177, 498-532; spherical_average.h):
 * table entry `slot` (n_weights <= slot < max_speakers, e.g. the reference's extra entry n_speakers)
 * becomes the morph of the real speakers 0..n_weights-1 under `weights`: weights below 0.01 are dropped and
 * the eight largest kept (as the reference prepares them), the additive embedding and the 384 key/value
 * embeddings are weighted spherical means (one wavefront each, one launch), projections are refreshed.
 * Streams whose target speaker is `slot` then (a) re-install the entry's key/value blocks one per hop and
 * (b) use, at every step, the codebook of ONE real speaker drawn with the weights as odds from a
 * std::mt19937 seeded with `seed` the first time (the reference seeds once, from std::random_device).  Unlike the reference
 * host, which spreads the means over five hops to bound its CPU time, the new embeddings are complete
 * when the call returns.  Agreement with the host computation: float rounding (<= 1e-5), not bit-exact. */
int BeatriceBatch_MorphSpeaker(BeatriceBatch* b, int slot, const float* weights, int n_weights, unsigned seed);
/* The lottery engine is per stream, as the reference's is per plugin instance (processor_core_2.h:48,145), and like the
 * reference's it is seeded ONCE: the first BeatriceBatch_MorphSpeaker of a batch's life seeds stream s with
 * std::mt19937(seed + s); later calls (weights moving, other entries) leave every engine running, so a caller that moves
 * morph weights every step still gets a per-hop lottery and a morph on one entry does not restart the draws of streams on
 * another.  BeatriceBatch_SeedLottery re-seeds one stream (or all, -1) with std::mt19937(seed) at any time -- e.g. from the
 * stream's global identity when streams are sharded over batches or GPUs, so that a stream's draws do not depend on where it
 * runs -- and a later BeatriceBatch_MorphSpeaker keeps those engines.  One draw per hop, also in block mode.
 * The entry's key/value projections are replaced in place by BeatriceBatch_MorphSpeaker: streams already on the entry
 * see all four blocks change at that step (the reference, which computes the means over four hops on the audio
 * thread, installs them one block per hop). */
int BeatriceBatch_SeedLottery(BeatriceBatch* b, int stream, unsigned seed);
/* The reference's own timeline of a weight change for streams that are already morphing (processor_core_2.cc:51-177,
 * 144-172): on the next hop h0 the new additive embedding and the new lottery odds are in force, the OLD key/value blocks keep
 * playing over h0 .. h0+3 (the reference computes a quarter of the key/value means per hop), the new ones are installed one
 * block per hop over h0+4 .. h0+7.  The new morph is computed into ANOTHER table entry `slot` (same rules as
 * BeatriceBatch_MorphSpeaker) and every stream whose target speaker is `from_slot` moves to it on that timeline.  -3: some
 * stream still has key/value blocks of `slot` installed (rotate over three entries when the weights move faster than every
 * eight hops; as in the reference, new blocks are then never installed while the weights keep moving: each call restarts the
 * four-hop wait). */
int BeatriceBatch_MorphSpeakerStaged(BeatriceBatch* b, int slot, int from_slot, const float* weights, int n_weights, unsigned seed);
/* Raw embeddings of a table entry as currently held on the device: additive [256], key_value [384][128]. */
int BeatriceBatch_GetSpeakerEmbeddings(BeatriceBatch* b, int speaker, float* additive, float* key_value);

int BeatriceBatch_SetTargetSpeaker(BeatriceBatch* b, int stream, int speaker);
/* n (stream, speaker) pairs at once (the same effect as n calls of BeatriceBatch_SetTargetSpeaker, processor_core_2.cc:431-466 per
 * stream; all or nothing: -1 and no change when any pair is out of range). */
int BeatriceBatch_SetTargetSpeakers(BeatriceBatch* b, int n, const int* streams, const int* speakers);
int BeatriceBatch_FlushSpeaker(BeatriceBatch* b, int stream); /* install all pending K/V blocks now */
int BeatriceBatch_SetFormantShift(BeatriceBatch* b, int stream, double formant_shift);
int BeatriceBatch_SetVQNumNeighbors(BeatriceBatch* b, int stream, int k);
int BeatriceBatch_SetMinSourcePitch(BeatriceBatch* b, int stream, double midi_note);
int BeatriceBatch_SetMaxSourcePitch(BeatriceBatch* b, int stream, double midi_note);
int BeatriceBatch_SetPitchShift(BeatriceBatch* b, int stream, double semitones);
int BeatriceBatch_SetAverageSourcePitch(BeatriceBatch* b, int stream, double midi_note);
int BeatriceBatch_SetIntonationIntensity(BeatriceBatch* b, int stream, double v);
int BeatriceBatch_SetPitchCorrection(BeatriceBatch* b, int stream, double v);
int BeatriceBatch_SetPitchCorrectionType(BeatriceBatch* b, int stream, int type);
int BeatriceBatch_ResetStream(BeatriceBatch* b, int stream);

/* One step (H hops, H = 1 unless created with BeatriceBatch_CreateBlock) for every stream.
 * Host variant: in [B][H*160] @16 kHz, out [B][H*240] @24 kHz, synchronous.
 * Device variant: pointers are device memory, work is enqueued on the batch's HIP stream and the
 * call returns without waiting (BeatriceBatch_Synchronize to wait). */
int BeatriceBatch_ConvertFrames(BeatriceBatch* b, const float* in, float* out);
int BeatriceBatch_ConvertFramesDevice(BeatriceBatch* b, const float* d_in, float* d_out);
int BeatriceBatch_Synchronize(BeatriceBatch* b);

/* Resident I/O (utterances or stream buffers that already live on the device): bind
 *   d_in [n_slots][B][H*160] and d_out [n_slots][B][H*240];
 * afterwards every BeatriceBatch_ConvertFramesDevice(b, NULL, NULL) converts the next slot (k mod n_slots) in
 * place -- no per-step copy, same results.  While bound, the host-buffer and 48 kHz entry points return -1.
 * NULL, NULL unbinds.  Re-binding restarts at slot 0. */
int BeatriceBatch_BindResidentIO(BeatriceBatch* b, const float* d_in, float* d_out, int n_slots);

/* 48 kHz host-rate blocks with the reference's wrapper ON THE DEVICE (BASELINE.json configs[4]):
 * per call one 10 ms block of every stream, planar float, `channels` = 1 or 2:
 *   in [B][channels][480] @48 kHz -> out [B][channels][480] @48 kHz.
 * Performs, per stream, what the reference host does around the model for a 48 kHz host at 0 dB gains:
 * stereo downmix (L+R)*0.5 (reference src/vst/processor.cc:183-192), 31-tap low-pass + keep every 3rd
 * sample (src/common/resample.h:130-159, 384-386), the 480-sample FIFO (:343-363, i.e. +10 ms), the
 * model hop, zero-stuffing x2 (:390-393), 32-tap low-pass (:168-206), copy to every output channel
 * (processor.cc:221-225).  Bit-identical to the host chain.  Other host rates and non-zero gains: use
 * the C++ host layer (beatrice-vst_amd/host).  Per 10 ms block only: returns -1 on a block-mode batch (H > 1). */
int BeatriceBatch_ConvertBlocks48k(BeatriceBatch* b, const float* in, float* out, int channels);
int BeatriceBatch_ConvertBlocks48kDevice(BeatriceBatch* b, const float* d_in, float* d_out, int channels);
/* Throughput form: n_slots (> BeatriceBatch_TickStages()) resident 48 kHz blocks per direction, [n_slots][B][channels][480], with
 * the tick pipeline between them (the batch allocates its own 16 / 24 kHz slots).  While bound, block k --
 * BeatriceBatch_ConvertBlocks48kDevice(b, NULL, NULL, channels) -- is taken from slot k mod n_slots, and its converted block
 * appears in the same slot of d_out48 BeatriceBatch_TickStages() - 1 calls later or after BeatriceBatch_Synchronize: the
 * samples of the in-order call, later.  NULL, NULL unbinds (and leaves tick mode).
 * On a batch with H = 2 or 4 hops per step (BeatriceBatch_CreateBlock(..., H)) a slot holds H consecutive blocks per stream,
 * [n_slots][B][H][channels][480], and every call converts them all (the samples of H in-order calls). */
int BeatriceBatch_BindResidentIO48k(BeatriceBatch* b, const float* d_in48, float* d_out48, int channels, int n_slots);
/* The shell's rule "a block whose down-mix is all zeros is not converted: the core is not called, its state and its 10 ms FIFO
 * stand still, the output is that down-mix" (reference src/vst/processor.cc:204-214), PER STREAM, for the in-order 48 kHz blocks
 * (one hop per step, no pipelining).  Enabled, BeatriceBatch_ConvertBlocks48k finds the silent streams of every call itself with
 * the shell's own test; for BeatriceBatch_ConvertBlocks48kDevice the caller flags them (flags[B], non-zero = silent) before the
 * call, once per block.  A silent stream's model state, wrapper state, pending key/value installs and codebook lottery stand
 * still as if the block had never existed (the batch runs the step for every stream and puts the silent ones back: rings rotated,
 * in-place state restored).
 * In TICK MODE (BeatriceBatch_EnableTickPipeline, or the 48 kHz blocks around the ticks, BeatriceBatch_BindResidentIO48k; one hop
 * per step): enable the rule after entering the mode; the streams flagged with BeatriceBatch_SetSilentStreams sit the NEXT step out --
 * the step carries each stream's own step counter through the pipeline, a second instance of the launch runs from then on.  At
 * every drained point (BeatriceBatch_Synchronize, leaving the mode) the streams are brought back to the batch's one counter
 * (their rings rotated by the steps they missed), so the common launch runs again and BeatriceBatch_EnableTickPipeline(b, 0) /
 * BeatriceBatch_BindResidentIO48k(b, NULL, NULL, ..) succeed as on any batch.  A batch of several hops per step takes the rule in tick mode
 * as well (round 6; plain or with the 48 kHz blocks around the ticks): a flagged stream then sits a WHOLE step -- its H hops, its H 48 kHz blocks, which come back as
 * their own down-mix -- out; a single silent block among sounding ones of a step is converted like them (the shell's per-block rule: one hop per step).
 * Returns -1 under host streaming or resident wrapper blocks, and (in order) with resident I/O, stage pipelining or several hops per step. */
int BeatriceBatch_EnableSilentBlockRule(BeatriceBatch* b, int enable);
int BeatriceBatch_SetSilentStreams(BeatriceBatch* b, const unsigned char* flags);

/* The same wrapper at ANY host rate and block size, with the dB-ramped gains (the whole of the reference's
 * ProcessorCore2::Process, src/common/processor_core_2.cc:24-48, per stream on the device): input gain (gain.h:41-71,
 * 2 dB/ms towards the target) -> host rate to 48 kHz (resample.h:130-237: Stern-Brocot ratio, Hann-windowed sinc tables) ->
 * exact-480 FIFO -> every third sample -> model hop -> zero-stuffing -> 48 kHz to host rate -> output gain -> every channel.
 * BeatriceBatch_ConfigureWrapper sets the host rate for the batch (restarting resampler and FIFO, like SetSampleRate);
 * BeatriceBatch_ProcessBlocks[Device] converts one block of n samples per stream, [B][channels][n] planar, channels 1 or 2
 * (stereo is down-mixed (L+R)*0.5 as src/vst/processor.cc:183-192 does; the shell's skip of an all-zero block, :204-214,
 * is NOT applied per stream -- the streams of a batch advance together, a silent stream is converted like any other; the
 * one-stream form of that rule is ProcessorProxy::ProcessChannels, beatrice_host.h); n may change from call to call, up to
 * BeatriceBatch_MaxWrapperBlock(b) (4088 samples at the higher of the two rates).  A model hop runs whenever 480 samples
 * at 48 kHz have accumulated (0, 1 or several times per call).  Bit-identical to the host chain.  One 10 ms hop per step
 * batches only, pipelining off (a batch of several hops per step takes BeatriceBatch_ConfigureWrapper for
 * BeatriceBatch_BindResidentBlocks, below; the in-order calls return -1 on it). */
int BeatriceBatch_ConfigureWrapper(BeatriceBatch* b, double host_sample_rate);
int BeatriceBatch_SetInputGain(BeatriceBatch* b, int stream, double gain_db);
int BeatriceBatch_SetOutputGain(BeatriceBatch* b, int stream, double gain_db);
int BeatriceBatch_ProcessBlocks(BeatriceBatch* b, const float* in, float* out, int channels, int n_samples);
int BeatriceBatch_ProcessBlocksDevice(BeatriceBatch* b, const float* d_in, float* d_out, int channels, int n_samples);
int BeatriceBatch_MaxWrapperBlock(const BeatriceBatch* b);
/* The same wrapper with clocks PER STREAM -- what the reference gives every plugin instance (src/common/resample.h:401-438,
 * processor_core_2.h:28): a batch whose streams come from hosts at different rates and block sizes.
 * BeatriceBatch_ConfigureWrapperRates: rates[B], the host rate of each stream (restarts every stream's resampler pair and FIFO;
 * gains keep their state; the uniform entry points above are off until BeatriceBatch_ConfigureWrapper is called again).
 * BeatriceBatch_ProcessBlocksRagged: for every stream s one block of n_samples[s] samples at its rate (0: the stream sits the
 * call out); in / out (host): the streams' planar blocks [channels][n_samples[s]] one after the other.  A stream fires a model
 * hop whenever ITS 480-sample FIFO fills; a model step runs for the streams that fire in it, the others stand still.
 * apply_silent_rule != 0: the shell's rule per stream (src/vst/processor.cc:204-214): a block whose down-mix is all zeros is
 * not converted -- gains, resampler clocks, FIFO, model state, key/value installs and codebook lottery of that stream do not
 * move, its output block is zeros.  In-order mode.  Bit-identical to one reference wrapper per stream. */
int BeatriceBatch_ConfigureWrapperRates(BeatriceBatch* b, const double* rates);
int BeatriceBatch_ProcessBlocksRagged(BeatriceBatch* b, const float* in, float* out, int channels, const int* n_samples, int apply_silent_rule);
/* Throughput form of the any-rate wrapper: the TICK pipeline between resident host-rate blocks (as BeatriceBatch_BindResidentIO48k
 * is for 48 kHz / 0 dB).  d_in / d_out: [n_slots][B][channels][n_samples] planar, at the rate of BeatriceBatch_ConfigureWrapper.
 * Call k = BeatriceBatch_ProcessBlocksDevice(b, NULL, NULL, channels, n_samples) reads slot k mod n_slots; its output block is in
 * the same slot of d_out BeatriceBatch_ResidentBlocksDelay() (= TickStages() - 1) calls later, or after BeatriceBatch_Synchronize.
 * Gains (BeatriceBatch_SetInputGain / SetOutputGain) and every per-stream setting apply to the call that follows them, as in
 * order.  Same samples as BeatriceBatch_ProcessBlocksDevice in order.  n_slots >= TickStages() + 1.  NULL pointers unbind.
 * Binding and unbinding restart the wrapper (resampler histories, FIFO) as BeatriceBatch_ConfigureWrapper does; gains keep their state.
 * On a batch with H = 2 or 4 hops per step (BeatriceBatch_CreateBlock(..., H); BeatriceBatch_ConfigureWrapper accepts it for this
 * entry point only): the model hops the FIFOs fire are collected H at a time -- hop k of the batch is hop k mod H of step k / H --
 * and a step enters the pipeline when it is full, so every tick launch carries H hops per stream as in plain tick mode; ticks run
 * only then (no idle tick per call).  A call's block is out once the step of its newest hop has filled and TickStages() - 1 further
 * steps have gone in behind it: BeatriceBatch_ResidentBlocksDelay() = the calls that bring ((H - 1) + (TickStages() - 1) x H) x 480
 * inner samples, + 1 (the output half rides in the next call's launch: one launch per call beside the tick launches) (ask BeatriceBatch_ResidentBlocksDelayFor(b, n_samples) before binding: n_slots >= that + 2; -1 for blocks shorter
 * than two inner samples).  BeatriceBatch_Synchronize drains the ticks and completes the calls whose hops' steps are full; the last
 * calls, which end on a step still filling, stay owed (BeatriceBatch_ResidentBlocksOwed, 0 at one hop per step) until later calls
 * fill it -- an offline caller ends a file with BeatriceBatch_FlushResidentBlocks (or with that many blocks of silence).  Same samples as one
 * hop per step. */
int BeatriceBatch_BindResidentBlocks(BeatriceBatch* b, const float* d_in, float* d_out, int channels, int n_samples, int n_slots);
int BeatriceBatch_ResidentBlocksDelay(const BeatriceBatch* b);
int BeatriceBatch_ResidentBlocksDelayFor(const BeatriceBatch* b, int n_samples);
int BeatriceBatch_ResidentBlocksOwed(const BeatriceBatch* b);
/* End of the material (reference src/common/resample.h:331-364: the FIFO hands a block's samples out one block later -- whatever follows):
 * every call made so far gets its output block, as if blocks of silence had followed -- the library completes the step that is still filling
 * with hops of silence itself (no caller slot is read or written for them), drains the pipeline, runs the output halves still owed --, then
 * the binding STARTS OVER as a new one does: resampler pair and FIFO restarted (the gains keep their state), the next call is call 0 and reads
 * slot 0; the streams' model state carries on.  BeatriceBatch_ResidentBlocksOwed() is 0 afterwards.  With nothing owed (always so at one hop
 * per step and with clocks per stream) it is BeatriceBatch_Synchronize and the binding is left as it is.  -1 without a binding. */
int BeatriceBatch_FlushResidentBlocks(BeatriceBatch* b);
/* The throughput form with clocks PER STREAM (reference src/common/resample.h:401-438: every plugin instance owns its resampler
 * pair; here a batch whose streams come from hosts at 44.1, 48, 96 kHz ... with block sizes of their own, around ONE tick pipeline).
 * After BeatriceBatch_ConfigureWrapperRates: d_in / d_out = [n_slots][B][channels * max_samples]; stream s's block of a call, planar
 * [channels][n_samples[s]], sits at the start of its cell.  Call k = BeatriceBatch_ProcessBlocksRaggedDevice(b, n_samples) reads slot
 * k mod n_slots: n_samples[B], 0 <= n_samples[s] <= max_samples (and the per-rate limit of BeatriceBatch_ProcessBlocksRagged),
 * 0 = the stream sits the call out (nothing of it moves, no output).  A stream fires a model hop when ITS 480-sample FIFO fills; the
 * step that enters the ticks carries the streams that fired, the others sit it out with step counters of their own (the tick launch's
 * ragged steps).  The output blocks of call k are in the same slot of d_out BeatriceBatch_ResidentBlocksDelay() (= TickStages() - 1)
 * calls later, or after BeatriceBatch_Synchronize.  Gains and per-stream settings apply to the call that follows them.  Same samples
 * as BeatriceBatch_ProcessBlocksRagged in order = one reference wrapper per stream.  One hop per step, n_slots >= TickStages() + 1.
 * NULL pointers unbind (either bind call unbinds either form); binding and unbinding restart every stream's resampler pair and FIFO. */
int BeatriceBatch_BindResidentBlocksRagged(BeatriceBatch* b, const float* d_in, float* d_out, int channels, int max_samples, int n_slots);
int BeatriceBatch_ProcessBlocksRaggedDevice(BeatriceBatch* b, const int* n_samples);

/* Execution control: use an externally owned hipStream_t (e.g. the framework's current stream);
 * replay the per-hop kernel chain from a captured hipGraph (default on). */
int BeatriceBatch_SetStream(BeatriceBatch* b, void* hip_stream);
void* BeatriceBatch_GetStream(const BeatriceBatch* b);
int BeatriceBatch_EnableGraph(BeatriceBatch* b, int enable);
/* Throughput mode for callers that enqueue steps ahead of their completion (BeatriceBatch_ConvertFramesDevice
 * without waiting, resident I/O).  The per-hop chain is ~40 dependent, latency-bound launches that leave most of
 * the chip idle; with a pipeline depth n = 2..4 it is cut into n stages (front end = content encoder + pitch
 * estimator; the waveform generator in 1..3 parts) on n HIP streams, and stage s of step t+1 overlaps stage s+1 of
 * step t.  Same samples (bit for bit).  enable: 0 = off (default: everything in order on the batch's stream),
 * 1 = depth 2, 2..4 = that depth.  A step's output is complete after BeatriceBatch_Synchronize, or in stream order
 * on BeatriceBatch_GetWaveStream.  The 48 kHz entry points need it off.  Depth 4 needs more hardware queues than ROCm
 * hands a process by default (environment GPU_MAX_HW_QUEUES=8, set before the HIP runtime starts): streams that
 * share a queue do not overlap, and depth 4 then runs slower than depth 3. */
int BeatriceBatch_EnablePipelining(BeatriceBatch* b, int enable);
void* BeatriceBatch_GetWaveStream(const BeatriceBatch* b);
/* Tick pipelining: the deepest form of the same idea, on ONE stream.  Every layer of the chain is its own pipeline
 * stage (the conditioned blocks: two stages each, as row-local chains); each BeatriceBatch_ConvertFramesDevice(b, NULL, NULL)
 * is one "tick" -- a single launch of 512-thread workgroups in which stage s works on the step fed s ticks ago, so ~26
 * steps are in flight and their ~1900 independent workgroups (at 256 streams) fill the chip two per CU, with the kernel boundary between
 * ticks as the only synchronisation.  Same samples, bit for bit; a step's output lands
 * in its resident-I/O slot BeatriceBatch_TickStages() - 1 ticks after its input was fed, and BeatriceBatch_Synchronize
 * drains the pipeline (that many ticks without new input).  Settings changed between steps apply to exactly the step
 * they precede.  Requirements (-1 otherwise): one, two or four hops per step, at most 4096 streams (tested to 4096), resident I/O bound with more slots
 * than stages and at most 4096 of them, and the caller must leave a step's INPUT slot untouched for BeatriceBatch_TickStages() further steps.
 * The host-buffer, 48 kHz and profiling entry points return -1 while it is on.
 * H = 2 or 4 hops per step (a batch from BeatriceBatch_CreateBlock(..., H); slots of [B][H * 160] in, [B][H * 240] out): every stage
 * works on all hops of its step in one launch -- the fixed cost of a launch is paid once per H hops, and a short run has fewer
 * partly filled launches per hop (256 streams, 20 steps + drain: 2.72 / 3.44 / 3.9 M frames/s at 1 / 2 / 4 hops per step; steady
 * 3.8 / 4.35 / 4.38 M; 64 speakers on 256 streams: 2.30 / 3.01 / 3.45 M), same samples as one hop per step, settings still apply
 * per step and key/value installs per hop.  The 48 kHz wrapper around the ticks takes such a batch too
 * (BeatriceBatch_BindResidentIO48k: a slot then holds H consecutive blocks per stream, [B][H][channels][480], and a call converts
 * them all: 64 stereo streams 1.41 / 2.32 / 2.9 M frames/s), and so does host streaming (BeatriceBatch_StreamFrames then takes
 * [B][H * 160] and returns [B][H * 240]) and the any-rate wrapper around the ticks (BeatriceBatch_BindResidentBlocks: a step enters the
 * pipeline when the FIFOs have fired H hops); the silent-block rule needs one hop per step. */
int BeatriceBatch_EnableTickPipeline(BeatriceBatch* b, int enable);
int BeatriceBatch_TickStages(const BeatriceBatch* b);
/* Host streaming: tick pipelining for callers whose audio lives in HOST memory (offline conversion of files, a network
 * front end).  BeatriceBatch_EnableHostStreaming(b, 1) gives the batch its own resident slots + pinned mirrors and turns
 * tick mode on (requirements as above, nothing else bound); every BeatriceBatch_StreamFrames(b, in, out) then takes one
 * hop of every stream ([B][160] in) and -- once the pipeline is full -- returns 1 with `out` ([B][240]) holding the samples
 * of the step fed BeatriceBatch_HostStreamDelay() calls earlier (0 while filling: `out` untouched).  The resident slots ARE
 * the pinned host mirrors: the first stages read their hop and the last stage writes its samples over PCIe themselves, so
 * there is no copy command and no second stream, and the transfers hide among the tick's workgroups.  After the last
 * input, BeatriceBatch_StreamFlush(b, out) returns the remaining steps one per call (1, then 0 when none is left).
 * Same samples as BeatriceBatch_ConvertFrames, bit for bit.  This is the host-buffer form of what src/common's
 * ProcessorCore2::Process does per stream (processor_core_2.cc:24-48: in -> model hop -> out), for a whole batch. */
int BeatriceBatch_EnableHostStreaming(BeatriceBatch* b, int enable);
int BeatriceBatch_HostStreamDelay(const BeatriceBatch* b);
int BeatriceBatch_StreamFrames(BeatriceBatch* b, const float* in, float* out);
int BeatriceBatch_StreamFlush(BeatriceBatch* b, float* out);
/* Measurement hook (tick mode on, pipeline full): `ticks` (<= 64) more ticks back to back between one pair of HIP events
 * on the batch's stream: mean microseconds per launch (boundary to the next launch included) + the launch's algorithmic
 * FLOPs and bytes. */
int BeatriceBatch_TimeTickLaunch(BeatriceBatch* b, int ticks, float* us_per_launch, double* flops, double* bytes);
/* Optional: capture the hipGraphs of the current mode now (settings, I/O binding, pipelining as they stand; nothing
 * runs), so that the first steps do not spend milliseconds on it.  Otherwise it happens inside the first step. */
int BeatriceBatch_Prepare(BeatriceBatch* b);

/* Resident buffers of the batch ([B][H*160] in, [B][H*240] out) for callers that produce / consume
 * audio on the device. */
float* BeatriceBatch_DeviceInput(BeatriceBatch* b);
float* BeatriceBatch_DeviceOutput(BeatriceBatch* b);

/* Test hook: copies the last step's intermediate results (any pointer may be NULL):
 * phone [B][H][128], raw bins [B][H], transformed bins [B][H], features [B][H][4]. */
int BeatriceBatch_GetIntermediates(BeatriceBatch* b, float* phone, int* q_raw, int* q, float* feat);

/* Measurement hook: runs `steps` steps (H hops each) on the resident input buffer, timed with HIP events recorded
 * on the batch's own stream; returns total milliseconds in *ms. */
int BeatriceBatch_TimeSteps(BeatriceBatch* b, int steps, float* ms);

/* Per-kernel measurement hook: runs ONE hop eagerly (no graph) with every launch of the chain
 * bracketed by HIP events on the batch's stream; each launch is issued `repeats` times back to back
 * inside its bracket and the bracket time is divided by `repeats` (launches that update state in place -- the GRUs,
 * the pitch head, the upsampler tail -- are issued once, so the profiled hop is an ordinary hop of the stream).  Launches with the same name are accumulated.  Fills up to max_entries rows:
 * names (64 bytes each, NUL terminated), launches per hop, mean microseconds per launch, and the
 * algorithmic FLOPs and bytes of one launch.  Returns the number of rows. */
int BeatriceBatch_ProfileKernels(BeatriceBatch* b, int repeats, int max_entries, char* names, int* launches,
                                 double* mean_us, double* flops, double* bytes);

BEATRICE_ABI_END

#endif /* BEATRICE_BATCH_H_ */
