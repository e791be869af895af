"""Tick pipelining (BeatriceBatch_EnableTickPipeline): every layer of the chain its own pipeline stage, one launch per
tick, 26 steps in flight.  Must give the samples of the in-order chain bit for bit -- with per-stream settings
changing between steps (speaker switches installing one K/V block per hop, k-NN on/off, pitch and formant
settings), across pipeline drains, a stream reset in mid-flight, and the return to the in-order chain."""
import numpy as np
import pytest

from oracle_batch import oracle_leg, pick_streams, scripted_streams
from test_gpu_resident_io import Hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,steps,vq_from_start", [(24, 75, True), (5, 60, False), (256, 50, True), (600, 45, True), (1, 40, True), (17, 40, False), (33, 36, True), (40, 420, True)])
def test_tick_pipeline_matches_in_order_chain(bv, oracle, product, model_dir, B, steps, vq_from_start):
    hip = Hip()
    m = bv.Models(product, model_dir)
    bv.bind_batch(product)
    audio = np.stack([bv.synth_audio(160 * steps, seed=8000 + s) for s in range(B)])  # [B][steps*160]
    tail_steps = 4   # after the pipelined part: back to the in-order chain on the same streams

    def settings(batch):
        for s in range(B):
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % 3)
            if vq_from_start:
                batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, s % 3)
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)

    def change(batch, k):
        a, h = batch.a, batch.h
        if k % 4 == 1:
            s = (7 * k) % B
            a.BeatriceBatch_SetTargetSpeaker(h, s, (k + s) % 3)            # K/V blocks follow, one per hop
            a.BeatriceBatch_SetFormantShift(h, (s + 1) % B, float(k % 5) - 2.0)
            a.BeatriceBatch_SetPitchShift(h, (s + 2) % B, float(k % 7) - 3.0)
        if k % 9 == 5 and vq_from_start:
            a.BeatriceBatch_SetVQNumNeighbors(h, (3 * k) % B, k % 5)
        if k == 20 and not vq_from_start:
            a.BeatriceBatch_SetVQNumNeighbors(h, 1 % B, 3)                    # the k-NN stage appears: pipeline drains, table rebuilt
        if k == 33:
            assert a.BeatriceBatch_ResetStream(h, 2 % B) == 0                 # fresh state for one stream
        if k == 41:
            a.BeatriceBatch_SetMinSourcePitch(h, 0, 50.0)
            a.BeatriceBatch_SetPitchCorrection(h, 1 % B, 0.6)

    total = steps + tail_steps
    audio_all = np.concatenate([audio, audio[:, :160 * tail_steps]], axis=1)
    ref_batch = bv.Batch(m, B)
    settings(ref_batch)
    ref = []
    for k in range(total):
        change(ref_batch, k)
        ref.append(ref_batch.convert(audio_all[:, k * 160:(k + 1) * 160]))
    ref = np.stack(ref)
    ref_q = ref_batch.intermediates()[1].copy()
    ref_batch.close()

    batch = bv.Batch(m, B)
    settings(batch)
    a, h = batch.a, batch.h
    stages = a.BeatriceBatch_TickStages(h)
    assert 8 <= stages <= 48
    slots = stages + 6
    d_in, d_out = hip.malloc(slots * B * 160 * 4), hip.malloc(slots * B * 240 * 4)
    assert a.BeatriceBatch_EnableTickPipeline(h, 1) == -1          # needs resident I/O
    assert a.BeatriceBatch_BindResidentIO(h, d_in, d_out, slots) == 0
    assert a.BeatriceBatch_EnableTickPipeline(h, 1) == 0
    assert a.BeatriceBatch_EnablePipelining(h, 2) == -1
    got = np.zeros_like(ref)
    k0 = 0
    import itertools
    for chunk in itertools.cycle((slots, 7, slots - 3)):     # fill slots, feed without waiting, drain, read back; wraps around
        n = min(chunk, steps - k0)                             # (never more steps at once than there are slots)
        if n <= 0:
            break
        buf = np.zeros((slots, B, 160), np.float32)
        hip.d2h(buf, d_in)
        for k in range(k0, k0 + n):
            buf[k % slots] = audio_all[:, k * 160:(k + 1) * 160]
        hip.h2d(d_in, buf)
        for k in range(k0, k0 + n):
            change(batch, k)
            assert a.BeatriceBatch_ConvertFramesDevice(h, None, None) == 0
        assert a.BeatriceBatch_Synchronize(h) == 0
        out = np.zeros((slots, B, 240), np.float32)
        hip.d2h(out, d_out)
        for k in range(k0, k0 + n):
            got[k] = out[k % slots]
        k0 += n
    assert k0 == steps
    # back to the in-order chain: same streams, same slots, state carried over
    assert a.BeatriceBatch_EnableTickPipeline(h, 0) == 0
    buf = np.zeros((slots, B, 160), np.float32)
    for k in range(steps, total):
        buf[k % slots] = audio_all[:, k * 160:(k + 1) * 160]
    hip.h2d(d_in, buf)
    for k in range(steps, total):
        change(batch, k)
        assert a.BeatriceBatch_ConvertFramesDevice(h, None, None) == 0
    assert a.BeatriceBatch_Synchronize(h) == 0
    out = np.zeros((slots, B, 240), np.float32)
    hip.d2h(out, d_out)
    for k in range(steps, total):
        got[k] = out[k % slots]
    q = batch.intermediates()[1]
    batch.close()
    m.close()
    hip.free(d_in); hip.free(d_out)
    bad = [k for k in range(total) if not np.array_equal(ref[k], got[k])]
    print("tick pipeline B=%d, %d stages, %d steps: %s" % (B, stages, steps, "bit-identical" if not bad else
                                                          "steps that differ: %s, max-abs %g" % (bad[:12], np.abs(ref - got).max())))
    assert np.abs(got).max() > 0.05
    assert not bad
    assert np.array_equal(q, ref_q)
    # the ORACLE leg: a sample of the streams (tile corners + the streams the script addresses) as independent oracle
    # streams driven by the same script through the reference protocol -- the tick pipeline against the oracle directly
    sample = sorted(set(pick_streams(B, 6)) | set(scripted_streams(B, total, change, 6)))
    sample, want = oracle_leg(bv, oracle, model_dir, B, lambda k: audio_all[:, k * 160:(k + 1) * 160], total, settings, change, sample)
    dev = float(np.abs(got[:, sample] - want).max())
    print("tick pipeline vs ORACLE, streams %s, %d steps: max-abs %g" % (sample, total, dev))
    assert dev <= 1e-4


def test_tick_pipeline_with_a_morph_slot(bv, product, model_dir):
    """A morphed speaker draws a codebook per stream and step (processor_core_2.cc:94-121).  In tick mode the draw is made
    when the step is fed and must reach the k-NN stage eleven ticks later with THAT step -- through the settings snapshot
    of the step, not through whatever the host has drawn since.  Same seeds, same draws: tick == in order, bit for bit."""
    hip = Hip()
    m = bv.Models(product, model_dir)
    bv.bind_batch(product)
    B, steps = 12, 48
    n = m.tables.n_speakers
    audio = np.stack([bv.synth_audio(160 * steps, seed=9300 + s) for s in range(B)])
    w = np.zeros(n, np.float32)
    w[:3] = (0.5, 0.3, 0.2)

    def setup(batch):
        a, h = batch.a, batch.h
        assert a.BeatriceBatch_MorphSpeaker(h, n, bv.fptr(w), n, 1234) == 0          # entry n = the morph slot, lottery seeded
        for s in range(B):
            a.BeatriceBatch_SetTargetSpeaker(h, s, n if s % 2 == 0 else s % n)       # half the streams on the morph
            a.BeatriceBatch_SetVQNumNeighbors(h, s, 1 + s % 3)                        # k-NN on: the drawn codebook matters
        a.BeatriceBatch_FlushSpeaker(h, -1)

    ref_batch = bv.Batch(m, B)
    setup(ref_batch)
    ref = np.stack([ref_batch.convert(np.ascontiguousarray(audio[:, k * 160:(k + 1) * 160])) for k in range(steps)])
    ref_batch.close()

    batch = bv.Batch(m, B)
    setup(batch)
    a, h = batch.a, batch.h
    slots = steps   # every step has its own slot: no wrap to think about here
    d_in, d_out = hip.malloc(slots * B * 160 * 4), hip.malloc(slots * B * 240 * 4)
    hip.h2d(d_in, np.ascontiguousarray(audio.reshape(B, steps, 160).transpose(1, 0, 2)))
    assert a.BeatriceBatch_BindResidentIO(h, d_in, d_out, slots) == 0
    assert a.BeatriceBatch_EnableTickPipeline(h, 1) == 0
    for k in range(steps):
        assert a.BeatriceBatch_ConvertFramesDevice(h, None, None) == 0
    assert a.BeatriceBatch_Synchronize(h) == 0
    got = np.zeros((slots, B, 240), np.float32)
    hip.d2h(got, d_out)
    bad = [(k, int(s_)) for k in range(steps) for s_ in np.nonzero(np.abs(got[k] - ref[k]).max(axis=1))[0]]
    assert not bad, "differing (step, stream): %s" % bad[:30]
    # the draws did vary (otherwise this test shows nothing): the morph streams differ from what a fixed codebook gives
    assert np.abs(ref).max() > 0.05
    hip.free(d_in); hip.free(d_out)
    batch.close()
    m.close()
