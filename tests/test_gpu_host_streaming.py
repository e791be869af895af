"""BeatriceBatch_StreamFrames: the tick pipeline fed from and drained to host buffers (uploads, ticks and downloads on
three HIP streams).  Must hand back, with a fixed delay, exactly the samples BeatriceBatch_ConvertFrames gives."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,steps,vq_from_start", [(7, 90, True), (256, 70, True), (6, 80, False), (9, 330, True)])
def test_streamed_host_buffers_match_in_order_chain(bv, oracle, product, model_dir, B, steps, vq_from_start):
    m = bv.Models(product, model_dir)
    bv.bind_batch(product)
    audio = np.stack([bv.synth_audio(160 * steps, seed=4100 + s) for s in range(B)]).reshape(B, steps, 160)

    def settings(batch):
        for s in range(B):
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % 3)
            if vq_from_start:
                batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, s % 4)
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)

    def change(batch, k):
        if k % 11 == 3:
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, (5 * k) % B, (k + 1) % 3)
            batch.a.BeatriceBatch_SetPitchShift(batch.h, (3 * k) % B, float(k % 5) - 2.0)
        if k == 31 and not vq_from_start:
            batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, 1 % B, 3)     # the k-NN stage appears: the pipeline drains in mid-stream
        if k == 44:
            assert batch.a.BeatriceBatch_ResetStream(batch.h, 2 % B) == 0   # drains, too

    ref_batch = bv.Batch(m, B)
    settings(ref_batch)
    ref = []
    for k in range(steps):
        change(ref_batch, k)
        ref.append(ref_batch.convert(np.ascontiguousarray(audio[:, k])))
    ref_batch.close()

    batch = bv.Batch(m, B)
    settings(batch)
    a, h = batch.a, batch.h
    out = np.zeros((B, 240), np.float32)
    assert a.BeatriceBatch_StreamFrames(h, bv.fptr(np.ascontiguousarray(audio[:, 0])), bv.fptr(out)) == -1   # not enabled
    assert a.BeatriceBatch_EnableHostStreaming(h, 1) == 0
    assert a.BeatriceBatch_EnableTickPipeline(h, 0) == -1      # owned by the host streaming mode while that is on
    delay = a.BeatriceBatch_HostStreamDelay(h)
    assert 8 <= delay <= 64
    got = []
    for k in range(steps):
        change(batch, k)
        x = np.ascontiguousarray(audio[:, k])
        rc = a.BeatriceBatch_StreamFrames(h, bv.fptr(x), bv.fptr(out))
        assert rc in (0, 1)
        if rc == 1:
            got.append(out.copy())
        else:
            assert len(got) == 0 and k < delay + 2      # only while the pipeline fills
    while True:
        rc = a.BeatriceBatch_StreamFlush(h, bv.fptr(out))
        assert rc in (0, 1)
        if rc == 0:
            break
        got.append(out.copy())
    assert len(got) == steps
    bad = [(k, int(s_)) for k in range(steps) for s_ in np.nonzero(np.abs(got[k] - ref[k]).max(axis=1))[0]]
    assert not bad, "differing (step, stream): %s" % bad[:40]
    # the ORACLE leg: sampled streams as independent oracle streams under the same script (reference protocol)
    from oracle_batch import oracle_leg, pick_streams, scripted_streams
    sample = sorted(set(pick_streams(B, 6)) | set(scripted_streams(B, steps, change, 6)))
    sample, want = oracle_leg(bv, oracle, model_dir, B, lambda k: audio[:, k], steps, settings, change, sample)
    dev = float(np.abs(np.stack(got)[:, sample] - want).max())
    print("host streaming vs ORACLE, streams %s, %d steps: max-abs %g" % (sample, steps, dev))
    assert dev <= 1e-4
    # back to the in-order chain on the same streams: state carried over
    assert a.BeatriceBatch_EnableHostStreaming(h, 0) == 0
    batch.close()
    m.close()


@pytest.mark.parametrize("H,B,steps", [(2, 7, 60), (2, 256, 45), (4, 7, 50), (4, 256, 36)])
def test_streamed_host_buffers_two_hops_per_step(bv, oracle, product, model_dir, H, B, steps):
    """Host streaming with a batch of two or four hops per step: a call takes [B][H x 160] and hands back [B][H x 240] (the tick
    pipeline with H hops per stage per launch underneath).  Must equal the in-order chain at one hop per step, the script applied
    before a step -- and a sample of the streams the ORACLE driven through the reference protocol."""
    m = bv.Models(product, model_dir)
    bv.bind_batch(product)
    audio = np.stack([bv.synth_audio(160 * H * steps, seed=4700 + s) for s in range(B)])

    def settings(batch):
        for s in range(B):
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % 3)
            batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, s % 4)
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)

    def change(batch, k):
        if k % 11 == 3:
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, (5 * k) % B, (k + 1) % 3)
            batch.a.BeatriceBatch_SetPitchShift(batch.h, (3 * k) % B, float(k % 5) - 2.0)
        if k == 29:
            assert batch.a.BeatriceBatch_ResetStream(batch.h, 2 % B) == 0   # drains

    ref_batch = bv.Batch(m, B)
    settings(ref_batch)
    ref = np.zeros((steps, B, H * 240), np.float32)
    for k in range(steps):
        change(ref_batch, k)
        for hh in range(H):
            j = H * k + hh
            ref[k][:, hh * 240:(hh + 1) * 240] = ref_batch.convert(np.ascontiguousarray(audio[:, j * 160:(j + 1) * 160]))
    ref_batch.close()

    batch = bv.Batch(m, B, hops_per_step=H)
    settings(batch)
    a, h = batch.a, batch.h
    out = np.zeros((B, H * 240), np.float32)
    assert a.BeatriceBatch_EnableHostStreaming(h, 1) == 0
    got = []
    for k in range(steps):
        change(batch, k)
        x = np.ascontiguousarray(audio[:, k * H * 160:(k + 1) * H * 160])
        rc = a.BeatriceBatch_StreamFrames(h, bv.fptr(x), bv.fptr(out))
        assert rc in (0, 1)
        if rc == 1:
            got.append(out.copy())
    while True:
        rc = a.BeatriceBatch_StreamFlush(h, bv.fptr(out))
        assert rc in (0, 1)
        if rc == 0:
            break
        got.append(out.copy())
    assert len(got) == steps
    bad = [(k, int(s_)) for k in range(steps) for s_ in np.nonzero(np.abs(got[k] - ref[k]).max(axis=1))[0]]
    assert not bad, "differing (step, stream): %s" % bad[:40]
    assert np.abs(ref).max() > 0.05
    assert a.BeatriceBatch_EnableHostStreaming(h, 0) == 0
    batch.close()
    m.close()
    # the ORACLE leg: sampled streams as independent oracle streams, one hop at a time, the script before every H-th hop
    from oracle_batch import oracle_leg, pick_streams, scripted_streams
    sample = sorted(set(pick_streams(B, 6)) | set(scripted_streams(B, steps, change, 6)))
    sample, want = oracle_leg(bv, oracle, model_dir, B, lambda j: audio[:, j * 160:(j + 1) * 160], steps * H, settings,
                              lambda ob, hop: change(ob, hop // H) if hop % H == 0 else None, sample)
    want = want.reshape(steps, H, len(sample), 240).transpose(0, 2, 1, 3).reshape(steps, len(sample), H * 240)
    dev = float(np.abs(np.stack(got)[:, sample] - want).max())
    print("host streaming (%d hops per call) vs ORACLE, streams %s, %d steps: max-abs %g" % (H, sample, steps, dev))
    assert dev <= 1e-4
