"""Tick pipelining with TWO or FOUR hops per stage per tick (a batch created with BeatriceBatch_CreateBlock(..., 2 / 4) in tick
mode): a step is H consecutive 10 ms hops of every stream, every stage of the launch works on H times the rows, the cells of the
two recurrent layers are linked hop to hop inside the launch (H - 1 links of tagged granules), at H = 4 the 32- and 16-channel tail
stages run a step as two sub-steps.  Must give the samples of the in-order chain at one hop per step bit for bit -- with settings
that change between steps (speaker switches installing one K/V block per HOP, k-NN, pitch and formant settings), across
drains, and on the way back to the in-order chain."""
import itertools

import numpy as np
import pytest

from oracle_batch import oracle_leg, pick_streams, scripted_streams
from test_gpu_morph_device import host_lib  # noqa: F401  (fixture: the host solver for the oracle leg of the morph test)
from test_gpu_resident_io import Hip

pytestmark = pytest.mark.gpu


# (1 024 / 4 096 streams: more first-hop GRU cells than the chip has workgroup slots -- the links rest on dispatch order = index order)
@pytest.mark.parametrize("H,B,steps", [(2, 24, 40), (2, 5, 34), (2, 256, 36), (2, 1, 33), (2, 37, 70), (4, 24, 40), (4, 5, 34), (4, 256, 36), (4, 1, 33), (4, 37, 50),
                                       (2, 1024, 30), (4, 1024, 30), (2, 4096, 30)])
def test_tick_two_hops_per_step_matches_in_order_chain(bv, oracle, product, model_dir, H, B, steps):
    hip = Hip()
    m = bv.Models(product, model_dir)
    bv.bind_batch(product)
    tail_steps = 3   # after the pipelined part: the in-order chain (two hops per step) on the same streams
    total = steps + tail_steps
    audio = np.stack([bv.synth_audio(160 * H * total, seed=8100 + s) for s in range(B)])  # [B][total*320]

    def settings(batch):
        for s in range(B):
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % 3)
            batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, s % 3)
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)

    def change(batch, k):   # before STEP k (hops H k .. H k + H - 1)
        a, h = batch.a, batch.h
        if k % 4 == 1:
            s = (7 * k) % B
            a.BeatriceBatch_SetTargetSpeaker(h, s, (k + s) % 3)            # K/V blocks follow, one per hop
            a.BeatriceBatch_SetFormantShift(h, (s + 1) % B, float(k % 5) - 2.0)
            a.BeatriceBatch_SetPitchShift(h, (s + 2) % B, float(k % 7) - 3.0)
        if k % 9 == 5:
            a.BeatriceBatch_SetVQNumNeighbors(h, (3 * k) % B, k % 5)
        if k == 21:
            assert a.BeatriceBatch_ResetStream(h, 2 % B) == 0
        if k == 27:
            a.BeatriceBatch_SetMinSourcePitch(h, 0, 50.0)
            a.BeatriceBatch_SetPitchCorrection(h, 1 % B, 0.6)

    # reference: the in-order chain, one hop per step, the script applied before every H-th hop
    ref_batch = bv.Batch(m, B)
    settings(ref_batch)
    ref = np.zeros((total, B, H * 240), np.float32)
    for k in range(total):
        change(ref_batch, k)
        for hh in range(H):
            j = k * H + hh
            ref[k][:, hh * 240:(hh + 1) * 240] = ref_batch.convert(audio[:, j * 160:(j + 1) * 160])
    ref_batch.close()

    batch = bv.Batch(m, B, hops_per_step=H)
    settings(batch)
    a, h = batch.a, batch.h
    stages = a.BeatriceBatch_TickStages(h)
    assert stages == 28
    slots = stages + 5
    d_in, d_out = hip.malloc(slots * B * H * 160 * 4), hip.malloc(slots * B * H * 240 * 4)
    assert a.BeatriceBatch_BindResidentIO(h, d_in, d_out, slots) == 0
    assert a.BeatriceBatch_EnableTickPipeline(h, 1) == 0
    assert a.BeatriceBatch_EnableSilentBlockRule(h, 1) == 0 and a.BeatriceBatch_EnableSilentBlockRule(h, 0) == 0   # (round 6: whole steps sat out, tests/test_gpu_tick_ragged.py)
    got = np.zeros_like(ref)
    k0 = 0
    for chunk in itertools.cycle((slots, 7, slots - 3)):
        n = min(chunk, steps - k0)
        if n <= 0:
            break
        buf = np.zeros((slots, B, H * 160), np.float32)
        hip.d2h(buf, d_in)
        for k in range(k0, k0 + n):
            buf[k % slots] = audio[:, k * H * 160:(k + 1) * H * 160]
        hip.h2d(d_in, buf)
        for k in range(k0, k0 + n):
            change(batch, k)
            assert a.BeatriceBatch_ConvertFramesDevice(h, None, None) == 0
        assert a.BeatriceBatch_Synchronize(h) == 0
        out = np.zeros((slots, B, H * 240), np.float32)
        hip.d2h(out, d_out)
        for k in range(k0, k0 + n):
            got[k] = out[k % slots]
        k0 += n
    # back to the in-order chain: same streams, same slots, state carried over
    assert a.BeatriceBatch_EnableTickPipeline(h, 0) == 0
    buf = np.zeros((slots, B, H * 160), np.float32)
    for k in range(steps, total):
        buf[k % slots] = audio[:, k * H * 160:(k + 1) * H * 160]
    hip.h2d(d_in, buf)
    for k in range(steps, total):
        change(batch, k)
        assert a.BeatriceBatch_ConvertFramesDevice(h, None, None) == 0
    assert a.BeatriceBatch_Synchronize(h) == 0
    out = np.zeros((slots, B, H * 240), np.float32)
    hip.d2h(out, d_out)
    for k in range(steps, total):
        got[k] = out[k % slots]
    batch.close()
    m.close()
    hip.free(d_in); hip.free(d_out)
    bad = [k for k in range(total) if not np.array_equal(ref[k], got[k])]
    if bad:
        k = bad[0]
        rows = sorted(set(np.argwhere(ref[k] != got[k])[:, 0].tolist()))
        cols = np.argwhere(ref[k] != got[k])[:, 1]
        print("first differing step %d: streams %s, samples %d..%d" % (k, rows[:16], cols.min(), cols.max()))
    print("tick pipeline, %d hops per step, B=%d, %d stages, %d steps: %s" % (H, B, stages, steps, "bit-identical" if not bad else
          "steps that differ: %s, max-abs %g" % (bad[:12], np.abs(ref - got).max())))
    assert np.abs(got).max() > 0.05
    assert not bad
    # the ORACLE leg: a sample of the streams as independent oracle streams driven by the same script through the reference
    # protocol, one hop at a time (the script's changes precede a step = every second hop)
    sample = sorted(set(pick_streams(B, 6)) | set(scripted_streams(B, total, change, 6)))
    sample, want = oracle_leg(bv, oracle, model_dir, B, lambda k: audio[:, k * 160:(k + 1) * 160], total * H, settings,
                              lambda ob, hop: change(ob, hop // H) if hop % H == 0 else None, sample)
    want = want.reshape(total, H, len(sample), 240).transpose(0, 2, 1, 3).reshape(total, len(sample), H * 240)
    dev = float(np.abs(got[:, sample] - want).max())
    print("tick pipeline (%d hops per step) vs ORACLE, streams %s, %d steps: max-abs %g" % (H, sample, total, dev))
    assert dev <= 1e-4


@pytest.mark.parametrize("H", [2, 4])
def test_tick_two_hops_per_step_with_a_morph_slot(bv, oracle, product, host_lib, model_dir, H):
    """A morphed speaker draws a codebook per stream and HOP (processor_core_2.cc:94-121): the hops of a step may then use
    different codebooks (the k-NN body's separate-codebook path), and the draws must come in the order of one hop per step."""
    B, steps = 12, 40
    hip = Hip()
    m = bv.Models(product, model_dir)
    bv.bind_batch(product)
    n = m.tables.n_speakers
    audio = np.stack([bv.synth_audio(160 * H * steps, seed=8300 + s) for s in range(B)])
    w = np.zeros(n, np.float32)
    w[:3] = (0.5, 0.3, 0.2)

    def setup(batch):
        a, h = batch.a, batch.h
        assert a.BeatriceBatch_MorphSpeaker(h, n, bv.fptr(w), n, 1234) == 0
        for s in range(B):
            a.BeatriceBatch_SetTargetSpeaker(h, s, n if s % 2 == 0 else s % n)
            a.BeatriceBatch_SetVQNumNeighbors(h, s, 1 + s % 3)
        a.BeatriceBatch_FlushSpeaker(h, -1)

    ref_batch = bv.Batch(m, B)
    setup(ref_batch)
    ref = np.stack([ref_batch.convert(np.ascontiguousarray(audio[:, k * 160:(k + 1) * 160])) for k in range(H * steps)])   # [hops][B][240]
    ref_batch.close()
    ref = ref.reshape(steps, H, B, 240).transpose(0, 2, 1, 3).reshape(steps, B, H * 240)

    batch = bv.Batch(m, B, hops_per_step=H)
    setup(batch)
    a, h = batch.a, batch.h
    slots = steps
    d_in, d_out = hip.malloc(slots * B * H * 160 * 4), hip.malloc(slots * B * H * 240 * 4)
    hip.h2d(d_in, np.ascontiguousarray(audio.reshape(B, steps, H * 160).transpose(1, 0, 2)))
    assert a.BeatriceBatch_BindResidentIO(h, d_in, d_out, slots) == 0
    assert a.BeatriceBatch_EnableTickPipeline(h, 1) == 0
    for k in range(steps):
        assert a.BeatriceBatch_ConvertFramesDevice(h, None, None) == 0
    assert a.BeatriceBatch_Synchronize(h) == 0
    got = np.zeros((slots, B, H * 240), np.float32)
    hip.d2h(got, d_out)
    bad = [(k, int(s_)) for k in range(steps) for s_ in np.nonzero(np.abs(got[k] - ref[k]).max(axis=1))[0]]
    assert not bad, "differing (step, stream): %s" % bad[:30]
    assert np.abs(ref).max() > 0.05
    hip.free(d_in); hip.free(d_out)
    batch.close()
    m.close()
    # the ORACLE leg: every stream as an independent oracle stream -- the morph entry of the caller-owned tables filled by the host
    # solver, one codebook drawn per morphing stream and hop from that stream's own engine (std::mt19937(seed + stream)), as
    # tests/test_gpu_morph_device.py::test_morph_end_to_end does it
    from test_gpu_morph_device import _host_entry, _mt_draws
    mo = bv.Models(oracle, model_dir)
    t = mo.tables
    add, kv, pruned, order = _host_entry(host_lib, t, n, list(w))
    t.additive[n] = add
    t.kv[n] = kv
    odds = pruned[order[:8]]
    total = np.float32(0)
    for v in odds:
        total = np.float32(total + v)
    draws = {s_: _mt_draws(1234 + s_) for s_ in range(B)}
    streams = [bv.Stream1(mo, speaker=(n if s_ % 2 == 0 else s_ % n), vq_k=1 + s_ % 3) for s_ in range(B)]
    want = np.zeros((H * steps, B, 240), np.float32)
    for j in range(H * steps):
        for s_ in range(0, B, 2):   # the morphing streams
            r = np.float32(next(draws[s_]) * total)
            idx = int(order[0])
            for i in range(len(odds)):
                r = np.float32(r - odds[i])
                if r < 0:
                    idx = int(order[i])
                    break
            streams[s_].a.SetCodebook(streams[s_].pc, bv.fptr(t.codebooks[idx]))
        for s_ in range(B):
            want[j, s_] = streams[s_].hop(audio[s_, j * 160:(j + 1) * 160])
    for st in streams:
        st.close()
    mo.close()
    want = want.reshape(steps, H, B, 240).transpose(0, 2, 1, 3).reshape(steps, B, H * 240)
    dev = float(np.abs(got - want).max())
    print("tick pipeline (%d hops per step) with a morph slot vs ORACLE: max-abs %g" % (H, dev))
    assert dev <= 1e-4
