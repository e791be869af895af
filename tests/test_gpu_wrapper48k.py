"""BASELINE.json configs[4] path: 48 kHz (stereo) blocks with the resampling wrapper on the device
(BeatriceBatch_ConvertBlocks48k) against the host chain: oracle wrapper (pinned to the reference's
resample.h) around the oracle model, per stream."""
import numpy as np
import pytest

import wrapperlib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,channels", [(3, 2), (20, 1)])
def test_device_wrapper_48k_matches_host_chain(bv, oracle, product, model_dir, B, channels):
    blocks = 14
    rng_sig = [wrapperlib.test_signal(480 * blocks, 48000, seed=900 + 2 * s) for s in range(B)]
    rng_sig2 = [wrapperlib.test_signal(480 * blocks, 48000, seed=901 + 2 * s) for s in range(B)]
    x = np.zeros((B, channels, 480 * blocks), np.float32)
    for s in range(B):
        x[s, 0] = rng_sig[s]
        if channels == 2:
            x[s, 1] = 0.7 * rng_sig2[s]
    # host reference per stream
    mo = bv.Models(oracle, model_dir)
    want = np.zeros((B, 480 * blocks), np.float32)
    for s in range(B):
        st = bv.Stream1(mo, speaker=s % 3, vq_k=s % 2)

        def hop(in160, out240, _u, st=st):
            o = st.hop(np.ctypeslib.as_array(in160, (160,)).copy())
            for i in range(240):
                out240[i] = o[i]

        mono = x[s, 0] if channels == 1 else ((x[s, 0] + x[s, 1]) * np.float32(0.5)).astype(np.float32)
        want[s] = wrapperlib.oracle_wrapper().run_chain(48000, mono, 480, hop=hop)
        st.close()
    mo.close()
    # device
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    for s in range(B):
        batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % 3)
        batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, s % 2)
    batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)
    got = np.zeros_like(x)
    for k in range(blocks):
        got[:, :, 480 * k:480 * (k + 1)] = batch.convert48k(x[:, :, 480 * k:480 * (k + 1)], channels)
    batch.close()
    m.close()
    dev = max(float(np.abs(got[:, c] - want).max()) for c in range(channels))
    print("48k wrapper B=%d ch=%d max-abs %g" % (B, channels, dev))
    assert np.abs(want).max() > 1e-3
    if channels == 2:
        assert np.array_equal(got[:, 0], got[:, 1])
    assert dev <= 1e-4


def oracle_leg_48k(bv, oracle, model_dir, B, channels, x, blocks, settings, change_block, got, n_pick=4):
    """The ORACLE leg of the 48 kHz tests: a sample of the streams as the host chain -- the ref-pinned wrapper oracle around an
    independent oracle stream, `change_block(script, k)` applied before block k; a reset restarts the stream's wrapper too
    (SetSampleRate semantics).  got[k] = [B][channels][480].  Returns (sample, max-abs)."""
    import ctypes as C
    from oracle_batch import OracleBatch, pick_streams, scripted_streams
    sample = sorted(set(pick_streams(B, n_pick)) | set(scripted_streams(B, blocks, change_block, n_pick)))
    ob = OracleBatch(bv, oracle, model_dir, B, sample=sample)
    settings(ob)
    wl = wrapperlib.oracle_wrapper()
    wrappers, callbacks = {}, {}

    def make_wrapper(s):
        def hop(in160, out240, _u, s=s):
            o = ob.st[s]["s1"].hop(np.ctypeslib.as_array(in160, (160,)).copy())
            C.memmove(out240, o.ctypes.data, 240 * 4)
        if s in wrappers:
            wl.f_destroy(wrappers[s])
        callbacks[s] = wrapperlib.HOP_FN(hop)
        wrappers[s] = wl.f_create(48000.0, callbacks[s], None)

    class Script:   # what the script sees: the oracle streams, plus the wrapper restart that goes with a stream reset
        def __init__(self):
            self.h, self.a = None, self

        def BeatriceBatch_ResetStream(self, h_, stream):
            rc = ob.a.BeatriceBatch_ResetStream(h_, stream)
            if stream in wrappers:
                make_wrapper(stream)
            return rc

        def __getattr__(self, name):
            return getattr(ob.a, name)

    script = Script()
    for s_ in ob.sample:
        make_wrapper(s_)
    dev = 0.0
    for k in range(blocks):
        change_block(script, k)
        for s_ in ob.sample:
            blk_in = x[s_, :, 480 * k:480 * (k + 1)]
            mono = np.ascontiguousarray(blk_in[0] if channels == 1 else ((blk_in[0] + blk_in[1]) * np.float32(0.5)).astype(np.float32))
            y = np.zeros(480, np.float32)
            assert wl.f_process(wrappers[s_], bv.fptr(mono), bv.fptr(y), 480) == 0
            for c in range(channels):
                dev = max(dev, float(np.abs(got[k][s_, c] - y).max()))
    for s_ in ob.sample:
        wl.f_destroy(wrappers[s_])
    sample = list(ob.sample)
    ob.close()
    return sample, dev


@pytest.mark.parametrize("B,channels,blocks", [(5, 2, 70), (64, 2, 45)])
def test_48k_wrapper_around_the_tick_pipeline_matches_in_order(bv, oracle, product, model_dir, B, channels, blocks):
    """BeatriceBatch_BindResidentIO48k: resident 48 kHz slots, the tick pipeline in between; block k's converted samples
    land in slot k mod n_slots (pipeline depth later).  Must equal the in-order device wrapper block for block, across a
    wrap of the slot ring and a drain in the middle."""
    from test_gpu_resident_io import Hip
    hip = Hip()
    x = np.stack([np.stack([wrapperlib.test_signal(480 * blocks, 48000, seed=300 + 7 * s + c) for c in range(channels)])
                  for s in range(B)]).astype(np.float32)                      # [B][ch][blocks*480]
    m = bv.Models(product, model_dir)

    def settings(batch):
        for s in range(B):
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % 3)
            batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, s % 2)
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)

    def change(batch, k):   # settings moving while blocks are in flight; a stream reset (drains the pipeline, restarts its wrapper)
        if k % 7 == 3:
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, (3 * k) % B, (k + 1) % 3)
            batch.a.BeatriceBatch_SetPitchShift(batch.h, (5 * k) % B, float(k % 5) - 2.0)
        if k == 31:
            assert batch.a.BeatriceBatch_ResetStream(batch.h, 1 % B) == 0

    ref_batch = bv.Batch(m, B)
    settings(ref_batch)
    want = []
    for k in range(blocks):
        change(ref_batch, k)
        want.append(ref_batch.convert48k(np.ascontiguousarray(x[:, :, 480 * k:480 * (k + 1)]), channels).copy())
    ref_batch.close()

    batch = bv.Batch(m, B)
    settings(batch)
    a, h = batch.a, batch.h
    stages = a.BeatriceBatch_TickStages(h)
    slots = stages + 5
    blk = B * channels * 480
    d_in, d_out = hip.malloc(slots * blk * 4), hip.malloc(slots * blk * 4)
    assert a.BeatriceBatch_BindResidentIO48k(h, d_in, d_out, channels, stages) == -1     # too few slots
    assert a.BeatriceBatch_BindResidentIO48k(h, d_in, d_out, channels, slots) == 0
    assert a.BeatriceBatch_ConvertBlocks48kDevice(h, d_in, d_out, channels) == -1        # bound: NULL, NULL only
    got = [None] * blocks
    k0 = 0
    for chunk in (slots, 9, 10 ** 9):         # each chunk: upload its blocks, run them, drain, read them back
        n = min(chunk, blocks - k0, slots)
        while n > 0:
            buf = np.zeros((slots, B, channels, 480), np.float32)
            for k in range(k0, k0 + n):
                buf[k % slots] = x[:, :, 480 * k:480 * (k + 1)]
            hip.h2d(d_in, buf)
            for k in range(k0, k0 + n):
                change(batch, k)
                assert a.BeatriceBatch_ConvertBlocks48kDevice(h, None, None, channels) == 0
            assert a.BeatriceBatch_Synchronize(h) == 0
            out = np.zeros((slots, B, channels, 480), np.float32)
            hip.d2h(out, d_out)
            for k in range(k0, k0 + n):
                got[k] = out[k % slots].copy()
            k0 += n
            n = min(chunk, blocks - k0, slots) if chunk > slots else 0
    assert k0 == blocks
    for k in range(blocks):
        assert np.array_equal(got[k], want[k]), "block %d differs" % k
    assert a.BeatriceBatch_BindResidentIO48k(h, None, None, 0, 0) == 0
    sample, dev = oracle_leg_48k(bv, oracle, model_dir, B, channels, x, blocks, settings, change, got)
    print("48k wrapper around the tick pipeline vs ORACLE host chain, streams %s, %d blocks: max-abs %g" % (sample, blocks, dev))
    assert dev <= 1e-4
    # in order again on the same streams (wrapper and model state carried over)
    y = batch.convert48k(np.ascontiguousarray(x[:, :, :480]), channels)
    assert np.isfinite(y).all()
    hip.free(d_in); hip.free(d_out)
    batch.close()
    m.close()


@pytest.mark.parametrize("H,B,channels,steps", [(2, 5, 2, 36), (2, 64, 2, 24), (2, 3, 1, 33), (4, 5, 2, 33), (4, 64, 2, 16), (4, 3, 1, 30)])
def test_48k_wrapper_around_the_tick_pipeline_two_blocks_per_step(bv, oracle, product, model_dir, H, B, channels, steps):
    """The same with a batch of two or four hops per step: a slot holds H consecutive 480-sample blocks per stream
    ([B][H][channels][480]), every call converts them all -- the resamplers and the FIFO run block after block inside the wrapper
    launch.  Must equal the in-order device wrapper (one block per call) block for block, and a sample of the streams the ORACLE
    host chain (this is the mode `bench.py --config 4` times); settings move between steps."""
    from test_gpu_resident_io import Hip
    hip = Hip()
    blocks = H * steps
    x = np.stack([np.stack([wrapperlib.test_signal(480 * blocks, 48000, seed=900 + 7 * s + c) for c in range(channels)])
                  for s in range(B)]).astype(np.float32)                      # [B][ch][blocks*480]
    m = bv.Models(product, model_dir)

    def settings(batch):
        for s in range(B):
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % 3)
            batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, s % 2)
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)

    def change(batch, k):   # before step k
        if k % 5 == 3:
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, (3 * k) % B, (k + 1) % 3)
            batch.a.BeatriceBatch_SetPitchShift(batch.h, (5 * k) % B, float(k % 5) - 2.0)

    ref_batch = bv.Batch(m, B)
    settings(ref_batch)
    want = []
    for k in range(blocks):
        if k % H == 0:
            change(ref_batch, k // H)
        want.append(ref_batch.convert48k(np.ascontiguousarray(x[:, :, 480 * k:480 * (k + 1)]), channels).copy())
    ref_batch.close()

    batch = bv.Batch(m, B, hops_per_step=H)
    settings(batch)
    a, h = batch.a, batch.h
    slots = a.BeatriceBatch_TickStages(h) + 4
    blk = B * H * channels * 480
    d_in, d_out = hip.malloc(slots * blk * 4), hip.malloc(slots * blk * 4)
    assert a.BeatriceBatch_BindResidentIO48k(h, d_in, d_out, channels, slots) == 0
    got = [None] * blocks
    k0 = 0
    while k0 < steps:
        n = min(slots if k0 == 0 else 7, steps - k0)
        buf = np.zeros((slots, B, H, channels, 480), np.float32)
        for k in range(k0, k0 + n):
            for hh in range(H):
                buf[k % slots, :, hh] = x[:, :, 480 * (H * k + hh):480 * (H * k + hh + 1)]
        hip.h2d(d_in, buf)
        for k in range(k0, k0 + n):
            change(batch, k)
            assert a.BeatriceBatch_ConvertBlocks48kDevice(h, None, None, channels) == 0
        assert a.BeatriceBatch_Synchronize(h) == 0
        out = np.zeros((slots, B, H, channels, 480), np.float32)
        hip.d2h(out, d_out)
        for k in range(k0, k0 + n):
            for hh in range(H):
                got[H * k + hh] = out[k % slots, :, hh].copy()
        k0 += n
    bad = [k for k in range(blocks) if not np.array_equal(got[k], want[k])]
    assert not bad, "blocks that differ: %s" % bad[:20]
    assert max(float(np.abs(w).max()) for w in want) > 1e-3
    assert a.BeatriceBatch_BindResidentIO48k(h, None, None, 0, 0) == 0
    batch.close()
    m.close()
    hip.free(d_in); hip.free(d_out)
    sample, dev = oracle_leg_48k(bv, oracle, model_dir, B, channels, x, blocks, settings,
                                 lambda script, kb: change(script, kb // H) if kb % H == 0 else None, got)
    print("48k wrapper around the tick pipeline, %d blocks per step, vs ORACLE host chain, streams %s, %d blocks: max-abs %g" % (H, sample, blocks, dev))
    assert dev <= 1e-4
