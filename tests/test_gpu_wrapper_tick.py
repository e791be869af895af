"""the reference host's whole wrapper (any host rate and block size, moving gains; processor_core_2.cc:24-48)
AROUND THE TICK PIPELINE (BeatriceBatch_BindResidentBlocks): resident host-rate blocks in, the model hops through the tick
launches, the output half of every call TickStages() - 1 calls later -- against the wrapper oracle (pinned to the reference's
gain.h / resample.h) around the oracle model, per stream, bit for bit, and against the same batch code in order."""
import numpy as np
import pytest

import wrapperlib
from tick_driver import Hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sr,block,channels,B,H", [(48000, 480, 2, 4, 1), (44100, 441, 1, 5, 1), (96000, 960, 1, 3, 1), (44100, 64, 1, 3, 1),
                                                   (32000, 1000, 2, 3, 1), (16000, 333, 1, 2, 1),
                                                   # several hops per step: a step enters the ticks when the FIFOs have fired H hops
                                                   (48000, 480, 2, 4, 2), (44100, 441, 1, 5, 2), (96000, 960, 1, 3, 4), (44100, 64, 1, 3, 2),
                                                   (32000, 1000, 2, 3, 4), (16000, 333, 1, 2, 2), (44100, 441, 1, 3, 4),
                                                   (44100, 441, 1, 3, -4),    # (H < 0: |H| hops per step over a long run, every ring of the binding wraps)
                                                   # (H + 100: the material ends with BeatriceBatch_FlushResidentBlocks instead of blocks of silence from the caller)
                                                   (44100, 441, 1, 3, 104), (48000, 480, 2, 4, 102), (44100, 64, 1, 3, 102), (32000, 1000, 2, 3, 104)])
def test_any_rate_wrapper_around_the_tick_pipeline(bv, oracle, product, model_dir, sr, block, channels, B, H):
    use_flush, H = H > 100, H % 100 if H > 100 else H
    long_run, H = H < 0, abs(H)
    n_blocks = max(40, int(0.45 * sr) // block)            # longer than the pipeline is deep, whatever the block size
    if long_run:
        n_blocks = 1200
    total = block * n_blocks
    x = np.zeros((B, channels, total), np.float32)
    for s in range(B):
        for c in range(channels):
            x[s, c] = (0.6 if c else 1.0) * wrapperlib.test_signal(total, sr, seed=2300 + 7 * s + c)
    ev_in = {s: [(block * 2, -6.0 - s), (block * 9, 3.0), (block * 20, -30.0 if s == 1 else 0.0)] for s in range(B)}
    ev_out = {s: [(block * 5, 4.0 + s), (block * 17, -12.0)] for s in range(B)}
    switch_at, switch_to = block * 11, {0: 2, 1: 0}        # speaker switches travel with the call they precede
    # Several hops per step: a setting applies to a STEP (the H hops that enter the ticks together), so the switch is made on a step
    # boundary -- at the first hop, a multiple of H, whose step fills in a later call than the step before it (the clocks are the
    # batch's: every stream fires its hops in the same calls).  fired[k] = the call in which hop k fired, from a dry run of the
    # wrapper oracle with a hop that does nothing.
    fired = []
    if H > 1:
        blk = {"k": 0}
        wo = wrapperlib.oracle_wrapper()
        cb0 = wrapperlib.HOP_FN(lambda i, o, u: fired.append(blk["k"]))
        p0 = wo.f_create(float(sr), cb0, None)
        z, zo = np.zeros(block, np.float32), np.zeros(block, np.float32)
        for k in range(n_blocks):
            blk["k"] = k
            wo.f_process(p0, z.ctypes.data_as(wrapperlib._f32p), zo.ctypes.data_as(wrapperlib._f32p), block)
        wo.f_destroy(p0)
        first = next(i for i, c in enumerate(fired) if c >= 11)
        switch_hop = next(k for k in range((first + H - 1) // H * H, len(fired) - H, H) if fired[k - 1] < fired[k + H - 1])
        switch_call = fired[switch_hop + H - 1]            # the call that fills (and feeds) the step the switch belongs to

    mo = bv.Models(oracle, model_dir)
    want = np.zeros((B, total), np.float32)
    for s in range(B):
        st = bv.Stream1(mo, speaker=s % 3, vq_k=s % 2)
        state = {"fed": 0}

        def hop(in160, out240, _u, st=st, s=s, state=state):
            if H > 1 and state["fed"] == switch_hop and s in switch_to:
                st.set_target_speaker(switch_to[s])
            state["fed"] += 1
            np.ctypeslib.as_array(out240, (240,))[:] = st.hop(np.ctypeslib.as_array(in160, (160,)).copy())

        # the wrapper oracle calls `hop` when its FIFO fills; a speaker switch set before block k reaches the model with the
        # first hop fired in or after block k: run the chain block by block so that the switch lands between the right calls
        wo = wrapperlib.oracle_wrapper()
        cb = wrapperlib.HOP_FN(hop)
        p = wo.f_create(float(sr), cb, None)
        mono = x[s, 0] if channels == 1 else ((x[s, 0] + x[s, 1]) * np.float32(0.5)).astype(np.float32)
        mono = np.ascontiguousarray(mono)
        out = np.zeros_like(mono)
        e_in, e_out = list(ev_in[s]), list(ev_out[s])
        for k in range(n_blocks):
            pos = k * block
            while e_in and e_in[0][0] <= pos:
                wo.f_in_gain(p, e_in.pop(0)[1])
            while e_out and e_out[0][0] <= pos:
                wo.f_out_gain(p, e_out.pop(0)[1])
            if H == 1 and pos == switch_at and s in switch_to:
                st.set_target_speaker(switch_to[s])
            wo.f_process(p, mono[pos:pos + block].ctypes.data_as(wrapperlib._f32p), out[pos:pos + block].ctypes.data_as(wrapperlib._f32p), block)
        wo.f_destroy(p)
        want[s] = out
        st.close()
    mo.close()

    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B, hops_per_step=H)
    a, h = batch.a, batch.h
    for s in range(B):
        a.BeatriceBatch_SetTargetSpeaker(h, s, s % 3)
        a.BeatriceBatch_SetVQNumNeighbors(h, s, s % 2)
    a.BeatriceBatch_FlushSpeaker(h, -1)
    assert a.BeatriceBatch_ConfigureWrapper(h, float(sr)) == 0
    hip = Hip()
    stages = a.BeatriceBatch_TickStages(h)
    want_delay = a.BeatriceBatch_ResidentBlocksDelayFor(h, block)
    assert want_delay == stages - 1 if H == 1 else want_delay > 0          # (several hops per step: the calls that bring (H - 1) + (stages - 1) H hops)
    slots = 3 * stages + (want_delay - (stages - 1))         # chunks longer than the delay: most output halves run while the pipeline is full
    d_in, d_out = hip.malloc(slots * B * channels * block * 4), hip.malloc(slots * B * channels * block * 4)
    assert a.BeatriceBatch_BindResidentBlocks(h, d_in, d_out, channels, block, want_delay + 1) == -1      # too few slots
    assert a.BeatriceBatch_BindResidentBlocks(h, d_in, d_out, channels, block, slots) == 0
    delay = a.BeatriceBatch_ResidentBlocksDelay(h)
    assert delay == want_delay
    assert a.BeatriceBatch_EnableSilentBlockRule(h, 0) == -1 and a.BeatriceBatch_EnableSilentBlockRule(h, 1) == -1   # not under a binding
    # several hops per step: the last calls end on a step that is still filling and stay owed at a drained point; blocks of silence
    # behind the material bring them out (the wrapper oracle is not asked about those)
    n_real, n_blocks = n_blocks, n_blocks + (0 if H == 1 or use_flush else delay - (stages - 1) + 1)
    x = np.concatenate([x, np.zeros((B, channels, block * (n_blocks - n_real)), np.float32)], axis=2)
    got = np.zeros_like(x)
    have = 0                                                 # calls whose output block has been collected
    e_in, e_out = {s: list(v) for s, v in ev_in.items()}, {s: list(v) for s, v in ev_out.items()}
    chunk = slots - delay - 1 if slots - delay - 1 >= 1 else 1
    k0 = 0
    buf_in = np.zeros((slots, B, channels, block), np.float32)
    while k0 < n_blocks:
        nk = min(chunk, n_blocks - k0)
        for k in range(k0, k0 + nk):
            buf_in[k % slots] = x[:, :, k * block:(k + 1) * block]
        hip.h2d(d_in, buf_in)
        for k in range(k0, k0 + nk):
            pos = k * block
            for s in range(B):
                while e_in[s] and e_in[s][0][0] <= pos:
                    a.BeatriceBatch_SetInputGain(h, s, e_in[s].pop(0)[1])
                while e_out[s] and e_out[s][0][0] <= pos:
                    a.BeatriceBatch_SetOutputGain(h, s, e_out[s].pop(0)[1])
                if (pos == switch_at if H == 1 else k == switch_call) and s in switch_to:
                    a.BeatriceBatch_SetTargetSpeaker(h, s, switch_to[s])
            assert a.BeatriceBatch_ProcessBlocksDevice(h, None, None, channels, block) == 0
        if use_flush and k0 + nk == n_blocks:                    # the end of the material: nothing stays owed, no silence from the caller
            assert a.BeatriceBatch_FlushResidentBlocks(h) == 0
            assert a.BeatriceBatch_ResidentBlocksOwed(h) == 0
        assert a.BeatriceBatch_Synchronize(h) == 0               # drains the pipeline and runs the output halves still owed
        owed = a.BeatriceBatch_ResidentBlocksOwed(h)
        assert owed == 0 if H == 1 else 0 <= owed <= delay - (stages - 1) + 1
        out = np.zeros((slots, B, channels, block), np.float32)
        hip.d2h(out, d_out)
        for k in range(have, k0 + nk - owed):
            got[:, :, k * block:(k + 1) * block] = out[k % slots]
        have = k0 + nk - owed
        k0 += nk
    assert have >= n_real
    got, x = got[:, :, :n_real * block], x[:, :, :n_real * block]
    if use_flush:   # the binding has started over: call 0 reads slot 0 again, the first block out is the FIFO's silence, then sound; a second flush ends it
        buf_in[:min(slots, n_real)] = x[:, :, :min(slots, n_real) * block].reshape(B, channels, -1, block).transpose(2, 0, 1, 3)
        hip.h2d(d_in, buf_in)
        n_more = min(slots - 1, 3 * H + 5)
        for k in range(n_more):
            assert a.BeatriceBatch_ProcessBlocksDevice(h, None, None, channels, block) == 0
        assert a.BeatriceBatch_FlushResidentBlocks(h) == 0 and a.BeatriceBatch_ResidentBlocksOwed(h) == 0
        again = np.zeros((slots, B, channels, block), np.float32)
        hip.d2h(again, d_out)
        assert np.isfinite(again[:n_more]).all() and np.abs(again[1:n_more]).max() > 1e-4
        if block * 48000 <= 480 * sr:
            assert np.abs(again[0]).max() == 0.0               # (at most 10 ms: nothing but the restarted FIFO's zeros)
    assert a.BeatriceBatch_BindResidentBlocks(h, None, None, 0, 0, 0) == 0
    batch.close()
    m.close()
    hip.free(d_in)
    hip.free(d_out)
    dev = float(np.abs(got[:, 0] - want).max())
    print("wrapper around the tick pipeline sr=%d block=%d ch=%d: max-abs %g" % (sr, block, channels, dev))
    assert np.abs(want).max() > 1e-3
    if channels == 2:
        assert np.array_equal(got[:, 0], got[:, 1])
    assert dev == 0.0
