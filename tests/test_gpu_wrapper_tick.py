"""the reference host's whole wrapper (any host rate and block size, moving gains; processor_core_2.cc:24-48)
AROUND THE TICK PIPELINE (BeatriceBatch_BindResidentBlocks): resident host-rate blocks in, the model hops through the tick
launches, the output half of every call TickStages() - 1 calls later -- against the wrapper oracle (pinned to the reference's
gain.h / resample.h) around the oracle model, per stream, bit for bit, and against the same batch code in order."""
import numpy as np
import pytest

import wrapperlib
from tick_driver import Hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sr,block,channels,B", [(48000, 480, 2, 4), (44100, 441, 1, 5), (96000, 960, 1, 3), (44100, 64, 1, 3),
                                                 (32000, 1000, 2, 3), (16000, 333, 1, 2)])
def test_any_rate_wrapper_around_the_tick_pipeline(bv, oracle, product, model_dir, sr, block, channels, B):
    n_blocks = max(40, int(0.45 * sr) // block)            # longer than the pipeline is deep, whatever the block size
    total = block * n_blocks
    x = np.zeros((B, channels, total), np.float32)
    for s in range(B):
        for c in range(channels):
            x[s, c] = (0.6 if c else 1.0) * wrapperlib.test_signal(total, sr, seed=2300 + 7 * s + c)
    ev_in = {s: [(block * 2, -6.0 - s), (block * 9, 3.0), (block * 20, -30.0 if s == 1 else 0.0)] for s in range(B)}
    ev_out = {s: [(block * 5, 4.0 + s), (block * 17, -12.0)] for s in range(B)}
    switch_at, switch_to = block * 11, {0: 2, 1: 0}        # speaker switches travel with the call they precede

    mo = bv.Models(oracle, model_dir)
    want = np.zeros((B, total), np.float32)
    for s in range(B):
        st = bv.Stream1(mo, speaker=s % 3, vq_k=s % 2)
        state = {"fed": 0}

        def hop(in160, out240, _u, st=st, s=s, state=state):
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
            if pos == switch_at and s in switch_to:
                st.set_target_speaker(switch_to[s])
            wo.f_process(p, mono[pos:pos + block].ctypes.data_as(wrapperlib._f32p), out[pos:pos + block].ctypes.data_as(wrapperlib._f32p), block)
        wo.f_destroy(p)
        want[s] = out
        st.close()
    mo.close()

    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    a, h = batch.a, batch.h
    for s in range(B):
        a.BeatriceBatch_SetTargetSpeaker(h, s, s % 3)
        a.BeatriceBatch_SetVQNumNeighbors(h, s, s % 2)
    a.BeatriceBatch_FlushSpeaker(h, -1)
    assert a.BeatriceBatch_ConfigureWrapper(h, float(sr)) == 0
    hip = Hip()
    stages = a.BeatriceBatch_TickStages(h)
    slots = 3 * stages                                       # chunks longer than the delay: most output halves run while the pipeline is full
    d_in, d_out = hip.malloc(slots * B * channels * block * 4), hip.malloc(slots * B * channels * block * 4)
    assert a.BeatriceBatch_BindResidentBlocks(h, d_in, d_out, channels, block, stages) == -1      # too few slots
    assert a.BeatriceBatch_BindResidentBlocks(h, d_in, d_out, channels, block, slots) == 0
    delay = a.BeatriceBatch_ResidentBlocksDelay(h)
    assert delay == stages - 1
    got = np.zeros_like(x)
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
                if pos == switch_at and s in switch_to:
                    a.BeatriceBatch_SetTargetSpeaker(h, s, switch_to[s])
            assert a.BeatriceBatch_ProcessBlocksDevice(h, None, None, channels, block) == 0
        assert a.BeatriceBatch_Synchronize(h) == 0               # drains the pipeline and runs the output halves still owed
        out = np.zeros((slots, B, channels, block), np.float32)
        hip.d2h(out, d_out)
        for k in range(k0, k0 + nk):
            got[:, :, k * block:(k + 1) * block] = out[k % slots]
        k0 += nk
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
