"""The reference host's whole wrapper on the device at any host rate (BeatriceBatch_ConfigureWrapper /
BeatriceBatch_ProcessBlocks): moving input and output gains, rational resampling both ways, the 480-sample FIFO --
against the wrapper oracle (pinned bit for bit to the reference's gain.h / resample.h, tests/test_wrapper_oracle.py)
around the oracle model, per stream.  SURVEY.md section 8 rows a2-a6 on the device."""
import numpy as np
import pytest

import wrapperlib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sr,block,channels,B", [(48000, 480, 2, 5), (44100, 441, 1, 4), (44100, 512, 2, 3), (96000, 960, 1, 3),
                                                 (24000, 240, 2, 4), (16000, 333, 1, 3), (88200, 64, 1, 2), (32000, 1000, 1, 2)])
def test_device_wrapper_any_rate_with_gains(bv, oracle, product, model_dir, sr, block, channels, B):
    n_blocks = max(6, int(0.16 * sr) // block)
    total = block * n_blocks
    x = np.zeros((B, channels, total), np.float32)
    for s in range(B):
        for c in range(channels):
            x[s, c] = (0.6 if c else 1.0) * wrapperlib.test_signal(total, sr, seed=1300 + 7 * s + c)
    # gain events (sample index, dB); applied at the first block that starts at or after the index
    ev_in = {s: [(block * 1, -6.0 - s), (block * 3, 3.0), (block * 4, -40.0 if s == 1 else 0.0)] for s in range(B)}
    ev_out = {s: [(block * 2, 4.0 + s), (block * 4, -12.0)] for s in range(B)}

    mo = bv.Models(oracle, model_dir)
    want = np.zeros((B, total), np.float32)
    for s in range(B):
        st = bv.Stream1(mo, speaker=s % 3, vq_k=s % 2)

        def hop(in160, out240, _u, st=st):
            np.ctypeslib.as_array(out240, (240,))[:] = st.hop(np.ctypeslib.as_array(in160, (160,)).copy())

        mono = x[s, 0] if channels == 1 else ((x[s, 0] + x[s, 1]) * np.float32(0.5)).astype(np.float32)
        want[s] = wrapperlib.oracle_wrapper().run_chain(sr, mono, block, hop=hop, in_gain_events=ev_in[s], out_gain_events=ev_out[s])
        st.close()
    mo.close()

    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    a, h = batch.a, batch.h
    for s in range(B):
        a.BeatriceBatch_SetTargetSpeaker(h, s, s % 3)
        a.BeatriceBatch_SetVQNumNeighbors(h, s, s % 2)
    a.BeatriceBatch_FlushSpeaker(h, -1)
    assert a.BeatriceBatch_ProcessBlocks(h, bv.fptr(x[:, :, :block].copy()), bv.fptr(np.zeros((B, channels, block), np.float32)), channels, block) == -1  # not configured
    assert a.BeatriceBatch_ConfigureWrapper(h, float(sr)) == 0
    assert a.BeatriceBatch_MaxWrapperBlock(h) >= block
    got = np.zeros_like(x)
    for k in range(n_blocks):
        pos = k * block
        for s in range(B):
            while ev_in[s] and ev_in[s][0][0] <= pos:
                a.BeatriceBatch_SetInputGain(h, s, ev_in[s].pop(0)[1])
            while ev_out[s] and ev_out[s][0][0] <= pos:
                a.BeatriceBatch_SetOutputGain(h, s, ev_out[s].pop(0)[1])
        xin = np.ascontiguousarray(x[:, :, pos:pos + block])
        out = np.zeros_like(xin)
        assert a.BeatriceBatch_ProcessBlocks(h, bv.fptr(xin), bv.fptr(out), channels, block) == 0
        got[:, :, pos:pos + block] = out
    batch.close()
    m.close()
    dev = float(np.abs(got[:, 0] - want).max())
    print("device wrapper sr=%d block=%d ch=%d: max-abs %g %s" % (sr, block, channels, dev, "bit-identical" if np.array_equal(got[:, 0], want) else ""))
    assert np.abs(want).max() > 1e-3
    if channels == 2:
        assert np.array_equal(got[:, 0], got[:, 1])
    assert dev <= 1e-6


def test_device_wrapper_varying_block_sizes(bv, oracle, product, model_dir):
    """A DAW may hand over a different block size on every call."""
    sr, B = 44100, 2
    sizes = [441, 100, 1, 512, 37, 882, 441, 441, 300, 64, 1024, 441]
    total = sum(sizes)
    x = np.stack([wrapperlib.test_signal(total, sr, seed=1700 + s) for s in range(B)])
    mo = bv.Models(oracle, model_dir)
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    assert batch.a.BeatriceBatch_ConfigureWrapper(batch.h, float(sr)) == 0
    want = np.zeros_like(x)
    w = wrapperlib.oracle_wrapper()
    got = np.zeros_like(x)
    for s in range(B):
        st = bv.Stream1(mo, speaker=0, vq_k=0)

        def hop(in160, out240, _u, st=st):
            np.ctypeslib.as_array(out240, (240,))[:] = st.hop(np.ctypeslib.as_array(in160, (160,)).copy())

        cb = wrapperlib.HOP_FN(hop)
        p = w.f_create(float(sr), cb, None)
        pos = 0
        for n in sizes:
            seg = np.ascontiguousarray(x[s, pos:pos + n])
            out = np.zeros(n, np.float32)
            assert w.f_process(p, seg.ctypes.data_as(wrapperlib._f32p), out.ctypes.data_as(wrapperlib._f32p), n) == 0
            want[s, pos:pos + n] = out
            pos += n
        w.f_destroy(p)
        st.close()
    pos = 0
    for n in sizes:
        xin = np.ascontiguousarray(x[:, None, pos:pos + n])
        out = np.zeros_like(xin)
        assert batch.a.BeatriceBatch_ProcessBlocks(batch.h, bv.fptr(xin), bv.fptr(out), 1, n) == 0
        got[:, pos:pos + n] = out[:, 0]
        pos += n
    batch.close(); m.close(); mo.close()
    assert np.abs(want).max() > 1e-3
    assert np.array_equal(got, want), "max-abs %g" % np.abs(got - want).max()
