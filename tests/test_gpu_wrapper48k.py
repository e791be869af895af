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
