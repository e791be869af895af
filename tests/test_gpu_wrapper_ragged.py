"""per-stream wrapper clocks (a batch with 44.1 kHz and 48 kHz callers) and the shell's silent-block rule with
the any-rate wrapper.  BeatriceBatch_ConfigureWrapperRates / ProcessBlocksRagged give every stream of a batch its own host rate,
block length and FIFO phase (reference: every plugin instance owns its resampler pair, src/common/resample.h:401-438) and let a
stream sit a call out -- no block in this call, or a block the shell would not convert (src/vst/processor.cc:204-214).
Reference per stream: ProcessorProxy::ProcessChannels of the host layer on the ORACLE core at that stream's rate and block size,
which is CPU-pinned to the reference wrapper (test_host_layer.py, test_wrapper_oracle.py)."""
import ctypes as C

import numpy as np
import pytest

import wrapperlib
from test_host_proxy import K_MODEL, K_VOICE, K_VQ, Proxy

pytestmark = pytest.mark.gpu
_f32p = C.POINTER(C.c_float)


@pytest.mark.parametrize("channels,rule", [(1, True), (2, True), (1, False)])
def test_mixed_rate_batch_with_silent_blocks_matches_one_proxy_per_stream(bv, product, model_dir, channels, rule):
    rates = [44100.0, 48000.0, 96000.0, 44100.0, 32000.0, 48000.0]
    blocks = [441, 480, 1024, 300, 512, 64]          # host samples per call, per stream
    B, calls = len(rates), 44
    absent = {3: {5, 6, 20}, 5: {0, 1, 2, 30}}        # calls in which a stream hands in no block at all
    silent = {0: {4, 5, 17}, 1: {0, 9, 10, 11}, 2: {12}, 4: {3, 25, 26}} if rule else {}
    switch = {0: (4, 2), 1: (12, 0), 2: (7, 1), 4: (26, 2)}    # stream -> (before call, speaker); some right before silent blocks
    x = []
    for s in range(B):
        sig = np.stack([(0.6 if c else 1.0) * wrapperlib.test_signal(calls * blocks[s], int(rates[s]), seed=4100 + 7 * s + c) for c in range(channels)])
        for k in silent.get(s, ()):
            sig[:, k * blocks[s]:(k + 1) * blocks[s]] = 0.0
        x.append(sig.astype(np.float32))

    # ---- reference: one proxy on the oracle core per stream, its own rate and block size; absent calls simply do not happen
    want = [np.zeros_like(x[s]) for s in range(B)]
    for s in range(B):
        p = Proxy(rates[s])
        assert p.call("SetString", K_MODEL, (model_dir + "/model.toml").encode()) == 0
        p.call("SetInt", K_VOICE, s % 3)
        p.call("SetNumber", K_VQ, float(s % 3))
        n = blocks[s]
        for k in range(calls):
            if s in switch and switch[s][0] == k:
                p.call("SetInt", K_VOICE, switch[s][1])
            if k in absent.get(s, ()):
                continue
            sl = slice(k * n, (k + 1) * n)
            in0 = np.ascontiguousarray(x[s][0, sl])
            in1 = np.ascontiguousarray(x[s][1, sl]) if channels == 2 else None
            o0, o1 = np.zeros(n, np.float32), np.zeros(n, np.float32)
            if rule:
                flag = p.call("ProcessChannels", in0.ctypes.data_as(_f32p), in1.ctypes.data_as(_f32p) if in1 is not None else None,
                              o0.ctypes.data_as(_f32p), o1.ctypes.data_as(_f32p) if channels == 2 else None, n)
                assert flag == (1 if k in silent.get(s, ()) else 0)
            else:   # no rule: the core converts every block (mono down-mix by the caller, as the shell does)
                mono = in0 if channels == 1 else ((in0 + in1) * np.float32(0.5)).astype(np.float32)
                assert p.call("Process", mono.ctypes.data_as(_f32p), o0.ctypes.data_as(_f32p), n) == 0
                o1 = o0
            want[s][0, sl] = o0
            if channels == 2:
                want[s][1, sl] = o1
        p.close()

    # ---- product: one batch, every stream its own clocks
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    a, h = batch.a, batch.h
    for s in range(B):
        a.BeatriceBatch_SetTargetSpeaker(h, s, s % 3)
        a.BeatriceBatch_SetVQNumNeighbors(h, s, s % 3)
    assert a.BeatriceBatch_ProcessBlocksRagged(h, None, None, channels, (C.c_int * B)(*blocks), 0) == -1      # not configured
    assert a.BeatriceBatch_ConfigureWrapperRates(h, (C.c_double * B)(*rates)) == 0
    got = [np.zeros_like(x[s]) for s in range(B)]
    for k in range(calls):
        for s in range(B):
            if s in switch and switch[s][0] == k:
                a.BeatriceBatch_SetTargetSpeaker(h, s, switch[s][1])
        ns = [0 if k in absent.get(s, ()) else blocks[s] for s in range(B)]
        parts = [np.ascontiguousarray(x[s][:, k * blocks[s]:(k + 1) * blocks[s]]).reshape(-1) for s in range(B) if ns[s]]
        xin = np.concatenate(parts).astype(np.float32)
        out = np.zeros_like(xin)
        assert a.BeatriceBatch_ProcessBlocksRagged(h, bv.fptr(xin), bv.fptr(out), channels, (C.c_int * B)(*ns), 1 if rule else 0) == 0
        at = 0
        for s in range(B):
            if not ns[s]:
                continue
            cnt = channels * ns[s]
            got[s][:, k * blocks[s]:(k + 1) * blocks[s]] = out[at:at + cnt].reshape(channels, ns[s])
            at += cnt
    batch.close()
    m.close()
    for s in range(B):
        assert np.abs(want[s]).max() > 1e-3
        for k in silent.get(s, ()):
            assert not got[s][:, k * blocks[s]:(k + 1) * blocks[s]].any()
        d = np.abs(got[s] - want[s])
        assert np.array_equal(got[s], want[s]), "stream %d (%.0f Hz, %d-sample blocks): max-abs %g, first differing call %d" % (
            s, rates[s], blocks[s], d.max(), int(np.argmax(d.max(axis=0) > 0)) // blocks[s])
