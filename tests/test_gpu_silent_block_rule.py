"""a batch honours the shell's all-zero-block rule PER STREAM (reference src/vst/processor.cc:204-214: a
block whose down-mix is all zeros is not converted -- the core is not called, its state and its 10 ms FIFO stand still, the
output is that silence).  Reference per stream: the host layer's ProcessorProxy::ProcessChannels on the ORACLE core (the same
rule, CPU-tested in test_host_proxy.py), one proxy per stream; the batch: BeatriceBatch_ConvertBlocks48k with
BeatriceBatch_EnableSilentBlockRule, every stream with its own pattern of silent blocks, speaker switches landing right before
silent blocks (their key/value installs must wait), k-NN on."""
import ctypes as C
import os

import numpy as np
import pytest

import hostlib
import wrapperlib
from test_host_proxy import K_MODEL, K_VOICE, K_VQ, Proxy

pytestmark = pytest.mark.gpu
_f32p = C.POINTER(C.c_float)


@pytest.mark.parametrize("channels,device_flags", [(2, False), (1, True), (2, "ahead")])
def test_silent_blocks_per_stream_match_the_host_proxy(bv, product, model_dir, channels, device_flags):
    B, blocks, n = 5, 26, 480
    rng = np.random.default_rng(5)
    x = np.zeros((B, channels, blocks * n), np.float32)
    for s in range(B):
        for c in range(channels):
            x[s, c] = (0.7 if c else 1.0) * wrapperlib.test_signal(blocks * n, 48000, seed=3100 + 5 * s + c)
    silent = {0: {3, 4, 5, 11}, 1: {0, 1, 9, 20, 21}, 2: set(), 3: {6, 7, 8, 9, 10, 11, 12}, 4: {2, 13, 14, 24}}
    for s, ks in silent.items():
        for k in ks:
            x[s, :, k * n:(k + 1) * n] = 0.0
    if channels == 2:   # a block whose channels cancel is silent too: (L + R) * 0.5 == 0
        x[2, 1, 15 * n:16 * n] = -x[2, 0, 15 * n:16 * n]
        silent[2] = {15}
    switch = {0: (3, 2), 1: (8, 0), 3: (5, 1), 4: (13, 2)}   # stream -> (before block, speaker): some right before silent blocks

    # ---- reference: one ProcessorProxy on the oracle core per stream
    want = np.zeros((B, channels, blocks * n), np.float32)
    for s in range(B):
        p = Proxy(48000.0)                       # binds hostlib.HOST_ON_ORACLE
        assert p.call("SetString", K_MODEL, (model_dir + "/model.toml").encode()) == 0
        p.call("SetInt", K_VOICE, s % 3)
        p.call("SetNumber", K_VQ, float(s % 3))
        flags = []
        for k in range(blocks):
            if s in switch and switch[s][0] == k:
                p.call("SetInt", K_VOICE, switch[s][1])
            sl = slice(k * n, (k + 1) * n)
            in0 = np.ascontiguousarray(x[s, 0, sl])
            in1 = np.ascontiguousarray(x[s, 1, sl]) if channels == 2 else None
            o0, o1 = np.zeros(n, np.float32), np.zeros(n, np.float32)
            flags.append(p.call("ProcessChannels", in0.ctypes.data_as(_f32p), in1.ctypes.data_as(_f32p) if in1 is not None else None,
                                o0.ctypes.data_as(_f32p), o1.ctypes.data_as(_f32p) if channels == 2 else None, n))
            want[s, 0, sl] = o0
            if channels == 2:
                want[s, 1, sl] = o1
        assert flags == [1 if k in silent[s] else 0 for k in range(blocks)]
        p.close()

    # ---- product
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    a, h = batch.a, batch.h
    for s in range(B):
        a.BeatriceBatch_SetTargetSpeaker(h, s, s % 3)
        a.BeatriceBatch_SetVQNumNeighbors(h, s, s % 3)
    # (no FlushSpeaker: like the proxy's core after SetTargetSpeaker, the streams install their key/value blocks one per
    #  CONVERTED hop -- stream 1 starts with two silent blocks, so its installs begin at block 2)
    assert a.BeatriceBatch_SetSilentStreams(h, bytes(B)) == -1          # rule not enabled
    assert a.BeatriceBatch_EnableSilentBlockRule(h, 1) == 0
    got = np.zeros_like(x)
    hip = None
    ahead = device_flags == "ahead"   # every block enqueued before ONE synchronisation: the flags of 26 steps in flight at once
    if device_flags:
        from tick_driver import Hip
        hip = Hip()
        per = B * channels * n * 4
        d_in, d_out = hip.malloc(per * (blocks if ahead else 1)), hip.malloc(per * (blocks if ahead else 1))
        if ahead:
            for k in range(blocks):
                hip.h2d(C.c_void_p(d_in.value + k * per), np.ascontiguousarray(x[:, :, k * n:(k + 1) * n]))
    for k in range(blocks):
        for s in range(B):
            if s in switch and switch[s][0] == k:
                a.BeatriceBatch_SetTargetSpeaker(h, s, switch[s][1])
        sl = slice(k * n, (k + 1) * n)
        xin = np.ascontiguousarray(x[:, :, sl])
        out = np.zeros_like(xin)
        if device_flags:   # the caller names the silent streams of the block
            flags = bytes(1 if k in silent[s] else 0 for s in range(B))
            assert a.BeatriceBatch_SetSilentStreams(h, flags) == 0
            if ahead:
                assert a.BeatriceBatch_ConvertBlocks48kDevice(h, C.c_void_p(d_in.value + k * per), C.c_void_p(d_out.value + k * per), channels) == 0
                continue
            hip.h2d(d_in, xin)
            assert a.BeatriceBatch_ConvertBlocks48kDevice(h, d_in, d_out, channels) == 0
            assert a.BeatriceBatch_Synchronize(h) == 0
            hip.d2h(out, d_out)
        else:              # host buffers: the library applies the shell's own test
            assert a.BeatriceBatch_ConvertBlocks48k(h, bv.fptr(xin), bv.fptr(out), channels) == 0
        got[:, :, sl] = out
    if ahead:
        assert a.BeatriceBatch_Synchronize(h) == 0
        for k in range(blocks):
            out = np.zeros((B, channels, n), np.float32)
            hip.d2h(out, C.c_void_p(d_out.value + k * per))
            got[:, :, k * n:(k + 1) * n] = out
    # the rule is a mode of the in-order 48 kHz blocks: the throughput modes refuse while it is on, and it refuses inside them
    assert a.BeatriceBatch_EnableTickPipeline(h, 1) == -1 and a.BeatriceBatch_EnablePipelining(h, 2) == -1
    assert a.BeatriceBatch_EnableSilentBlockRule(h, 0) == 0
    batch.close()
    m.close()
    assert np.abs(want).max() > 1e-3
    for s in range(B):
        for k in silent[s]:
            assert not got[s, :, k * n:(k + 1) * n].any()
        assert np.array_equal(got[s], want[s]), "stream %d: max-abs %g, first differing block %d" % (
            s, np.abs(got[s] - want[s]).max(), int(np.argmax(np.abs(got[s] - want[s]).max(axis=0) > 0)) // n)
