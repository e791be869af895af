"""the shell's silent-block rule PER STREAM in the throughput mode.  In tick mode every stage of one launch works on
a different step, so a stream that sits a step out cannot be "put back" afterwards as the in-order chain does: instead the step
carries the step counter of each of its streams (-1: absent) through all 28 stages, every body addresses its rings with the
row's own counter and leaves absent rows alone (csrc/ring.h stepc::hopv, batch_tick.hip.h).  Reference per stream: one Stream1
on the ORACLE driven through the reference's per-hop protocol, whose hop is simply not called for the steps the stream sits out
(the shell does not call the core for a silent block, src/vst/processor.cc:204-214) -- state, pending key/value installs and
k-NN settings wait."""
import numpy as np
import pytest

from oracle_batch import OracleBatch
from tick_driver import run_tick

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,steps,H", [(7, 70, 1), (40, 45, 1), (7, 40, 2), (40, 34, 4), (7, 36, 4)])
def test_streams_that_sit_steps_out_in_tick_mode_match_the_oracle(bv, oracle, product, model_dir, B, steps, H):
    """H > 1 (round 6): a batch of several hops per step in plain tick mode -- a flagged stream sits a WHOLE step (its H hops) out; every body's
    ragged instance works on (stream, frame of the step) rows with the stream's own step counter, the linked GRU cells of a step's hops skip
    the rows of absent streams."""
    rng = np.random.default_rng(11 + B)
    tail = 6   # steps in order after the pipelined part: streams that sat steps out are brought back to one step counter when the batch leaves tick mode
    x = np.stack([bv.synth_audio(160 * H * (steps + tail), seed=7300 + s) for s in range(B)]).reshape(B, steps + tail, H * 160)
    # which steps each stream sits out: none for some, single steps, long runs, the very first steps, ...
    out = {s: set() for s in range(B)}
    for s in range(B):
        kind = s % 5
        if kind == 1:
            out[s] = {5, 6, 7, 30}
        elif kind == 2:
            out[s] = {0, 1, 2} | set(range(20, 36))
        elif kind == 3:
            out[s] = set(int(k) for k in rng.choice(steps, size=steps // 4, replace=False))
        elif kind == 4:
            out[s] = {steps - 1, steps - 2, 11}
    switch = {1: (5, 2), 2: (19, 0), 3: (9, 1), 4: (11, 2), 6: (33, 1)}   # stream -> (before step, speaker): some right before absent steps
    sample = list(range(B)) if B <= 8 else sorted(set([0, 1, 2, 3, 4, 16, 17, 18, 31, 33, 39]))

    # ---- reference: one oracle stream per sampled stream (tests/oracle_batch.py: the batch's defaults and setters on independent
    # Stream1 objects); a step a stream sits out is a hop that is never made
    ob = OracleBatch(bv, oracle, model_dir, B, sample=sample)
    for s in range(B):
        ob.a.BeatriceBatch_SetTargetSpeaker(None, s, s % 3)
        ob.a.BeatriceBatch_SetVQNumNeighbors(None, s, s % 3)
    ob.a.BeatriceBatch_FlushSpeaker(None, -1)
    def oracle_step(s, xs):   # the step's H hops of one stream, one after the other
        return np.concatenate([ob.st[s]["s1"].hop(xs[hh * 160:(hh + 1) * 160]) for hh in range(H)])

    want = np.zeros((steps, B, H * 240), np.float32)
    for k in range(steps):
        for s in ob.sample:
            if s in switch and switch[s][0] == k:
                ob.a.BeatriceBatch_SetTargetSpeaker(None, s, switch[s][1])
            if k not in out[s]:
                want[k, s] = oracle_step(s, x[s, k])
    sample = ob.sample
    want_tail = np.zeros((tail, B, H * 240), np.float32)
    for k in range(tail):
        for s in sample:
            want_tail[k, s] = oracle_step(s, x[s, steps + k])
    ob.close()

    # ---- product: tick mode, flags name the streams that sit the next step out
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B, hops_per_step=H)
    a, h = batch.a, batch.h
    for s in range(B):
        a.BeatriceBatch_SetTargetSpeaker(h, s, s % 3)
        a.BeatriceBatch_SetVQNumNeighbors(h, s, s % 3)
    a.BeatriceBatch_FlushSpeaker(h, -1)
    enabled = []

    def change(batch_, k):
        if not enabled:   # (tick mode is on by now: the rule is enabled inside it)
            assert a.BeatriceBatch_EnableSilentBlockRule(h, 1) == 0
            enabled.append(1)
        for s in range(B):
            if s in switch and switch[s][0] == k:
                a.BeatriceBatch_SetTargetSpeaker(h, s, switch[s][1])
        flags = bytes(1 if k in out[s] else 0 for s in range(B))
        if any(flags):
            assert a.BeatriceBatch_SetSilentStreams(h, flags) == 0

    got = run_tick(bv, batch, steps, lambda k: x[:, k], change=change, chunk=13)   # (every drain brings the streams back to one counter: ring_rotate_kernel)
    # ... and leaves tick mode: the in-order chain (ONE counter for all streams) takes the batch over
    assert a.BeatriceBatch_EnableSilentBlockRule(h, 0) == 0
    got_tail = np.stack([batch.convert(np.ascontiguousarray(x[:, steps + k])) for k in range(tail)])
    batch.close()
    m.close()
    assert np.abs(want).max() > 0.05
    bad = [(s, k) for s in sample for k in range(steps) if k not in out[s] and not np.array_equal(got[k, s], want[k, s])]
    assert not bad, "tick mode, (stream, step) that differ: %s" % bad[:12]
    bad_tail = [(s, k, float(np.abs(got_tail[k, s] - want_tail[k, s]).max())) for s in sample for k in range(tail) if not np.array_equal(got_tail[k, s], want_tail[k, s])]
    assert not bad_tail, "in order after tick mode, (stream, step, max-abs): %s" % bad_tail[:12]


@pytest.mark.parametrize("channels,H", [(1, 1), (2, 1), (2, 2), (1, 4), (2, 4)])
def test_silent_48k_blocks_per_stream_around_the_tick_pipeline(bv, product, model_dir, channels, H):
    """The same rule where the shell applies it: 48 kHz blocks, here resident on the device with the tick pipeline between the two
    halves of the wrapper (BeatriceBatch_BindResidentIO48k).  Reference per stream: ProcessorProxy::ProcessChannels of the host
    layer on the ORACLE core (the rule as the shell has it: the block is handed back, nothing moves).
    H = 2, 4 (round 6): a batch of several blocks per step -- a flagged stream sits a WHOLE step out, so the silent blocks here come in whole steps
    (to the reference each of them is a silent block like any other); speaker switches fall on step boundaries, where a batch applies them."""
    import ctypes as C
    import wrapperlib
    from test_host_proxy import K_MODEL, K_VOICE, K_VQ, Proxy
    from tick_driver import Hip
    _f32p = C.POINTER(C.c_float)
    B, n = 6, 480
    steps = 44 if H == 1 else (26 if H == 2 else 16)
    blocks = steps * H
    x = np.zeros((B, channels, blocks * n), np.float32)
    for s in range(B):
        for c in range(channels):
            x[s, c] = (0.7 if c else 1.0) * wrapperlib.test_signal(blocks * n, 48000, seed=5200 + 5 * s + c)
    silent_steps = ({0: {3, 4, 5, 11}, 1: {0, 1, 9, 20, 21}, 2: set(), 3: set(range(6, 19)), 4: {2, 13, 14, 43}, 5: {30}} if H == 1 else
                    {0: {3, 4, 11}, 1: {0, 1, 9}, 2: set(), 3: set(range(5, 12)), 4: {2, 13, 14, steps - 1}, 5: {7}})
    silent = {s: {st * H + hh for st in ks if st < steps for hh in range(H)} for s, ks in silent_steps.items()}
    for s, ks in silent.items():
        for k in ks:
            x[s, :, k * n:(k + 1) * n] = 0.0
    switch = {0: (3 * H, 2), 1: (8 * H, 0), 3: (5 * H, 1), 4: (13 * H, 2)}
    want = np.zeros_like(x)
    for s in range(B):
        p = Proxy(48000.0)
        assert p.call("SetString", K_MODEL, (model_dir + "/model.toml").encode()) == 0
        p.call("SetInt", K_VOICE, s % 3)
        p.call("SetNumber", K_VQ, float(s % 3))
        for k in range(blocks):
            if s in switch and switch[s][0] == k:
                p.call("SetInt", K_VOICE, switch[s][1])
            sl = slice(k * n, (k + 1) * n)
            in0 = np.ascontiguousarray(x[s, 0, sl])
            in1 = np.ascontiguousarray(x[s, 1, sl]) if channels == 2 else None
            o0, o1 = np.zeros(n, np.float32), np.zeros(n, np.float32)
            flag = p.call("ProcessChannels", in0.ctypes.data_as(_f32p), in1.ctypes.data_as(_f32p) if in1 is not None else None,
                          o0.ctypes.data_as(_f32p), o1.ctypes.data_as(_f32p) if channels == 2 else None, n)
            assert flag == (1 if k in silent[s] else 0)
            want[s, 0, sl] = o0
            if channels == 2:
                want[s, 1, sl] = o1
        p.close()

    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B, hops_per_step=H)
    a, h = batch.a, batch.h
    for s in range(B):
        a.BeatriceBatch_SetTargetSpeaker(h, s, s % 3)
        a.BeatriceBatch_SetVQNumNeighbors(h, s, s % 3)
    hip = Hip()
    slots = a.BeatriceBatch_TickStages(h) + 3
    d_in, d_out = hip.malloc(slots * B * H * channels * n * 4), hip.malloc(slots * B * H * channels * n * 4)
    assert a.BeatriceBatch_BindResidentIO48k(h, d_in, d_out, channels, slots) == 0
    assert a.BeatriceBatch_EnableSilentBlockRule(h, 1) == 0
    got = np.zeros_like(x)
    xs = x.reshape(B, channels, steps, H, n).transpose(2, 0, 3, 1, 4)   # [step][B][H][channels][n]: a step's resident slot
    gs = np.zeros_like(xs)
    k0 = 0
    while k0 < steps:
        cnt = min(11, steps - k0)
        buf = np.zeros((slots, B, H, channels, n), np.float32)
        for k in range(k0, k0 + cnt):
            buf[k % slots] = xs[k]
        hip.h2d(d_in, buf)
        for k in range(k0, k0 + cnt):
            for s in range(B):
                if s in switch and switch[s][0] == k * H:
                    a.BeatriceBatch_SetTargetSpeaker(h, s, switch[s][1])
            flags = bytes(1 if k * H in silent[s] else 0 for s in range(B))
            if any(flags):
                assert a.BeatriceBatch_SetSilentStreams(h, flags) == 0
            assert a.BeatriceBatch_ConvertBlocks48kDevice(h, None, None, channels) == 0
        assert a.BeatriceBatch_Synchronize(h) == 0
        out = np.zeros((slots, B, H, channels, n), np.float32)
        hip.d2h(out, d_out)
        for k in range(k0, k0 + cnt):
            gs[k] = out[k % slots]
        k0 += cnt
    got = np.ascontiguousarray(gs.transpose(1, 3, 0, 2, 4)).reshape(B, channels, blocks * n)
    batch.close()
    m.close()
    hip.free(d_in); hip.free(d_out)
    assert np.abs(want).max() > 1e-3
    for s in range(B):
        d = np.abs(got[s] - want[s])
        assert np.array_equal(got[s], want[s]), "stream %d: max-abs %g, first differing block %d" % (s, d.max(), int(np.argmax(d.max(axis=0) > 0)) // n)
