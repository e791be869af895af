"""Parity at the shapes BASELINE.json names per GPU: configs[3] (256 streams on a 64-speaker table, every stream
rotating through the speakers, k-NN 4 -- the regime where the attention tile lists hold 64 key/value slots with
four rows each) and configs[4] (64 stereo 48 kHz streams through the device wrapper).  Every stream is compared with
an independent oracle stream driven through the reference's per-hop protocol."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import wrapperlib

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


@pytest.fixture(scope="module")
def model_dir64(tmp_path_factory):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_model
    d = str(tmp_path_factory.mktemp("model64"))
    make_model.make_model(d, n_speakers=64)
    return d


def _workers():
    return max(1, min(32, os.cpu_count() or 1))


def test_config3_256_streams_64_rotating_speakers(bv, oracle, product, model_dir64):
    B, S, hops = 256, 64, 13
    audio = np.stack([bv.synth_audio(160 * hops, seed=4000 + s) for s in range(B)])
    start = [s % S for s in range(B)]
    switch_at = [2 + (s % 7) for s in range(B)]          # staggered: hops 2..8, so block installs overlap across streams

    mo = bv.Models(oracle, model_dir64)

    def one_stream(s):  # the oracle's streams are independent objects; ctypes releases the GIL inside the library
        st = bv.Stream1(mo, speaker=start[s], vq_k=4)
        out = np.zeros((hops, bv.OUT_HOP), np.float32)
        for h in range(hops):
            if h == switch_at[s]:
                st.set_target_speaker((start[s] + 1) % S)
            out[h] = st.hop(audio[s, h * 160:(h + 1) * 160])
        st.close()
        return out

    with ThreadPoolExecutor(_workers()) as pool:
        ref = np.stack(list(pool.map(one_stream, range(B))), axis=1)   # [hops][B][240]
    mo.close()

    m = bv.Models(product, model_dir64)
    batch = bv.Batch(m, B)
    a, hnd = batch.a, batch.h
    for s in range(B):
        a.BeatriceBatch_SetTargetSpeaker(hnd, s, start[s])
    a.BeatriceBatch_FlushSpeaker(hnd, -1)
    a.BeatriceBatch_SetVQNumNeighbors(hnd, -1, 4)
    got = np.zeros_like(ref)
    for h in range(hops):
        for s in range(B):
            if h == switch_at[s]:
                a.BeatriceBatch_SetTargetSpeaker(hnd, s, (start[s] + 1) % S)
        got[h] = batch.convert(audio[:, h * 160:(h + 1) * 160])
    batch.close()
    m.close()
    dev = float(np.abs(ref - got).max())
    print("configs[3] shape (256 streams, 64 speakers rotating, k-NN 4): max-abs %g %s"
          % (dev, "bit-identical" if np.array_equal(ref, got) else ""))
    assert np.abs(got).max() > 0.05
    assert dev <= TOL


@pytest.mark.parametrize("H", [4, 2])
def test_config3_shape_through_the_tick_pipeline(bv, oracle, product, model_dir64, H):
    """The form bench.py --config 3 times: 256 streams on the 64-speaker table in tick mode with H hops per stage per launch
    (4: the bench's default -- a speaker's 4 streams x 4 hops are exactly one full 16-row attention tile; 2: half a tile; while
    its streams switch, the rest are quads), every stream rotating, k-NN 4 -- every stream against an independent oracle stream."""
    from tick_driver import run_tick
    B, S, steps = 256, 64, 9
    hops = H * steps
    audio = np.stack([bv.synth_audio(160 * hops, seed=4500 + s) for s in range(B)])
    start = [s % S for s in range(B)]
    switch_at = [1 + (s % 5) for s in range(B)]          # step index: the switch precedes the step's first hop

    mo = bv.Models(oracle, model_dir64)

    def one_stream(s):
        st = bv.Stream1(mo, speaker=start[s], vq_k=4)
        out = np.zeros((hops, bv.OUT_HOP), np.float32)
        for h in range(hops):
            if h == H * switch_at[s]:
                st.set_target_speaker((start[s] + 1) % S)
            out[h] = st.hop(audio[s, h * 160:(h + 1) * 160])
        st.close()
        return out

    with ThreadPoolExecutor(_workers()) as pool:
        ref = np.stack(list(pool.map(one_stream, range(B))), axis=0)   # [B][hops][240]
    mo.close()
    ref = ref.reshape(B, steps, H * 240).transpose(1, 0, 2)            # [steps][B][480]

    m = bv.Models(bv.bind_batch(product), model_dir64)
    batch = bv.Batch(m, B, hops_per_step=H)
    a, hnd = batch.a, batch.h
    for s in range(B):
        a.BeatriceBatch_SetTargetSpeaker(hnd, s, start[s])
    a.BeatriceBatch_FlushSpeaker(hnd, -1)
    a.BeatriceBatch_SetVQNumNeighbors(hnd, -1, 4)

    def change(bt, k):
        for s in range(B):
            if k == switch_at[s]:
                bt.a.BeatriceBatch_SetTargetSpeaker(bt.h, s, (start[s] + 1) % S)

    got = run_tick(bv, batch, steps, lambda k: audio[:, k * H * 160:(k + 1) * H * 160], change)
    batch.close()
    m.close()
    dev = float(np.abs(ref - got).max())
    print("configs[3] shape in tick mode, %d hops per step: max-abs %g %s" % (H, dev, "bit-identical" if np.array_equal(ref, got) else ""))
    assert np.abs(got).max() > 0.05
    assert dev <= TOL


def test_config4_64_stereo_streams_48k(bv, oracle, product, model_dir):
    B, blocks = 64, 12
    x = np.zeros((B, 2, 480 * blocks), np.float32)
    for s in range(B):
        x[s, 0] = wrapperlib.test_signal(480 * blocks, 48000, seed=5000 + 2 * s)
        x[s, 1] = 0.6 * wrapperlib.test_signal(480 * blocks, 48000, seed=5001 + 2 * s)
    mo = bv.Models(oracle, model_dir)
    want = np.zeros((B, 480 * blocks), np.float32)
    for s in range(B):   # (the wrapper oracle calls back into Python per hop: one stream after the other)
        st = bv.Stream1(mo, speaker=s % 3, vq_k=0)

        def hop(in160, out240, _u, st=st):
            o = st.hop(np.ctypeslib.as_array(in160, (160,)).copy())
            np.ctypeslib.as_array(out240, (240,))[:] = o

        mono = ((x[s, 0] + x[s, 1]) * np.float32(0.5)).astype(np.float32)
        want[s] = wrapperlib.oracle_wrapper().run_chain(48000, mono, 480, hop=hop)
        st.close()
    mo.close()
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    for s in range(B):
        batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % 3)
    batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)
    got = np.zeros_like(x)
    for k in range(blocks):
        got[:, :, 480 * k:480 * (k + 1)] = batch.convert48k(x[:, :, 480 * k:480 * (k + 1)], 2)
    batch.close()
    m.close()
    dev = float(np.abs(got[:, 0] - want).max())
    print("configs[4] share (64 stereo streams @48 kHz): max-abs %g" % dev)
    assert np.abs(want).max() > 1e-3
    assert np.array_equal(got[:, 0], got[:, 1])
    assert dev <= TOL


def test_device_blob_sharing_roundtrip(bv, product, model_dir):
    """The multi-GPU load path on one GPU: a second set of model objects receives the first set's PACKED device blobs
    through torch tensors that alias the library's memory (what the RCCL broadcast writes into), is marked ready, and
    must then produce the same samples as the file-loaded set."""
    import ctypes as C
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("beatrice_shard", os.path.join(REPO, "beatrice-vst_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    bv.bind_batch(product)
    src = bv.Models(product, model_dir)
    dst_objs = {"phone": product.CreatePhoneExtractor(), "pitch": product.CreatePitchEstimator(),
                "wave": product.CreateWaveformGenerator(), "embed": product.CreateEmbeddingSetter()}
    src_objs = {"phone": src.phone, "pitch": src.pitch, "wave": src.wave, "embed": src.embed}
    for name, kind, _reader, _f in shard.KINDS:
        p0, n0, p1, n1 = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        assert product.BeatriceHip_ModelBlob(kind, dst_objs[name], 0, C.byref(p1), C.byref(n1)) == -1   # nothing to share yet
        assert product.BeatriceHip_ModelBlob(kind, src_objs[name], 0, C.byref(p0), C.byref(n0)) == 0
        assert product.BeatriceHip_ModelBlob(kind, dst_objs[name], 1, C.byref(p1), C.byref(n1)) == 0
        assert n0.value == n1.value and p0.value != p1.value
        shard.device_bytes(torch, p1.value, n1.value).copy_(shard.device_bytes(torch, p0.value, n0.value))
        torch.cuda.synchronize()
        assert product.BeatriceHip_ModelBlobReady(kind, dst_objs[name]) == 0

    class Twin:
        pass
    twin = Twin()
    twin.abi, twin.tables = product, src.tables
    twin.phone, twin.pitch, twin.wave, twin.embed = (dst_objs[k] for k in ("phone", "pitch", "wave", "embed"))
    x = bv.synth_audio(160 * 6, seed=77)
    outs = []
    for models in (src, twin):
        batch = bv.Batch(models, 2)
        batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, 1, 2)
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)
        outs.append(np.stack([batch.convert(np.stack([x[h * 160:(h + 1) * 160]] * 2)) for h in range(6)]))
        batch.close()
    # device tables filled device to device: batch B's raw tables copied from batch A's, then projected
    ba, bb = bv.Batch(src, 2), bv.Batch(twin, 2, max_speakers=src.tables.n_speakers + 1, upload_tables=False)
    pa, na, pb, nb = (C.c_void_p * 4)(), (C.c_size_t * 4)(), (C.c_void_p * 4)(), (C.c_size_t * 4)()
    assert product.BeatriceBatch_SpeakerTablesDevice(ba.h, pa, na) == 0 and product.BeatriceBatch_SpeakerTablesDevice(bb.h, pb, nb) == 0
    for i in range(4):
        assert na[i] == nb[i]
        shard.device_bytes(torch, pb[i], nb[i]).copy_(shard.device_bytes(torch, pa[i], na[i]))
    torch.cuda.synchronize()
    assert product.BeatriceBatch_ProjectSpeakerTables(bb.h, src.tables.n_speakers + 1) == 0
    bb.apply_defaults()
    for b in (ba, bb):
        b.a.BeatriceBatch_SetTargetSpeaker(b.h, 1, 2)
        b.a.BeatriceBatch_FlushSpeaker(b.h, -1)
    o2 = [np.stack([b.convert(np.stack([x[h * 160:(h + 1) * 160]] * 2)) for h in range(6)]) for b in (ba, bb)]
    ba.close()
    bb.close()
    for o in dst_objs:
        pass
    product.DestroyPhoneExtractor(dst_objs["phone"])
    product.DestroyPitchEstimator(dst_objs["pitch"])
    product.DestroyWaveformGenerator(dst_objs["wave"])
    product.DestroyEmbeddingSetter(dst_objs["embed"])
    src.close()
    assert np.abs(outs[0]).max() > 0.01
    assert np.array_equal(outs[0], outs[1]), "objects filled device-to-device differ from file-loaded ones"
    assert np.array_equal(o2[0], o2[1]) and np.array_equal(o2[0], outs[0])


def test_set_target_speakers_equals_one_call_per_stream(bv, product, model_dir):
    """BeatriceBatch_SetTargetSpeakers(b, n, streams, speakers) = n calls of BeatriceBatch_SetTargetSpeaker (key/value blocks follow one
    per hop either way, processor_core_2.cc:431-466); an out-of-range pair refuses the whole call and changes nothing."""
    import ctypes as C
    B, hops = 12, 14
    x = np.stack([bv.synth_audio(160 * hops, seed=6100 + s) for s in range(B)]).reshape(B, hops, 160)
    moves = {3: [(0, 2), (5, 1), (11, 0)], 4: [(5, 2)], 9: [(s, (s + 1) % 3) for s in range(B)]}
    m = bv.Models(bv.bind_batch(product), model_dir)
    outs = []
    for batched in (False, True):
        batch = bv.Batch(m, B)
        a, h = batch.a, batch.h
        for s in range(B):
            a.BeatriceBatch_SetTargetSpeaker(h, s, s % 3)
            a.BeatriceBatch_SetVQNumNeighbors(h, s, 1 + s % 2)
        a.BeatriceBatch_FlushSpeaker(h, -1)
        got = []
        for k in range(hops):
            mv = moves.get(k, [])
            if mv and batched:
                n = len(mv)
                st, sp = (C.c_int * n)(*[p[0] for p in mv]), (C.c_int * n)(*[p[1] for p in mv])
                bad_st = (C.c_int * n)(*([p[0] for p in mv[:-1]] + [B]))          # last stream out of range: nothing may change
                assert a.BeatriceBatch_SetTargetSpeakers(h, n, bad_st, sp) == -1
                assert a.BeatriceBatch_SetTargetSpeakers(h, n, st, sp) == 0
            else:
                for s, spk in mv:
                    assert a.BeatriceBatch_SetTargetSpeaker(h, s, spk) == 0
            got.append(batch.convert(np.ascontiguousarray(x[:, k])).copy())
        batch.close()
        outs.append(np.stack(got))
    m.close()
    assert np.abs(outs[0]).max() > 0.05
    assert np.array_equal(outs[0], outs[1])
