"""Speaker morphing on the device: BeatriceBatch_MorphSpeaker against the
host computation (SphericalMean, which is pinned bit-exact to the reference's SphericalAverage by
tests/test_morph.py), and end to end against oracle streams whose morph entry was filled by the host."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import hostlib

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_f32p, _i32p = C.POINTER(C.c_float), C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def model_dir8(tmp_path_factory):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_model
    d = str(tmp_path_factory.mktemp("model8m"))
    make_model.make_model(d, n_speakers=8)
    return d


@pytest.fixture(scope="module")
def host_lib(built):
    lib = C.CDLL(hostlib.HOST_ON_ORACLE)  # test infrastructure: the host solver, CPU only
    lib.BeatriceHost_SphericalMean.argtypes = [C.c_int, C.c_int, _f32p, _f32p, _i32p, C.c_int, C.c_int, _f32p]
    return lib


def _prepare(weights):
    """reference weight preparation: < 0.01 dropped, descending order, eight kept"""
    w = np.array(weights, np.float32)
    w[w < np.float32(0.01)] = 0
    order = np.argsort(-w, kind="stable").astype(np.int32)
    pruned = np.zeros_like(w)
    pruned[order[:8]] = w[order[:8]]
    return pruned, order


def _host_mean(lib, pts, pruned, order):
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.zeros(pts.shape[1], np.float32)
    lib.BeatriceHost_SphericalMean(pts.shape[1], pts.shape[0], pts.ctypes.data_as(_f32p), pruned.ctypes.data_as(_f32p),
                                   order.ctypes.data_as(_i32p), 8, 4, out.ctypes.data_as(_f32p))
    return out


def _host_entry(lib, tables, n, weights):
    pruned, order = _prepare(weights)
    add = _host_mean(lib, tables.additive[:n], pruned, order)
    kv = np.stack([_host_mean(lib, tables.kv[:n, tok], pruned, order) for tok in range(tables.kv.shape[1])])
    return add, kv, pruned, order


@pytest.mark.parametrize("weights", [
    [0.2, 0.5, 0.3, 0, 0, 0, 0, 0],
    [0.05, 0.3, 0.005, 0.2, 0.1, 0.15, 0.1, 0.1],     # one weight below the 0.01 threshold
    [0, 0, 0, 1.0, 0, 0, 0, 0],                       # a single speaker: the mean is that speaker
    [0.125] * 8,
])
def test_morphed_embeddings_match_host(bv, product, host_lib, model_dir8, weights):
    m = bv.Models(product, model_dir8)
    t = m.tables
    n = t.n_speakers
    batch = bv.Batch(m, 2)
    w = np.array(weights, np.float32)
    assert batch.a.BeatriceBatch_MorphSpeaker(batch.h, n, bv.fptr(w), n, 7) == 0
    add = np.zeros(256, np.float32)
    kv = np.zeros((384, 128), np.float32)
    assert batch.a.BeatriceBatch_GetSpeakerEmbeddings(batch.h, n, bv.fptr(add), bv.fptr(kv)) == 0
    ref_add, ref_kv, _, _ = _host_entry(host_lib, t, n, weights)
    batch.close()
    m.close()
    scale = max(float(np.abs(ref_add).max()), float(np.abs(ref_kv).max()))
    dev = max(float(np.abs(add - ref_add).max()), float(np.abs(kv - ref_kv).max()))
    print("weights %s: max-abs %g (embedding scale %.2f)" % (weights[:4], dev, scale))
    assert scale > 0.1
    assert dev <= 1e-5 * max(1.0, scale)
    if weights[3] == 1.0:  # the degenerate morph reproduces the speaker (up to the normalise / de-normalise round trip)
        assert np.abs(add - t.additive[3]).max() <= 1e-5 * scale


def _mt_draws(seed):
    """std::uniform_real_distribution<float>(0, s)(std::mt19937(seed)) as libstdc++ computes it"""
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)
    while True:
        c = np.float32(np.float32(int(bg.random_raw())) / np.float32(4294967296.0))
        if c >= np.float32(1.0):
            c = np.nextafter(np.float32(1.0), np.float32(0.0))
        yield c


@pytest.mark.parametrize("vq_k", [0, 4])
def test_morph_end_to_end(bv, oracle, product, host_lib, model_dir8, vq_k):
    """Streams on a morphed entry: embeddings from the device solver, K/V blocks installed one per hop, and
    (k > 0) a codebook drawn per hop with the weights as odds -- against oracle streams driven with the
    host-computed entry and the same draws."""
    weights = [0.1, 0.0, 0.45, 0.0, 0.25, 0.2, 0.0, 0.0]
    seed, B, hops = 1234, 3, 18
    audio = np.stack([bv.synth_audio(160 * hops, seed=900 + s) for s in range(B)])
    switch_hop = {0: 0, 1: 6}                     # stream 2 never morphs

    # ---- oracle side: the morph entry of the caller-owned tables filled by the host solver
    mo = bv.Models(oracle, model_dir8)
    t = mo.tables
    n = t.n_speakers
    add, kv, pruned, order = _host_entry(host_lib, t, n, weights)
    t.additive[n] = add
    t.kv[n] = kv
    odds = pruned[order[:8]]
    total = np.float32(0)
    for v in odds:
        total = np.float32(total + v)
    draws = {s: _mt_draws(seed + s) for s in range(B)}   # every stream owns its engine: std::mt19937(seed + stream)
    streams = [bv.Stream1(mo, speaker=s, vq_k=vq_k) for s in range(B)]
    ref = np.zeros((hops, B, bv.OUT_HOP), np.float32)
    picks = []
    for h in range(hops):
        for s in range(B):
            if switch_hop.get(s) == h:
                streams[s].set_target_speaker(n)
            if s in switch_hop and h >= switch_hop[s]:   # one draw per morphing stream and hop
                r = np.float32(next(draws[s]) * total)
                idx = int(order[0])
                for i in range(8):
                    r = np.float32(r - odds[i])
                    if r < 0:
                        idx = int(order[i])
                        break
                picks.append(idx)
                streams[s].a.SetCodebook(streams[s].pc, bv.fptr(t.codebooks[idx]))
        for s in range(B):
            ref[h, s] = streams[s].hop(audio[s, h * 160:(h + 1) * 160])
    for st in streams:
        st.close()
    mo.close()

    # ---- product side
    m = bv.Models(product, model_dir8)
    batch = bv.Batch(m, B)
    a, hnd = batch.a, batch.h
    w = np.array(weights, np.float32)
    assert a.BeatriceBatch_MorphSpeaker(hnd, n, bv.fptr(w), n, seed) == 0
    for s in range(B):
        a.BeatriceBatch_SetTargetSpeaker(hnd, s, s)
        a.BeatriceBatch_SetVQNumNeighbors(hnd, s, vq_k)
    a.BeatriceBatch_FlushSpeaker(hnd, -1)
    got = np.zeros_like(ref)
    for h in range(hops):
        for s in range(B):
            if switch_hop.get(s) == h:
                a.BeatriceBatch_SetTargetSpeaker(hnd, s, n)
        got[h] = batch.convert(audio[:, h * 160:(h + 1) * 160])
    batch.close()
    m.close()
    dev = float(np.abs(ref - got).max())
    print("morph end to end k=%d: max-abs %g; codebook draws %s" % (vq_k, dev, picks[:10]))
    assert len(set(picks)) > 1 and set(picks) <= {0, 2, 4, 5}
    assert np.abs(got).max() > 0.05
    assert dev <= 1e-4


def test_moving_morph_weights_does_not_restart_the_lottery(bv, product, model_dir8):
    """The reference seeds its lottery engine once per instance and never when morph weights change
    (processor_core_2.h:48,145, .cc:94-121).  A caller that re-sends the (same) weights before every step must therefore
    get exactly the samples of a caller that sent them once -- the draw sequence runs on -- and a morph on a second entry
    must not restart the draws of the streams on the first."""
    B, hops = 6, 16
    audio = np.stack([bv.synth_audio(160 * hops, seed=1500 + s) for s in range(B)])
    m = bv.Models(product, model_dir8)
    n = m.tables.n_speakers
    w = np.array([0.1, 0.0, 0.45, 0.0, 0.25, 0.2, 0.0, 0.0], np.float32)
    w2 = np.array([0.5, 0.5, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], np.float32)

    def run(resend, second_entry_at=None):
        batch = bv.Batch(m, B, max_speakers=n + 2)
        a, h = batch.a, batch.h
        assert a.BeatriceBatch_MorphSpeaker(h, n, bv.fptr(w), n, 99) == 0
        a.BeatriceBatch_SetTargetSpeaker(h, -1, n)
        a.BeatriceBatch_SetVQNumNeighbors(h, -1, 2)
        a.BeatriceBatch_FlushSpeaker(h, -1)
        out = []
        for k in range(hops):
            if resend and k > 0:
                assert a.BeatriceBatch_MorphSpeaker(h, n, bv.fptr(w), n, 99) == 0
                a.BeatriceBatch_FlushSpeaker(h, -1)
            if second_entry_at == k:
                assert a.BeatriceBatch_MorphSpeaker(h, n + 1, bv.fptr(w2), n, 5) == 0
            out.append(batch.convert(np.ascontiguousarray(audio[:, k * 160:(k + 1) * 160])))
        batch.close()
        return np.stack(out)

    once = run(False)
    assert np.array_equal(run(True), once), "re-sending the weights restarted the draw sequences"
    assert np.array_equal(run(False, second_entry_at=7), once), "a morph on another entry restarted the draw sequences"
    # and the lottery does draw: with a frozen first variate every hop would use one codebook; compare with k-NN off
    m.close()
    assert np.abs(once).max() > 0.05


def test_staged_morph_follows_the_reference_timeline(bv, oracle, product, model_dir8):
    """the reference-faithful install of a weight change for streams that are already morphing
    (processor_core_2.cc:51-177): additive embedding on the next hop, the OLD key/value blocks for four more hops, the new ones
    one block per hop after that.  Oracle streams are driven through exactly that protocol with the embeddings the device
    solver produced (read back), so the comparison is bit for bit; k-NN off (no lottery in the way)."""
    B, hops, change_at = 3, 24, 7
    audio = np.stack([bv.synth_audio(160 * hops, seed=700 + s) for s in range(B)])
    w1 = np.array([0.5, 0.0, 0.3, 0.2, 0.0, 0.0, 0.0, 0.0], np.float32)
    w2 = np.array([0.0, 0.6, 0.0, 0.0, 0.1, 0.3, 0.0, 0.0], np.float32)
    m = bv.Models(product, model_dir8)
    batch = bv.Batch(m, B, max_speakers=m.tables.n_speakers + 3)   # room for two morph entries
    a, hnd = batch.a, batch.h
    n = m.tables.n_speakers
    e1, e2 = n, n + 1
    assert a.BeatriceBatch_MorphSpeaker(hnd, e1, bv.fptr(w1), n, 5) == 0
    a.BeatriceBatch_SetTargetSpeaker(hnd, 0, e1)
    a.BeatriceBatch_SetTargetSpeaker(hnd, 1, e1)
    a.BeatriceBatch_SetTargetSpeaker(hnd, 2, 1)           # a stream that is not morphing keeps out of it
    a.BeatriceBatch_FlushSpeaker(hnd, -1)
    emb = {}

    def read_entry(e):
        add, kv = np.zeros(bv.HID, np.float32), np.zeros((bv.KV_LEN, bv.KV_CH), np.float32)
        assert a.BeatriceBatch_GetSpeakerEmbeddings(hnd, e, bv.fptr(add), bv.fptr(kv)) == 0
        return add, kv

    emb[e1] = read_entry(e1)
    got = np.zeros((hops, B, bv.OUT_HOP), np.float32)
    for h in range(hops):
        if h == change_at:
            assert a.BeatriceBatch_MorphSpeakerStaged(hnd, e1, e1, bv.fptr(w2), n, 5) == -1       # in place is what MorphSpeaker does
            assert a.BeatriceBatch_MorphSpeakerStaged(hnd, e2, e1, bv.fptr(w2), n, 5) == 0
            emb[e2] = read_entry(e2)
        if h == change_at + 2:
            assert a.BeatriceBatch_MorphSpeakerStaged(hnd, e1, e2, bv.fptr(w1), n, 5) == -3       # e1's blocks are still installed
        got[h] = batch.convert(audio[:, h * 160:(h + 1) * 160])
    batch.close()
    m.close()

    mo = bv.Models(oracle, model_dir8)
    t = mo.tables
    want = np.zeros_like(got)
    for s in range(B):
        st = bv.Stream1(mo, speaker=1 if s == 2 else 0, vq_k=0)
        if s < 2:   # onto the first morph, all blocks at once (FlushSpeaker)
            st.a.SetAdditiveSpeakerEmbedding(mo.embed, bv.fptr(emb[e1][0]), st.ec, st.wc)
            st.a.RegisterKeyValueSpeakerEmbedding(mo.embed, bv.fptr(emb[e1][1]), st.ec)
            st.kv_count = 0
            while st.set_kv_block():
                pass
        for h in range(hops):
            if s < 2 and h == change_at:          # morph_counter_ == 0: the additive embedding at once
                st.a.SetAdditiveSpeakerEmbedding(mo.embed, bv.fptr(emb[e2][0]), st.ec, st.wc)
            if s < 2 and h == change_at + 4:      # morph_counter_ == 4: register; Stream1.hop installs one block per hop from here
                st.a.RegisterKeyValueSpeakerEmbedding(mo.embed, bv.fptr(emb[e2][1]), st.ec)
                st.kv_count = 0
            want[h, s] = st.hop(audio[s, h * 160:(h + 1) * 160])
        st.close()
    mo.close()
    assert np.abs(got).max() > 0.05
    for h in range(hops):
        assert np.array_equal(got[h], want[h]), "hop %d max-abs %g" % (h, np.abs(got[h] - want[h]).max())
    # the change is audible where the protocol says: nothing before h0, and the streams differ from an "all at once" install
    assert not np.array_equal(got[change_at:, :2], 0 * got[change_at:, :2])
