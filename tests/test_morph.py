"""Speaker morphing on the host: the product's SphericalMean against the
reference's SphericalAverage (library built from the reference header, and golden vectors minted from
it), and the morph branch of one hop against a step-by-step emulation of reference
src/common/processor_core_2.cc:51-177 that uses the REFERENCE solver for the means."""
import ctypes as C
import os

import numpy as np
import pytest

import hostlib
import wrapperlib

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_f32p, _i32p = C.POINTER(C.c_float), C.POINTER(C.c_int)
_SIG = [C.c_int, C.c_int, _f32p, _f32p, _i32p, C.c_int, C.c_int, _f32p]


def _mean(lib, fn, pts, w, order, limit, updates):
    f = getattr(lib, fn)
    f.argtypes = _SIG
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.zeros(pts.shape[1], np.float32)
    it = f(pts.shape[1], pts.shape[0], pts.ctypes.data_as(_f32p), np.ascontiguousarray(w, np.float32).ctypes.data_as(_f32p),
           np.ascontiguousarray(order, np.int32).ctypes.data_as(_i32p), limit, updates, out.ctypes.data_as(_f32p))
    return it, out


@pytest.fixture(scope="module")
def host_lib(built):
    if not os.path.exists(hostlib.HOST_ON_ORACLE):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(hostlib.REPO, "oracle"), "libhost_on_oracle.so"], env={k: v for k, v in os.environ.items() if k != "LD_PRELOAD"})
    return C.CDLL(hostlib.HOST_ON_ORACLE)


def _cases():
    rng = np.random.default_rng(2024)
    for trial in range(60):
        dim = 128 if trial % 2 else 256
        n = int(rng.integers(2, 9))
        base = rng.standard_normal(dim)
        pts = (base[None, :] * rng.uniform(0, 2) + rng.standard_normal((n, dim))).astype(np.float32)
        w = np.zeros(n, np.float32)
        k = int(rng.integers(1, n + 1))
        w[:k] = rng.random(k).astype(np.float32) + np.float32(0.01)
        rng.shuffle(w)
        order = np.argsort(-w, kind="stable").astype(np.int32)
        yield pts, w, order, min(n, 8), int(rng.integers(0, 5))


def test_spherical_mean_matches_golden(host_lib):
    g = np.load(os.path.join(GOLD, "morph_spherical_mean.npz"))
    for i in range(int(g["n_cases"][0])):
        it, out = _mean(host_lib, "BeatriceHost_SphericalMean", g["pts_%d" % i], g["w_%d" % i], g["order_%d" % i],
                        int(g["limit_%d" % i][0]), int(g["updates_%d" % i][0]))
        assert it == int(g["it_%d" % i][0])
        assert np.array_equal(out, g["out_%d" % i]), "case %d max-abs %g" % (i, np.abs(out - g["out_%d" % i]).max())


def test_spherical_mean_matches_reference_live(host_lib):
    ref = wrapperlib.ref_wrapper()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    for pts, w, order, limit, updates in _cases():
        a = _mean(ref.lib, "ref_spherical_average", pts, w, order, limit, updates)
        b = _mean(host_lib, "BeatriceHost_SphericalMean", pts, w, order, limit, updates)
        assert a[0] == b[0] and np.array_equal(a[1], b[1])


def test_morph_branch_matches_emulation_with_reference_solver(bv, oracle, host_lib, model_dir):
    """target speaker == n_speakers, VQ off (so the codebook lottery cannot influence the output):
    additive mean on the first hop after a weight change, K/V token means over four hops (96 tokens
    each), re-registration on the fifth, then one K/V block per hop."""
    ref = wrapperlib.ref_wrapper()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    sr, block, hops = 48000, 480, 16
    x = wrapperlib.test_signal(block * hops, sr, seed=77)
    weights = np.zeros(256, np.float32)
    weights[:3] = (0.5, 0.3, 0.2)

    # --- product host layer (on the oracle core)
    h = hostlib.Host(hostlib.HOST_ON_ORACLE, sr)
    h.lib.BeatriceHost_SetSpeakerMorphingWeights.argtypes = [C.c_void_p, _f32p, C.c_int]
    h.lib.BeatriceHost_SetMorphSeed.argtypes = [C.c_void_p, C.c_uint]
    assert h.load(model_dir) == 0
    h.lib.BeatriceHost_SetMorphSeed(h.h, 1234)
    n_spk = h.call("NumSpeakers")
    assert h.call("SetTargetSpeaker", n_spk) == 0
    assert h.lib.BeatriceHost_SetSpeakerMorphingWeights(h.h, weights.ctypes.data_as(_f32p), 256) == 0
    got, codes = h.process(x, block)
    assert set(codes) == {0}
    h.close()

    # --- emulation: oracle wrapper + oracle core + the reference's SphericalAverage
    m = bv.Models(oracle, model_dir)
    t = m.tables
    st = bv.Stream1(m, speaker=0, vq_k=0)
    order = np.array([0, 1, 2], np.int32)
    pruned = weights[:3].copy()
    state = {"counter": 0}
    # SetTargetSpeaker(n_speakers): morph slot (zeros) registered, K/V blocks restart
    st.set_target_speaker(n_spk)

    def hop(in160, out240, _u):
        c = state["counter"]
        if c == 0:
            _, add = _mean(ref.lib, "ref_spherical_average", t.additive[:n_spk], pruned, order, min(n_spk, 8), 4)
            t.additive[n_spk] = add
            st.a.SetAdditiveSpeakerEmbedding(m.embed, bv.fptr(t.additive[n_spk]), st.ec, st.wc)
        if c < 4:
            for i in range(384 * c // 4, 384 * (c + 1) // 4):
                _, tok = _mean(ref.lib, "ref_spherical_average", np.ascontiguousarray(t.kv[:n_spk, i, :]), pruned, order, min(n_spk, 8), 4)
                t.kv[n_spk, i] = tok
        elif c == 4:
            st.a.RegisterKeyValueSpeakerEmbedding(m.embed, bv.fptr(t.kv[n_spk]), st.ec)
            st.kv_count = 0
        if c <= 4:
            state["counter"] = c + 1
        o = st.hop(np.ctypeslib.as_array(in160, (160,)).copy())
        for i in range(240):
            out240[i] = o[i]

    want = wrapperlib.oracle_wrapper().run_chain(sr, x, block, hop=hop)
    st.close()
    m.close()
    assert np.abs(want).max() > 1e-3
    assert np.array_equal(got, want), "max-abs %g" % np.abs(got - want).max()
