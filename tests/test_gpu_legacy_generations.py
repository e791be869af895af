"""The legacy generations on the GPU: Beatrice20a2_* / Beatrice20b1_* of the HIP library against the oracle's (MODEL_SPEC
section 6), through the per-hop protocol of the reference's ProcessorCore0 / ProcessorCore1
(reference src/common/processor_core_1.cc:50-143): module outputs and PCM, speaker and formant changes between hops,
restricted pitch ranges, context resets."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


@pytest.fixture(scope="module")
def legacy_dir(tmp_path_factory):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_model
    d = str(tmp_path_factory.mktemp("legacy_gpu"))
    make_model.make_model_legacy(d, n_speakers=4)
    return d


def _run(bv, abi, legacy_dir, x, hops, script):
    m = bv.ModelsLegacy(abi, legacy_dir)
    st = bv.StreamLegacy(m)
    rec = []
    for h in range(hops):
        st = script(h, st, m) or st
        rec.append(st.hop(x[h * 160:(h + 1) * 160], return_all=True))
    st.close()
    m.close()
    return (np.stack([r[0] for r in rec]), np.stack([r[1] for r in rec]), np.array([r[2] for r in rec]),
            np.stack([r[3] for r in rec]), np.array([r[4] for r in rec]))


@pytest.mark.parametrize("generation", ["20b1", "20a2"])
def test_legacy_generation_matches_oracle(bv, built, product, legacy_dir, generation):
    hops = 40
    x = bv.synth_audio(160 * hops, seed=4242)
    oracle = bv.AbiLegacy(os.path.join(REPO, "oracle", "libbeatrice_oracle.so"), generation)
    hip = bv.AbiLegacy(bv.PRODUCT_LIB, generation)

    def script(h, st, m):
        if h == 5:
            st.speaker, st.formant_index = 2, 7
        if h == 11:
            st.pitch_params = dict(shift=3.0, correction=0.5, ctype=1)
        if h == 17:
            st.a.SetMinQuantizedPitch(st.tc, 120)
            st.a.SetMaxQuantizedPitch(st.tc, 200)
        if h == 23:
            st.speaker = 3
            st.a.SetMaxQuantizedPitch(st.tc, 100)      # max < min: the range collapses to min
        if h == 29:                                    # ResetContext: fresh contexts, settings re-applied (processor_core_1.cc:145-163)
            new = bv.StreamLegacy(m, speaker=st.speaker, formant_index=st.formant_index, min_q=30, max_q=383)
            new.pitch_params = st.pitch_params
            st.close()
            return new
        return None

    want = _run(bv, oracle, legacy_dir, x, hops, script)
    got = _run(bv, hip, legacy_dir, x, hops, script)
    assert np.abs(want[0]).max() > 0.05 and len(set(want[2].tolist())) > 3
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[4], want[4]), "pitch bins differ"
    dev = [float(np.abs(got[i] - want[i]).max()) for i in (0, 1, 3)]
    print("legacy %s: PCM max-abs %g, phone %g, features %g %s" % (generation, dev[0], dev[1], dev[2],
                                                                   "bit-identical" if np.array_equal(got[0], want[0]) else ""))
    assert dev[0] <= TOL and dev[1] <= TOL and dev[2] <= TOL


def test_legacy_objects_do_not_leak_and_unloaded_objects_stay_silent(bv, product, legacy_dir):
    import ctypes as C
    hiprt = C.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = C.c_size_t(0), C.c_size_t(0)
        assert hiprt.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    hip = bv.AbiLegacy(bv.PRODUCT_LIB, "20b1")
    x = bv.synth_audio(160 * 3, seed=1)

    def cycle():
        m = bv.ModelsLegacy(hip, legacy_dir)
        st = bv.StreamLegacy(m, speaker=1)
        for h in range(3):
            st.hop(x[h * 160:(h + 1) * 160])
        st.close()
        m.close()

    for _ in range(3):
        cycle()
    before = free_bytes()
    for _ in range(20):
        cycle()
    assert before - free_bytes() <= 8 << 20
    # objects that never saw Read*Parameters: zeros, bin 1
    pe, pc = hip.CreatePhoneExtractor(), hip.CreatePhoneContext1()
    out = np.full(256, 3.0, np.float32)
    hip.ExtractPhone1(pe, bv.fptr(x[:160].copy()), bv.fptr(out), pc)
    assert not out.any()
    hip.DestroyPhoneContext1(pc)
    hip.DestroyPhoneExtractor(pe)


@pytest.mark.parametrize("version,gen_version", [(1, "2.0.0-beta.1"), (0, "2.0.0-alpha.2")])
def test_proxy_runs_a_legacy_package_on_the_product(built, tmp_path, version, gen_version):
    """a `2.0.0-beta.1` (and `2.0.0-alpha.2`) package through ProcessorProxy -> ProcessorCoreLegacy ->
    Beatrice20b1_* (Beatrice20a2_*) on the HIP library = the same host source on the oracle core, with a speaker change,
    a formant row, the morph slot and a context reset on the way."""
    import ctypes as C
    import os
    import sys
    import hostlib
    import wrapperlib
    from test_host_proxy import K_FORMANT, K_MODEL, K_PITCH_SHIFT, K_VOICE, Proxy
    sys.path.insert(0, os.path.join(hostlib.REPO, "tools"))
    import make_model
    pkg = str(tmp_path / "pkg")
    os.makedirs(pkg)
    make_model.make_model_legacy(pkg, n_speakers=3, version=gen_version)
    sr, block = 44100, 441
    x = wrapperlib.test_signal(block * 40, sr, seed=17 + version)
    outs = {}
    for name, path in (("oracle", hostlib.HOST_ON_ORACLE), ("hip", hostlib.HOST_PRODUCT)):
        keep = hostlib.HOST_ON_ORACLE
        hostlib.HOST_ON_ORACLE = path   # Proxy binds whatever library this names
        try:
            p = Proxy(sr)
        finally:
            hostlib.HOST_ON_ORACLE = keep
        assert p.call("SetString", K_MODEL, (pkg + "/model.toml").encode()) == 0
        assert p.call("CoreVersion") == version
        parts = []
        o, codes = p.process(x[:block * 10], block); parts.append(o)
        assert set(codes) == {0}
        p.call("SetInt", K_VOICE, 2); p.call("SetNumber", K_FORMANT, 1.5); p.call("SetNumber", K_PITCH_SHIFT, 7.0)
        o, _ = p.process(x[block * 10:block * 20], block); parts.append(o)
        p.call("SetInt", K_VOICE, 3)                                  # morph slot: the proxy's default markers give the weights
        o, _ = p.process(x[block * 20:block * 30], block); parts.append(o)
        assert p.call("ResetContext") == 0
        p.call("SetInt", K_VOICE, 1)
        o, _ = p.process(x[block * 30:], block); parts.append(o)
        outs[name] = np.concatenate(parts)
        p.close()
    assert np.abs(outs["hip"]).max() > 1e-3
    assert np.array_equal(outs["hip"], outs["oracle"]), "max-abs %g" % np.abs(outs["hip"] - outs["oracle"]).max()
