"""End to end through the C++ host layer: the product build (host layer + HIP library) against the
same host source on the oracle core, at several host sample rates (BASELINE.json configs[0]/[1]
plumbing: any-rate audio in, same-length audio out)."""
import numpy as np
import pytest

import hostlib
import wrapperlib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sr,block", [(24000, 240), (48000, 480), (44100, 512)])
def test_host_layer_hip_matches_oracle(built, model_dir, sr, block):
    x = wrapperlib.test_signal(block * max(4, int(0.25 * sr) // block), sr, seed=sr + 7)
    outs = {}
    for name, path in (("oracle", hostlib.HOST_ON_ORACLE), ("hip", hostlib.HOST_PRODUCT)):
        h = hostlib.Host(path, sr)
        assert h.load(model_dir) == 0
        h.call("SetVQNumNeighbors", 2)
        h.call("SetPitchShift", -2.0)
        half = block * (len(x) // block // 2)
        a, ca = h.process(x[:half], block)
        h.call("SetTargetSpeaker", 1)
        h.call("SetFormantShift", 1.0)
        b, cb = h.process(x[half:], block)
        assert set(ca + cb) == {0}
        outs[name] = (np.concatenate([a, b]), h.pitch_trace())
        h.close()
    assert outs["hip"][1] == outs["oracle"][1]
    dev = float(np.abs(outs["hip"][0] - outs["oracle"][0]).max())
    print("host layer sr=%d max-abs %g" % (sr, dev))
    assert np.abs(outs["hip"][0]).max() > 1e-3
    assert dev <= 1e-4


def test_host_layer_morph_hip_matches_oracle(built, model_dir):
    """Morph mode (target speaker == n_speakers) end to end: spherical means on the host, lottery
    codebook (fixed seed, VQ on), additive / K/V slot re-installed on the device library."""
    import ctypes as C
    sr, block = 48000, 480
    x = wrapperlib.test_signal(block * 20, sr, seed=31)
    w = np.zeros(256, np.float32)
    w[:3] = (0.2, 0.5, 0.3)
    f32p = C.POINTER(C.c_float)
    outs = {}
    for name, path in (("oracle", hostlib.HOST_ON_ORACLE), ("hip", hostlib.HOST_PRODUCT)):
        h = hostlib.Host(path, sr)
        h.lib.BeatriceHost_SetSpeakerMorphingWeights.argtypes = [C.c_void_p, f32p, C.c_int]
        h.lib.BeatriceHost_SetMorphSeed.argtypes = [C.c_void_p, C.c_uint]
        assert h.load(model_dir) == 0
        h.lib.BeatriceHost_SetMorphSeed(h.h, 99)
        h.call("SetVQNumNeighbors", 3)
        assert h.call("SetTargetSpeaker", h.call("NumSpeakers")) == 0
        assert h.lib.BeatriceHost_SetSpeakerMorphingWeights(h.h, w.ctypes.data_as(f32p), 256) == 0
        a, _ = h.process(x[:block * 10], block)
        w2 = w.copy()
        w2[:3] = (0.6, 0.0, 0.4)
        assert h.lib.BeatriceHost_SetSpeakerMorphingWeights(h.h, w2.ctypes.data_as(f32p), 256) == 0
        b, _ = h.process(x[block * 10:], block)
        outs[name] = np.concatenate([a, b])
        h.close()
    dev = float(np.abs(outs["hip"] - outs["oracle"]).max())
    print("morph max-abs", dev)
    assert np.abs(outs["hip"]).max() > 1e-3
    assert dev <= 1e-4


def test_proxy_loads_package_by_toml_on_the_product(built, model_dir):
    """`load a package by its .toml` as a product feature: ProcessorProxy (TOML reader, version dispatch, parameter
    fan-out, state blob) linked against the HIP library gives the samples of the same host code on the oracle core,
    and a state blob written by one restores the other."""
    import ctypes as C
    from test_host_proxy import K_MODEL, K_PITCH_SHIFT, K_VOICE, K_VQ, Proxy
    x = wrapperlib.test_signal(480 * 16, 48000, seed=77)
    outs, blobs = {}, {}
    for name, path in (("oracle", hostlib.HOST_ON_ORACLE), ("hip", hostlib.HOST_PRODUCT)):
        hostlib_path = hostlib.HOST_ON_ORACLE
        hostlib.HOST_ON_ORACLE = path   # Proxy binds whatever library this names
        try:
            p = Proxy()
        finally:
            hostlib.HOST_ON_ORACLE = hostlib_path
        assert p.call("SetString", K_MODEL, (model_dir + "/model.toml").encode()) == 0
        assert p.call("CoreVersion") == 2
        p.call("SetInt", K_VOICE, 2)
        p.call("SetNumber", K_VQ, 2.0)
        p.call("SetNumber", K_PITCH_SHIFT, -1.5)
        outs[name], codes = p.process(x)
        assert set(codes) == {0}
        blobs[name] = p.state()
        assert p.call("LoadModel", b"/nonexistent.toml") == 1 and p.call("CoreVersion") == -1   # unloaded: zeros
        z, _ = p.process(x[:960])
        assert not z.any()
        p.close()
    assert blobs["hip"] == blobs["oracle"]
    assert np.abs(outs["hip"]).max() > 1e-3
    assert np.abs(outs["hip"] - outs["oracle"]).max() <= 1e-4
