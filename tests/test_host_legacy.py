"""The host core of the legacy generations (ProcessorCoreLegacy, beatrice-vst_amd/host/processor_core_legacy.cc; reference
src/common/processor_core_{0,1}.cc) and the proxy's version dispatch (reference processor_proxy.h:57-70), on CPU against
the oracle: Process() must equal [oracle wrapper chain] o [legacy oracle hop with the reference's per-hop protocol] bit for
bit, including the speaker vector assembly, the 383 pitch clamp, per-hop morphing and the guards."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import hostlib
import wrapperlib
from test_host_proxy import Proxy, K_VOICE, K_FORMANT, K_PITCH_SHIFT, OK, SPEAKER_RANGE, NOT_LOADED

REPO = hostlib.REPO
GEN = {0: "20a2", 1: "20b1"}
VERSION = {0: "2.0.0-alpha.2", 1: "2.0.0-beta.1"}
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def host_path(built):
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "libhost_on_oracle.so"], env={k: v for k, v in os.environ.items() if k != "LD_PRELOAD"})
    return hostlib.HOST_ON_ORACLE


@pytest.fixture(scope="module")
def packages(tmp_path_factory):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_model
    out = {}
    for v in (0, 1):
        d = str(tmp_path_factory.mktemp("legacy%d" % v))
        make_model.make_model_legacy(d, n_speakers=3, version=VERSION[v])
        out[v] = d
    return out


class LegacyHost(hostlib.Host):
    def __init__(self, path, sample_rate, version):
        super().__init__(path, sample_rate)
        self.lib.BeatriceHost_Destroy(self.h)
        self.lib.BeatriceHost_CreateVersion.restype = C.c_void_p
        self.lib.BeatriceHost_CreateVersion.argtypes = [C.c_double, C.c_int]
        self.lib.BeatriceHost_SetSpeakerMorphingWeights.argtypes = [C.c_void_p, _f32p, C.c_int]
        self.lib.BeatriceHost_GetVersion.argtypes = [C.c_void_p]
        self.h = self.lib.BeatriceHost_CreateVersion(float(sample_rate), version)
        self.lib.BeatriceHost_EnablePitchTrace(self.h, 4096)


def spherical_mean(lib, points, weights, updates):
    """the solver exactly as the host drives it: SetWeights over ALL speakers, `updates` x Update, Result"""
    lib.BeatriceHost_SphericalMean.argtypes = [C.c_int, C.c_int, _f32p, _f32p, _i32p, C.c_int, C.c_int, _f32p]
    pts = np.ascontiguousarray(points, np.float32)
    w = np.ascontiguousarray(weights, np.float32)
    out = np.zeros(pts.shape[1], np.float32)
    lib.BeatriceHost_SphericalMean(pts.shape[1], pts.shape[0], pts.ctypes.data_as(_f32p), w.ctypes.data_as(_f32p), None, 0, updates,
                                   out.ctypes.data_as(_f32p))
    return out


def reference_chain(bv, version, pkg, sr, x, block, events):
    """oracle wrapper (pinned to the reference headers) driving the legacy oracle through processor_core_1.cc's hop"""
    abi = bv.AbiLegacy(os.path.join(REPO, "oracle", "libbeatrice_oracle.so"), GEN[version])
    m = bv.ModelsLegacy(abi, pkg)
    st = bv.StreamLegacy(m, speaker=0)
    # a fresh core has not been told a pitch range: the contexts' own defaults hold until SetMin/MaxSourcePitch
    bins, morph = [], {"w": None, "hops": 0}

    def hop(in160, out240, _user):
        if morph["w"] is not None and st.speaker == m.n_speakers:
            morph["hops"] += 1
            m.speakers[m.n_speakers] = spherical_mean(morph["lib"], m.speakers[:m.n_speakers], morph["w"], morph["hops"])
        o, _, _, _, q2, _ = st.hop(np.ctypeslib.as_array(in160, (160,)).copy(), return_all=True)
        bins.append(q2)
        for i in range(240):
            out240[i] = o[i]

    wo = wrapperlib.oracle_wrapper()
    cb = wrapperlib.HOP_FN(hop)
    p = wo.f_create(float(sr), cb, None)
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    for pos in range(0, len(x), block):
        for at, fn in events:
            if at == pos:
                fn(st, m, morph)
        n = min(block, len(x) - pos)
        wo.f_process(p, x[pos:pos + n].ctypes.data_as(_f32p), out[pos:pos + n].ctypes.data_as(_f32p), n)
    wo.f_destroy(p)
    st.close()
    m.close()
    return out, bins


@pytest.mark.parametrize("version,sr,block", [(1, 48000, 480), (1, 44100, 441), (0, 48000, 256), (0, 24000, 240)])
def test_legacy_core_matches_wrapper_oracle(bv, built, host_path, packages, version, sr, block):
    pkg = packages[version]
    n_blocks = max(6, int(0.16 * sr) // block)
    x = wrapperlib.test_signal(block * n_blocks, sr, seed=sr + version)
    cut = [block * (n_blocks * k // 4) for k in (1, 2, 3)]
    weights = np.zeros(256, np.float32)
    weights[:3] = (0.5, 0.2, 0.3)

    def ev1(st, m, morph):   # another speaker, a formant row, shifted and corrected pitch
        st.speaker, st.formant_index = 2, 7                                  # round(1.5 * 2 + 4)
        st.pitch_params = dict(shift=11.0, intonation=1.7, correction=0.4, ctype=0)

    def ev2(st, m, morph):   # morph slot: weights first (slot = the solver's start), then the speaker
        morph.update(w=weights[:3].copy(), hops=0, lib=C.CDLL(host_path))
        m.speakers[m.n_speakers] = spherical_mean(morph["lib"], m.speakers[:m.n_speakers], morph["w"], 0)
        st.speaker = m.n_speakers

    def ev3(st, m, morph):
        st.speaker, st.formant_index = 1, 4
        st.pitch_params = dict(shift=24.0, intonation=3.0)                    # drives bins into the 383 clamp

    want, want_bins = reference_chain(bv, version, pkg, sr, x, block, list(zip(cut, (ev1, ev2, ev3))))
    h = LegacyHost(host_path, sr, version)
    assert h.call("GetVersion") == version
    assert h.load(pkg) == 0 and h.call("NumSpeakers") == 3
    parts, codes = [], []
    o, c = h.process(x[:cut[0]], block); parts.append(o); codes += c
    assert h.call("SetTargetSpeaker", 2) == 0 and h.call("SetFormantShift", 1.5) == 0
    h.call("SetPitchShift", 11.0); h.call("SetIntonationIntensity", 1.7); h.call("SetPitchCorrection", 0.4); h.call("SetPitchCorrectionType", 0)
    o, c = h.process(x[cut[0]:cut[1]], block); parts.append(o); codes += c
    assert h.call("SetSpeakerMorphingWeights", weights.ctypes.data_as(_f32p), 256) == 0
    assert h.call("SetTargetSpeaker", 3) == 0
    o, c = h.process(x[cut[1]:cut[2]], block); parts.append(o); codes += c
    assert h.call("SetTargetSpeaker", 1) == 0 and h.call("SetFormantShift", 0.0) == 0
    h.call("SetPitchShift", 24.0); h.call("SetIntonationIntensity", 3.0); h.call("SetPitchCorrection", 0.0)
    o, c = h.process(x[cut[2]:], block); parts.append(o); codes += c
    got = np.concatenate(parts)
    assert set(codes) == {0}
    trace = h.pitch_trace()
    assert trace == want_bins and max(trace) <= 383
    assert np.array_equal(got, want), "max-abs %g" % np.abs(got - want).max()
    assert np.abs(got).max() > 1e-3
    h.close()


def test_legacy_guards(bv, built, host_path, packages, model_dir):
    h = LegacyHost(host_path, 48000, 1)
    x = wrapperlib.test_signal(960, 48000, seed=2)
    out, codes = h.process(x, 480)
    assert codes == [9, 9] and not out.any()                     # kModelNotLoaded -> zeros
    assert h.call("SetTargetSpeaker", 700) == 0                  # the legacy cores check the id when audio arrives (core_1.cc:233-240)
    assert h.call("SetTargetSpeaker", -1) == 7
    assert h.load(model_dir) != 0                                # an rc.0 package is not a legacy package
    out, codes = h.process(x, 480)
    assert codes == [9, 9] and not out.any()
    assert h.load(packages[1]) == 0
    out, codes = h.process(x, 480)
    assert codes == [7, 7] and not out.any()                     # id 700 > n_speakers: kSpeakerIDOutOfRange, zeros (core_1.cc:35-40)
    assert h.call("SetTargetSpeaker", 3) == 0                    # the morph slot (zeros until weights arrive) is a valid id
    out, codes = h.process(x, 480)
    assert codes == [0, 0]
    assert h.call("SetVQNumNeighbors", 5) == 0                   # setters the generation does not have succeed and do nothing
    assert h.call("SetPitchCorrectionType", 3) == 8
    h.close()


def test_legacy_reset_context_keeps_the_pitch_range(bv, built, host_path, packages):
    sr, block = 48000, 480
    x = wrapperlib.test_signal(block * 10, sr, seed=9)

    def run(reset_after):
        h = LegacyHost(host_path, sr, 1)
        assert h.load(packages[1]) == 0
        h.call("SetMinSourcePitch", 50.0); h.call("SetMaxSourcePitch", 62.0)
        a, _ = h.process(x, block)
        if reset_after:
            assert h.call("ResetContext") == 0
        b, _ = h.process(x, block)
        t = h.pitch_trace()
        h.close()
        return a, b, t

    a0, b0, t0 = run(False)
    a1, b1, t1 = run(True)
    lo, hi = round((50.0 - 33.0) * 8), round((62.0 - 33.0) * 8)   # 96 bins per octave = 8 per semitone
    assert np.array_equal(a0, a1) and all(lo <= q <= hi for q in t0 + t1)
    assert not np.array_equal(b0, b1)                            # the model state restarted ...
    fresh_tail = a1[5 * block:]                                  # ... onto the stream a fresh core produces (wrapper history flushed)
    assert np.abs(b1[5 * block:] - fresh_tail).max() < 0.5


@pytest.mark.parametrize("version", [0, 1])
def test_proxy_dispatches_on_the_package_version(bv, built, host_path, packages, version):
    pkg = packages[version]
    sr, block = 48000, 480
    x = wrapperlib.test_signal(block * 8, sr, seed=4)
    p = Proxy(sr)
    assert p.call("CoreVersion") == -1
    assert p.call("LoadModel", os.path.join(pkg, "model.toml").encode()) == OK
    assert p.call("CoreVersion") == version and p.call("VoiceCount") == 3
    assert p.call("SetInt", K_VOICE, 2) == OK and p.call("SetNumber", K_FORMANT, -1.0) == OK and p.call("SetNumber", K_PITCH_SHIFT, 5.0) == OK
    got, codes = p.process(x, block)
    assert set(codes) == {OK}
    h = LegacyHost(host_path, sr, version)
    assert h.load(pkg) == 0
    # what the proxy's SyncAllParameters leaves on a freshly loaded core (defaults of the parameter table), then the three edits
    h.call("SetAverageSourcePitch", p.call("GetNumber", 5)); h.call("SetMinSourcePitch", p.call("GetNumber", 12)); h.call("SetMaxSourcePitch", p.call("GetNumber", 13))
    h.call("SetTargetSpeaker", 2); h.call("SetFormantShift", -1.0); h.call("SetPitchShift", 5.0)
    want, _ = h.process(x, block)
    assert np.array_equal(got, want) and np.abs(got).max() > 1e-3
    h.close()
    p.close()


def test_a_speaker_table_without_rows_is_refused(bv, built, host_path, packages, tmp_path):
    """A crafted package whose speaker file holds no row: LoadModel answers kInvalidFileSize and the core stays unloaded
    (nothing may index an empty table: the morph solver's result, the formant row, the lottery)."""
    import shutil
    import struct
    pkg = tmp_path / "empty"
    shutil.copytree(packages[1], pkg)
    (pkg / "speaker_embeddings.bin").write_bytes(struct.pack("<IIII", 0x43525442, 15, 1, 0))
    h = LegacyHost(host_path, 48000, 1)
    rc = h.load(str(pkg))
    assert rc != 0
    out, codes = h.process(wrapperlib.test_signal(480, 48000, seed=1), 480)
    assert codes == [9] and not out.any()
    h.close()
