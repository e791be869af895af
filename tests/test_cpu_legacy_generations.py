"""The legacy generations (Beatrice20a2_*, Beatrice20b1_*; reference lib/beatricelib/beatrice.h:39-203, callers
src/common/processor_core_0.cc / processor_core_1.cc) on the CPU oracle: the package readers and their error codes, the
per-hop protocol, and the oracle against the independent numpy restatement of MODEL_SPEC section 6."""
import os
import struct
import sys

import numpy as np
import pytest

import spec_numpy as sn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOPS = 12


@pytest.fixture(scope="module")
def legacy_dir(tmp_path_factory):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_model
    d = str(tmp_path_factory.mktemp("legacy"))
    make_model.make_model_legacy(d, n_speakers=3)
    return d


@pytest.fixture(scope="module")
def oracle_legacy(bv, built):
    return {g: bv.AbiLegacy(os.path.join(REPO, "oracle", "libbeatrice_oracle.so"), g) for g in ("20a2", "20b1")}


def test_readers_and_error_codes(bv, oracle_legacy, legacy_dir, model_dir, tmp_path):
    a = oracle_legacy["20b1"]
    m = bv.ModelsLegacy(a, legacy_dir)
    assert m.n_speakers == 3 and np.abs(m.speakers[:3]).max() > 0.1 and not m.speakers[3].any() and np.abs(m.formant).max() > 0.01
    m.close()
    obj = a.CreatePhoneExtractor()
    assert a.ReadPhoneExtractorParameters(obj, b"/nonexistent/phone_extractor.bin") == 1            # Beatrice_kFileOpenError
    # an rc.0 file is not a legacy file (kind differs): invalid, and the object stays unloaded (zeros out)
    assert a.ReadPhoneExtractorParameters(obj, os.path.join(model_dir, "phone_extractor.bin").encode()) == 4
    raw = open(os.path.join(legacy_dir, "phone_extractor.bin"), "rb").read()
    short, long_ = tmp_path / "short.bin", tmp_path / "long.bin"
    short.write_bytes(raw[:-400])
    long_.write_bytes(raw + b"\0" * 400)
    assert a.ReadPhoneExtractorParameters(obj, str(short).encode()) == 2                             # too small
    assert a.ReadPhoneExtractorParameters(obj, str(long_).encode()) == 3                             # too large
    ctx = a.CreatePhoneContext1()
    out = np.full(256, 7.0, np.float32)
    a.ExtractPhone1(obj, bv.fptr(np.zeros(160, np.float32)), bv.fptr(out), ctx)
    assert not out.any()
    a.DestroyPhoneContext1(ctx)
    a.DestroyPhoneExtractor(obj)
    n = np.zeros(1, np.int32)
    assert a.ReadNSpeakers(os.path.join(legacy_dir, "formant_shift_embeddings.bin").encode(), bv.iptr(n)) == 0 and n[0] == 9
    bad = tmp_path / "rows.bin"
    bad.write_bytes(struct.pack("<IIII", 0x43525442, 15, 1, 300) + b"\0" * 1200)                     # 300 floats: not rows of 256
    assert a.ReadNSpeakers(str(bad).encode(), bv.iptr(n)) == 4


def test_both_generations_run_the_same_network(bv, oracle_legacy, legacy_dir):
    x = bv.synth_audio(160 * HOPS, seed=31)
    outs = {}
    for g, a in oracle_legacy.items():
        m = bv.ModelsLegacy(a, legacy_dir)
        st = bv.StreamLegacy(m, speaker=1, formant_index=6)
        outs[g] = np.stack([st.hop(x[h * 160:(h + 1) * 160]) for h in range(HOPS)])
        st.close()
        m.close()
    assert np.abs(outs["20b1"]).max() > 0.05
    assert np.array_equal(outs["20a2"], outs["20b1"])


def test_legacy_oracle_matches_independent_restatement(bv, oracle_legacy, legacy_dir):
    """MODEL_SPEC section 6 restated in numpy float64 (offline, whole utterance) against the streaming C oracle, module by
    module: phone vectors, pitch features and bins where the decision is clear, PCM from the oracle's own intermediates."""
    a = oracle_legacy["20b1"]
    m = bv.ModelsLegacy(a, legacy_dir)
    x = bv.synth_audio(160 * HOPS, seed=2025)
    st = bv.StreamLegacy(m, speaker=2, formant_index=1, min_q=1, max_q=383)
    outs = [st.hop(x[h * 160:(h + 1) * 160], return_all=True) for h in range(HOPS)]
    st.close()
    m.close()
    pcm = np.concatenate([o[0] for o in outs])
    phone, q, feat = np.stack([o[1] for o in outs]), np.array([o[2] for o in outs]), np.stack([o[3] for o in outs])
    spk = np.stack([o[5] for o in outs])
    want_phone = sn.PhoneExtractor(legacy_dir, out_ch=256, kind=11)(x)[0]
    dev = float(np.abs(phone - want_phone).max())
    print("legacy phone: max-abs %.3g at scale %.3g" % (dev, np.abs(want_phone).max()))
    assert np.abs(want_phone).max() > 0.1 and dev <= 2e-5 * max(1.0, float(np.abs(want_phone).max()))
    bins, f, logits = sn.PitchEstimator(legacy_dir, bins=384, kind=12)(x, 1, 383)
    top2 = np.sort(logits[:, 1:384], axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-3
    assert clear.sum() >= HOPS // 2 and np.array_equal(q[clear], bins[clear])
    same = q == bins
    assert float(np.abs(feat[same, 0] - f[same, 0]).max()) <= 2e-5 and float(np.abs(feat[:, 1] - f[:, 1]).max()) <= 2e-5
    assert float(np.abs(feat[:, 3] - f[:, 3]).max()) <= 2e-5
    want = sn.LegacyWaveformGenerator(legacy_dir)(phone, q, feat, spk)
    dev = float(np.abs(pcm - want).max())
    print("legacy waveform: max-abs %.3g, rms %.3g" % (dev, np.sqrt((want ** 2).mean())))
    assert np.sqrt((want ** 2).mean()) > 0.02 and dev <= 1e-4
