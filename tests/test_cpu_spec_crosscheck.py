"""The oracle against an independent reading of MODEL_SPEC.md (tests/spec_numpy.py: float64, offline, written from
the spec text).  One author wrote the spec, the scalar C oracle and the HIP kernels, so a shared misreading would pass
every HIP-vs-oracle test bit for bit; this test breaks that correlation per module, and checks "streaming == offline"
on the way (the oracle runs hop by hop with per-stream state, the restatement convolves whole utterances).
Tolerances are float32-rounding sized; the discrete outputs (pitch bin, k-NN selection) are compared where the
float64 margins say the decision is not a rounding coin-flip."""
import numpy as np
import pytest

import spec_numpy as sn

HOPS = 14


@pytest.fixture(scope="module")
def driven(bv, oracle, model_dir):
    """One oracle stream through the reference's per-hop protocol, every intermediate kept."""
    m = bv.Models(oracle, model_dir)
    x = bv.synth_audio(160 * HOPS, seed=2024)
    rec = {}
    for k in (0, 3):
        st = bv.Stream1(m, speaker=1, formant_index=6, vq_k=k, min_q=1, max_q=447)
        outs = [st.hop(x[h * 160:(h + 1) * 160], return_all=True) for h in range(HOPS)]
        st.close()
        rec[k] = dict(pcm=np.concatenate([o[0] for o in outs]), phone=np.stack([o[1] for o in outs]),
                      q=np.array([o[2] for o in outs]), feat=np.stack([o[3] for o in outs]))
    tables = m.tables
    m.close()
    return x, rec, tables


def test_phone_extractor_matches_independent_restatement(driven, model_dir):
    x, rec, tables = driven
    pe = sn.PhoneExtractor(model_dir)
    want, raw, _ = pe(x)
    got = rec[0]["phone"]
    scale = float(np.abs(want).max())
    dev = float(np.abs(got - want).max())
    print("phone (no VQ): max-abs %.3g at scale %.3g" % (dev, scale))
    assert scale > 0.1 and dev <= 2e-5 * max(1.0, scale)
    # k-NN: same neighbours wherever the (k+1)-th candidate is clearly farther than the k-th
    vq, raw, d = pe(x, codebook=tables.codebooks[1], k=3)
    srt = np.sort(d, axis=1)
    clear = (srt[:, 3] - srt[:, 2]) > 1e-3
    assert clear.sum() >= HOPS // 2
    dev = float(np.abs(rec[3]["phone"][clear] - vq[clear]).max())
    print("phone (k-NN 3): max-abs %.3g on %d clear frames" % (dev, clear.sum()))
    assert dev <= 2e-5


def test_pitch_estimator_matches_independent_restatement(driven, model_dir):
    x, rec, _ = driven
    bins, feat, logits = sn.PitchEstimator(model_dir)(x, 1, 447)
    top2 = np.sort(logits[:, 1:448], axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-3
    assert clear.sum() >= HOPS // 2
    assert np.array_equal(rec[0]["q"][clear], bins[clear])
    same = rec[0]["q"] == bins
    prev_same = np.concatenate([[True], same[:-1]])
    for i, name in enumerate(("max-logit probability", "log energy", "bin delta", "voicing")):
        rows = same & prev_same if i == 2 else (same if i == 0 else np.ones(HOPS, bool))
        dev = float(np.abs(rec[0]["feat"][rows, i] - feat[rows, i]).max())
        print("pitch feature %d (%s): max-abs %.3g" % (i, name, dev))
        assert dev <= 2e-5
    assert len(set(bins.tolist())) > 1


def test_waveform_generator_matches_independent_restatement(driven, model_dir):
    """Fed with the ORACLE's phone / bin / features, so that only this module is compared.  The host applies the
    pitch transform between the two calls; with default settings it is the identity on the bin."""
    x, rec, tables = driven
    wg = sn.WaveformGenerator(model_dir)
    for k in (0, 3):
        r = rec[k]
        want = wg(r["phone"], r["q"], r["feat"], tables.additive[1], tables.formant[6], tables.kv[1])
        dev = float(np.abs(r["pcm"] - want).max())
        print("waveform (k=%d): max-abs %.3g, rms %.3g" % (k, dev, np.sqrt((want ** 2).mean())))
        assert np.sqrt((want ** 2).mean()) > 0.02
        assert dev <= 1e-4
