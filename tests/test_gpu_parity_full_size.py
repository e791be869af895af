"""Parity at BASELINE.json's full per-GPU size (256 concurrent streams) and on the inputs that stress the
discrete decisions of the path (silence: argmax ties; full-scale and DC: clamps and saturating tanh)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model_dir8(tmp_path_factory):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_model
    d = str(tmp_path_factory.mktemp("model8"))
    make_model.make_model(d, n_speakers=8)
    return d


def _oracle(bv, oracle, model_dir, audio, hops, setup, event):
    m = bv.Models(oracle, model_dir)
    B = audio.shape[0]
    out = np.zeros((hops, B, bv.OUT_HOP), np.float32)
    for s in range(B):  # stream after stream: the oracle streams are independent
        st = bv.Stream1(m, speaker=0, vq_k=0)
        setup(s, st, None)
        for h in range(hops):
            event(h, s, st, None)
            out[h, s] = st.hop(audio[s, h * 160:(h + 1) * 160])
        st.close()
    m.close()
    return out


def test_256_streams_rotating_speakers_knn(bv, oracle, product, model_dir8):
    """configs[2]/[3] shape on one GPU: 256 streams, 8 speakers, every stream switches speaker once (K/V blocks
    one per hop), k-NN codebook lookup on; all 256 streams compared with independent oracle streams."""
    B, hops = 256, 14
    audio = np.stack([bv.synth_audio(160 * hops, seed=500 + s) for s in range(B)])

    def setup(s, st, batch):
        if st is not None:
            st.set_target_speaker(s % 8)
            while st.set_kv_block():
                pass
            st.a.SetVQNumNeighbors(st.pc, 4 if s % 4 else 0)
        else:
            a, h = batch.a, batch.h
            a.BeatriceBatch_SetTargetSpeaker(h, s, s % 8)
            a.BeatriceBatch_FlushSpeaker(h, s)
            a.BeatriceBatch_SetVQNumNeighbors(h, s, 4 if s % 4 else 0)

    def event(h, s, st, batch):
        if h == 4 + (s % 5):
            spk = (s + 3) % 8
            if st is not None:
                st.set_target_speaker(spk)
            else:
                batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, spk)

    ref = _oracle(bv, oracle, model_dir8, audio, hops, setup, event)
    m = bv.Models(product, model_dir8)
    batch = bv.Batch(m, B)
    for s in range(B):
        setup(s, None, batch)
    got = np.zeros_like(ref)
    for h in range(hops):
        for s in range(B):
            event(h, s, None, batch)
        got[h] = batch.convert(audio[:, h * 160:(h + 1) * 160])
    batch.close()
    m.close()
    dev = float(np.abs(ref - got).max())
    print("B=256 rotating speakers + k-NN: max-abs %g %s" % (dev, "bit-identical" if np.array_equal(ref, got) else ""))
    assert np.abs(got).max() > 0.05
    assert dev <= TOL


@pytest.mark.parametrize("kind", ["silence", "full_scale_square", "dc", "impulses"])
def test_stress_inputs(bv, oracle, product, model_dir, kind):
    """Digital silence makes every logit of a frame depend on biases only (ties are broken towards the
    lowest index on both sides); full-scale square waves and DC exercise the clamps."""
    B, hops = 6, 20
    n = 160 * hops
    rng = np.random.Generator(np.random.PCG64(77))
    if kind == "silence":
        audio = np.zeros((B, n), np.float32)
        audio[1, 160 * 10:] = bv.synth_audio(n, seed=1)[160 * 10:]      # speech after silence
        audio[2, :160 * 8] = bv.synth_audio(n, seed=2)[:160 * 8]        # silence after speech
    elif kind == "full_scale_square":
        t = np.arange(n)
        audio = np.stack([np.where((t // (20 + 7 * s)) % 2 == 0, 1.0, -1.0) for s in range(B)]).astype(np.float32)
    elif kind == "dc":
        audio = np.stack([np.full(n, v, np.float32) for v in (1.0, -1.0, 0.5, 1e-3, -1e-6, 0.25)])
    else:
        audio = np.zeros((B, n), np.float32)
        for s in range(B):
            audio[s, rng.integers(0, n, 12)] = rng.choice([-1.0, 1.0], 12)

    def setup(s, st, batch):
        k = (0, 1, 8)[s % 3]
        if st is not None:
            st.a.SetVQNumNeighbors(st.pc, k)
        else:
            batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, k)

    ref = _oracle(bv, oracle, model_dir, audio, hops, setup, lambda *a: None)
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    for s in range(B):
        setup(s, None, batch)
    got = np.zeros_like(ref)
    q_trace = []
    for h in range(hops):
        got[h] = batch.convert(audio[:, h * 160:(h + 1) * 160])
        q_trace.append(batch.intermediates()[1].copy())
    batch.close()
    m.close()
    assert np.all(np.isfinite(got))
    dev = float(np.abs(ref - got).max())
    print("%s: max-abs %g %s; raw bins seen %s" % (kind, dev, "bit-identical" if np.array_equal(ref, got) else "",
                                                  sorted(set(np.concatenate(q_trace).tolist()))[:8]))
    assert dev <= TOL


def test_batch_and_shard_invariance(bv, product, model_dir):
    """A stream's output does not depend on the batch it runs in: the same 96 streams as one batch, as the three
    shards that `shard.stream_range` gives three ranks, and one of them alone (B = 1) -- bit for bit.  This is the
    single-GPU form of "sharded over N GPUs == one GPU" (streams never exchange anything)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("beatrice_shard", os.path.join(REPO, "beatrice-vst_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    B, hops = 96, 10
    audio = np.stack([bv.synth_audio(160 * hops, seed=700 + s) for s in range(B)])
    m = bv.Models(product, model_dir)

    def run(lo, hi):
        batch = bv.Batch(m, hi - lo)
        for s in range(lo, hi):
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s - lo, s % 3)
            batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s - lo, s % 4)
            batch.a.BeatriceBatch_SetPitchShift(batch.h, s - lo, float(s % 7) - 3.0)
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)
        out = np.stack([batch.convert(audio[lo:hi, h * 160:(h + 1) * 160]) for h in range(hops)])  # [hops][n][240]
        batch.close()
        return out

    whole = run(0, B)
    for rank in range(3):
        lo, hi = shard.stream_range(rank, 3, B)
        assert np.array_equal(run(lo, hi), whole[:, lo:hi]), "shard %d differs from the whole batch" % rank
    assert np.array_equal(run(41, 42), whole[:, 41:42])
    m.close()
    assert np.abs(whole).max() > 0.05


@pytest.mark.parametrize("B", [2048, 2091])
def test_large_batch_chain_matches_small_batches(bv, oracle, product, model_dir, B):
    """From 2048 streams on the in-order chain runs the conditioned blocks as row-local kernels (wave.hip); below that as
    six launches per block.  Same streams either way, bit for bit: 2048 streams in one batch against slices of them in
    batches of 300, with speakers and k-NN settings varied (per-speaker attention tiles) and a switch in mid-run."""
    hops = 6   # (2091: ragged last row tiles in every tiling -- 16-row block halves, 32-row convolutions, 64-row fall-backs)
    base = np.stack([bv.synth_audio(160 * hops, seed=5200 + s) for s in range(64)])
    audio = np.concatenate([np.roll(base, 37 * r, axis=1) * np.float32(1.0 - 0.01 * r) for r in range((B + 63) // 64)], axis=0)[:B]
    m = bv.Models(product, model_dir)

    def run(lo, hi):
        batch = bv.Batch(m, hi - lo)
        for s in range(lo, hi):
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s - lo, s % 3)
            batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s - lo, (s // 3) % 3)
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)
        out = []
        for h in range(hops):
            if h == 2:
                for s in range(lo, hi):
                    if s % 5 == 0:
                        batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s - lo, (s + 1) % 3)   # K/V blocks follow one per hop
            out.append(batch.convert(np.ascontiguousarray(audio[lo:hi, h * 160:(h + 1) * 160])))
        batch.close()
        return np.stack(out)

    whole = run(0, B)
    assert np.abs(whole).max() > 0.05
    for lo in (0, 900, B - 300):
        assert np.array_equal(run(lo, lo + 300), whole[:, lo:lo + 300]), "streams %d.. differ between the two chains" % lo
    m.close()
    # the ORACLE leg: a sample of the large batch's streams (tile corners, the ragged end, switching streams) as independent
    # oracle streams under the same settings and the same mid-run switch
    from oracle_batch import OracleBatch
    sample = [0, 5, 15, 16, 31, 32, 1000, 1023, 1024, B - 65, B - 33, B - 6, B - 1]
    sample = sorted(set(sample) | {s for s in range(B - 12, B) if s % 5 == 0})
    ob = OracleBatch(bv, oracle, model_dir, B, sample=sample)
    for s in ob.sample:
        ob.a.BeatriceBatch_SetTargetSpeaker(None, s, s % 3)
        ob.a.BeatriceBatch_SetVQNumNeighbors(None, s, (s // 3) % 3)
    ob.a.BeatriceBatch_FlushSpeaker(None, -1)
    dev = 0.0
    for h in range(hops):
        if h == 2:
            for s in ob.sample:
                if s % 5 == 0:
                    ob.a.BeatriceBatch_SetTargetSpeaker(None, s, (s + 1) % 3)
        want = ob.convert_rows(audio[:, h * 160:(h + 1) * 160])
        dev = max(dev, float(np.abs(whole[h][ob.sample] - want).max()))
    ob.close()
    print("B=%d in-order chain (row-local block kernels) vs ORACLE, streams %s: max-abs %g" % (B, ob.sample, dev))
    assert dev <= TOL
