"""Batched extension ABI vs B independent oracle streams (include/beatrice_batch.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _oracle_streams(bv, oracle, model_dir, B, hops, audio, script):
    m = bv.Models(oracle, model_dir)
    streams = [bv.Stream1(m, speaker=0, vq_k=0) for _ in range(B)]
    out = np.zeros((hops, B, bv.OUT_HOP), np.float32)
    for h in range(hops):
        for s in range(B):
            script(h, s, streams[s], None)
            out[h, s] = streams[s].hop(audio[s, h * 160:(h + 1) * 160])
    for st in streams:
        st.close()
    m.close()
    return out


@pytest.mark.parametrize("B,graph", [(1, 1), (5, 0), (37, 1)])
def test_batch_matches_independent_streams(bv, oracle, product, model_dir, B, graph):
    hops = 24
    audio = np.stack([bv.synth_audio(160 * hops, seed=100 + s) for s in range(B)])

    def script(h, s, st, batch):
        # per-stream events: speaker rotation, formant change, VQ on some streams, pitch params
        if h == 0:
            if st is not None:
                st.a.SetVQNumNeighbors(st.pc, s % 3)
                st.pitch_params = dict(shift=float(s % 5) - 2.0, intonation=1.0 + 0.1 * (s % 4))
            else:
                a, hnd = batch.a, batch.h
                a.BeatriceBatch_SetVQNumNeighbors(hnd, s, s % 3)
                a.BeatriceBatch_SetPitchShift(hnd, s, float(s % 5) - 2.0)
                a.BeatriceBatch_SetIntonationIntensity(hnd, s, 1.0 + 0.1 * (s % 4))
        if h == 6 + (s % 7):
            spk = 1 + (s % 2)
            if st is not None:
                st.set_target_speaker(spk)
            else:
                batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, spk)
        if h == 15 and s % 2 == 0:
            if st is not None:
                st.set_formant_index(6)
            else:
                batch.a.BeatriceBatch_SetFormantShift(batch.h, s, 1.0)

    ref = _oracle_streams(bv, oracle, model_dir, B, hops, audio, script)
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    batch.a.BeatriceBatch_EnableGraph(batch.h, graph)
    got = np.zeros_like(ref)
    for h in range(hops):
        for s in range(B):
            script(h, s, None, batch)
        got[h] = batch.convert(audio[:, h * 160:(h + 1) * 160])
    batch.close()
    m.close()
    dev = float(np.abs(ref - got).max())
    print("B=%d graph=%d max-abs %g %s" % (B, graph, dev, "bit-identical" if np.array_equal(ref, got) else ""))
    assert np.abs(got).max() > 0.05
    assert dev <= TOL


def test_batch_reset_stream(bv, oracle, product, model_dir):
    """ResetStream == destroying and recreating the contexts of that stream (processor_core_2.cc:258-291)."""
    B, hops = 4, 16
    audio = np.stack([bv.synth_audio(160 * hops, seed=200 + s) for s in range(B)])
    mo = bv.Models(oracle, model_dir)
    streams = [bv.Stream1(mo, speaker=1) for _ in range(B)]
    ref = np.zeros((hops, B, bv.OUT_HOP), np.float32)
    for h in range(hops):
        if h == 9:
            streams[2].close()
            streams[2] = bv.Stream1(mo, speaker=1)
        for s in range(B):
            ref[h, s] = streams[s].hop(audio[s, h * 160:(h + 1) * 160])
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, -1, 1)
    batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)
    got = np.zeros_like(ref)
    for h in range(hops):
        if h == 9:
            assert batch.a.BeatriceBatch_ResetStream(batch.h, 2) == 0
        got[h] = batch.convert(audio[:, h * 160:(h + 1) * 160])
    batch.close()
    dev = float(np.abs(ref - got).max())
    print("reset max-abs", dev)
    assert dev <= TOL
