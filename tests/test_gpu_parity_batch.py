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


def test_full_size_batch_invariance(bv, oracle, product, model_dir):
    """BASELINE.json configs[2] size (256 streams): a stream inside the full batch equals the same stream
    run alone through the oracle (checked on a sample of streams), and every stream produces sound."""
    B, hops = 256, 10
    audio = np.stack([bv.synth_audio(160 * hops, seed=500 + s) for s in range(B)])
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    for s in range(B):
        batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % 3)
        batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, (s // 3) % 9)
    batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)
    got = np.stack([batch.convert(audio[:, h * 160:(h + 1) * 160]) for h in range(hops)])
    batch.close()
    m.close()
    assert np.all(np.abs(got).reshape(hops, B, -1).max(axis=(0, 2)) > 1e-3)
    mo = bv.Models(oracle, model_dir)
    for s in (0, 1, 15, 16, 17, 100, 254, 255):
        st = bv.Stream1(mo, speaker=s % 3, vq_k=(s // 3) % 9)
        want = np.stack([st.hop(audio[s, h * 160:(h + 1) * 160]) for h in range(hops)])
        st.close()
        assert np.array_equal(got[:, s], want), "stream %d max-abs %g" % (s, np.abs(got[:, s] - want).max())
    mo.close()


def test_pitch_range_and_correction_settings(bv, oracle, product, model_dir):
    """Per-stream pitch search range (SetMin/MaxSourcePitch -> bins, processor_core_2.cc:561-583) and the
    pitch-correction branch of the transform (:199-249, uses pow) on the device vs the host."""
    B, hops = 6, 12
    audio = np.stack([bv.synth_audio(160 * hops, seed=700 + s) for s in range(B)])
    ranges = [(33.125, 80.875), (45.0, 60.0), (60.0, 45.0), (0.0, 128.0), (50.0, 50.0), (70.0, 88.0)]
    corr = [(0.0, 0), (0.5, 0), (0.5, 1), (1.0, 1), (0.25, 0), (0.9, 1)]

    def to_bin(note):
        note = min(max(note, 0.0), 128.0)
        r = (note - 33.0) * 8.0
        q = int(np.floor(r + 0.5)) if r >= 0 else -int(np.floor(-r + 0.5))
        return min(max(q, 1), 447)

    mo = bv.Models(oracle, model_dir)
    ref = np.zeros((hops, B, bv.OUT_HOP), np.float32)
    ref_q = np.zeros((hops, B), np.int32)
    for s in range(B):
        st = bv.Stream1(mo, speaker=1, min_q=to_bin(ranges[s][0]), max_q=to_bin(ranges[s][1]))
        st.pitch_params = dict(correction=corr[s][0], ctype=corr[s][1], shift=0.5 * s)
        for h in range(hops):
            o, _, _, _, q2 = st.hop(audio[s, h * 160:(h + 1) * 160], return_all=True)
            ref[h, s], ref_q[h, s] = o, q2
        st.close()
    mo.close()
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    a, hnd = batch.a, batch.h
    a.BeatriceBatch_SetTargetSpeaker(hnd, -1, 1)
    a.BeatriceBatch_FlushSpeaker(hnd, -1)
    for s in range(B):
        a.BeatriceBatch_SetMinSourcePitch(hnd, s, ranges[s][0])
        a.BeatriceBatch_SetMaxSourcePitch(hnd, s, ranges[s][1])
        a.BeatriceBatch_SetPitchCorrection(hnd, s, corr[s][0])
        a.BeatriceBatch_SetPitchCorrectionType(hnd, s, corr[s][1])
        a.BeatriceBatch_SetPitchShift(hnd, s, 0.5 * s)
    got = np.zeros_like(ref)
    got_q = np.zeros_like(ref_q)
    for h in range(hops):
        got[h] = batch.convert(audio[:, h * 160:(h + 1) * 160])
        got_q[h] = batch.intermediates()[2]
    batch.close()
    m.close()
    assert np.array_equal(got_q, ref_q)
    assert float(np.abs(got - ref).max()) <= TOL


@pytest.mark.parametrize("B,H,graph", [(3, 2, 1), (21, 4, 1), (5, 4, 0), (9, 8, 1)])
def test_block_mode_matches_single_hops(bv, oracle, product, model_dir, B, H, graph):
    """BeatriceBatch_CreateBlock: H hops per step == H single-hop steps of independent oracle streams,
    including a speaker switch whose four K/V blocks install on consecutive hops INSIDE a step."""
    hops = 32
    audio = np.stack([bv.synth_audio(160 * hops, seed=300 + s) for s in range(B)])

    def script(h, s, st, batch):  # settings change between steps only: every event hop is a multiple of 8
        if h == 0:
            if st is not None:
                st.a.SetVQNumNeighbors(st.pc, (s + 1) % 3)
                st.pitch_params = dict(shift=float(s % 3) - 1.0, correction=0.5 if s % 2 else 0.0)
            else:
                a, hnd = batch.a, batch.h
                a.BeatriceBatch_SetVQNumNeighbors(hnd, s, (s + 1) % 3)
                a.BeatriceBatch_SetPitchShift(hnd, s, float(s % 3) - 1.0)
                a.BeatriceBatch_SetPitchCorrection(hnd, s, 0.5 if s % 2 else 0.0)
        if h == 8 + 8 * (s % 3):
            spk = 1 + (s % 2)
            if st is not None:
                st.set_target_speaker(spk)
            else:
                batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, spk)
        if h == 16 and s % 2 == 1:
            if st is not None:
                st.set_formant_index(2)
            else:
                batch.a.BeatriceBatch_SetFormantShift(batch.h, s, -1.0)

    ref = _oracle_streams(bv, oracle, model_dir, B, hops, audio, script)  # [hops][B][240]
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B, hops_per_step=H)
    assert batch.a.BeatriceBatch_HopsPerStep(batch.h) == H
    batch.a.BeatriceBatch_EnableGraph(batch.h, graph)
    got = np.zeros_like(ref)
    for step in range(hops // H):
        for h in range(step * H, (step + 1) * H):
            for s in range(B):
                script(h, s, None, batch)
        y = batch.convert(audio[:, step * H * 160:(step + 1) * H * 160])  # [B][H*240]
        got[step * H:(step + 1) * H] = y.reshape(B, H, bv.OUT_HOP).transpose(1, 0, 2)
    phone, q_raw, q, feat = batch.intermediates()
    assert phone.shape == (B, H, bv.PHONE_CH) and q.shape == (B, H)
    # the 48 kHz wrapper is per 10 ms block: refused in block mode
    z = np.zeros((B, 1, 480), np.float32)
    assert batch.a.BeatriceBatch_ConvertBlocks48k(batch.h, bv.fptr(z), bv.fptr(z.copy()), 1) == -1
    batch.close()
    m.close()
    dev = float(np.abs(ref - got).max())
    print("B=%d H=%d graph=%d max-abs %g %s" % (B, H, graph, dev, "bit-identical" if np.array_equal(ref, got) else ""))
    assert np.abs(got).max() > 0.05
    assert dev <= TOL
