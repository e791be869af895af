"""The throughput mode the bench times (tick pipelining over resident I/O) against the ORACLE directly -- not against the
in-order HIP chain -- on the bench's own workload and over runs long enough for every ring of the pipeline to wrap many
times (the deepest ring of a stage boundary holds ~30 step slots; the resident I/O ring 32-64)."""
import numpy as np
import pytest

from oracle_batch import oracle_leg
from tick_driver import run_tick

pytestmark = pytest.mark.gpu
TOL = 1e-4   # north_star's bound on output PCM; observed 0.0


@pytest.mark.parametrize("H", [4, 2, 1])
def test_bench_workload_in_tick_mode_matches_oracle(bv, oracle, product, model_dir, H):
    """bench.py's headline: BASELINE.json configs[2] -- 256 streams, 1 speaker, k-NN 0, every tenth stream with 0.5 s of
    digital silence, 64 resident steps per stream cycled as I/O slots, steps of H hops (4: the bench's default,
    `bench.DEFAULT_HOPS_PER_STEP_TICK`; tests/test_cpu_bench_defaults_are_tested.py keeps test and bench in step) enqueued
    without waiting.  100 steps (the slot ring wraps), a sample of 12 streams (silence-gap streams 3, 13, 253 among them)
    against independent oracle streams."""
    B, n_cycle, steps = 256, 64, 100
    audio = np.stack([bv.synth_audio(160 * H * n_cycle, seed=s, silence_gap=(s % 10 == 3)) for s in range(B)]).reshape(B, n_cycle * H, 160)

    def settings(batch):
        batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, -1, 0)
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)

    def hop_input(j):   # hop j of the run
        return audio[:, j % (n_cycle * H)]

    def step_input(k):
        return np.concatenate([hop_input(k * H + hh) for hh in range(H)], axis=1)

    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B, hops_per_step=H)
    settings(batch)
    got = run_tick(bv, batch, steps, step_input, slots=n_cycle)
    batch.close()
    m.close()
    assert np.all(np.abs(got).reshape(steps, B, -1).max(axis=(0, 2)) > 1e-3)     # every stream produces sound
    sample = [0, 3, 13, 15, 16, 31, 32, 127, 128, 253, 254, 255]
    sample, want = oracle_leg(bv, oracle, model_dir, B, hop_input, steps * H, settings, lambda b, k: None, sample)
    want = want.reshape(steps, H, len(sample), 240).transpose(0, 2, 1, 3).reshape(steps, len(sample), H * 240)
    dev = float(np.abs(got[:, sample] - want).max())
    print("bench workload, tick mode (%d hop(s) per step) vs ORACLE: %d streams x %d steps, max-abs %g %s"
          % (H, len(sample), steps, dev, "bit-identical" if np.array_equal(got[:, sample], want) else ""))
    assert dev <= TOL


@pytest.mark.parametrize("B,steps,H", [(8, 330, 4), (8, 330, 1), (8, 330, 2)])
def test_tick_soak_vs_oracle(bv, oracle, product, model_dir, B, steps, H):
    """Soak: every stream of a small batch for 330 hops in tick mode against independent oracle streams driven through the
    reference protocol (processor_core_2.cc:50-256; a switch installs one K/V block per hop, :179-181): speaker switches
    throughout, k-NN on / off / changed, formant and pitch settings, pitch range, two stream resets -- early, in the middle and
    near the end, so that every ring of the pipeline has wrapped several times when they arrive."""
    # (H = 2 / 4: that many hops per stage per launch -- 4 is the bench's default --; the script then precedes a STEP, i.e. every H-th hop of
    #  the oracle streams)
    audio = np.stack([bv.synth_audio(160 * H * steps, seed=7700 + s, silence_gap=(s == 5)) for s in range(B)]).reshape(B, steps * H, 160)

    def settings(batch):
        for s in range(B):
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % 3)
            batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, (0, 1, 4)[s % 3])
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)

    def change(batch, k):
        a, h = batch.a, batch.h
        if k % 23 == 7:
            a.BeatriceBatch_SetTargetSpeaker(h, (k // 23) % B, (k // 23 + 1) % 3)      # K/V blocks follow, one per hop
        if k % 31 == 11:
            a.BeatriceBatch_SetVQNumNeighbors(h, (k // 31 + 2) % B, (k // 31) % 9)    # 0 turns the codebook off
        if k % 41 == 13:
            a.BeatriceBatch_SetFormantShift(h, (k // 41 + 3) % B, float(k % 5) - 2.0)
            a.BeatriceBatch_SetPitchShift(h, (k // 41 + 4) % B, float(k % 9) - 4.0)
        if k in (61, 211):
            assert a.BeatriceBatch_ResetStream(h, 6 if k == 61 else 1) == 0
        if k == 150:
            a.BeatriceBatch_SetMinSourcePitch(h, 0, 48.0)
            a.BeatriceBatch_SetMaxSourcePitch(h, 0, 70.0)
            a.BeatriceBatch_SetPitchCorrection(h, 2, 0.6)
            a.BeatriceBatch_SetPitchCorrectionType(h, 2, 1)
        if k == 300:
            a.BeatriceBatch_SetTargetSpeaker(h, -1, 2)                                # everyone, near the end

    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B, hops_per_step=H)
    settings(batch)
    got = run_tick(bv, batch, steps, lambda k: audio[:, k * H:(k + 1) * H].reshape(B, H * 160), change, chunk=29)          # (chunks of 29: drains land everywhere)
    batch.close()
    m.close()
    sample, want = oracle_leg(bv, oracle, model_dir, B, lambda j: audio[:, j], steps * H, settings,
                              lambda ob, j: change(ob, j // H) if j % H == 0 else None, list(range(B)))
    want = want.reshape(steps, H, B, 240).transpose(0, 2, 1, 3).reshape(steps, B, H * 240)
    bad = sorted({int(k) for k in np.nonzero(np.abs(got - want).reshape(steps, -1).max(axis=1) > TOL)[0]})
    dev = float(np.abs(got - want).max())
    print("tick soak vs ORACLE: %d streams x %d steps, max-abs %g %s" % (B, steps, dev, "bit-identical" if np.array_equal(got, want) else ""))
    assert np.abs(got).max() > 0.05
    assert not bad, "steps beyond tolerance: %s" % bad[:20]


def test_ragged_speaker_groups_in_tick_mode_match_oracle(bv, oracle, product, tmp_path):
    """The attention half's row lists in tick mode (batch.hip rebuild_tiles): a K/V slot's rows fill whole 16-row tiles, a
    remainder of >= 8 rows is one padded tile, a smaller one goes to pairs of quads (rowchain.hip.h block_bq_body).  53 streams
    on 7 speakers in groups of 17, 9, 8, 7, 5, 4, 2 and 1 rows -- a full tile + a single row, padded tiles of 9 and 8, quads of
    4 + 3, 4 + 1, 4, 2 and 1 rows, a pair with an empty second quad -- with speaker switches on the way that move rows between
    the lists (K/V blocks one per hop: blocks of one stream sit on different slots meanwhile); every stream against the oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_model
    model_dir = str(tmp_path / "m8")
    os.makedirs(model_dir)
    make_model.make_model(model_dir, n_speakers=8)
    groups = [17, 9, 8, 7, 5, 4, 2, 1]
    speaker_of = [g for g, n in enumerate(groups) for _ in range(n)]
    B, steps = len(speaker_of), 34
    audio = np.stack([bv.synth_audio(160 * steps, seed=5100 + s) for s in range(B)]).reshape(B, steps, 160)

    def settings(batch):
        for s in range(B):
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, speaker_of[s])
            batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, s % 3)
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)

    def change(batch, k):
        a, h = batch.a, batch.h
        if k == 6:      # the single row of speaker 7 joins speaker 0 (17 -> 18 rows: one tile + 2 rows as a quad)
            a.BeatriceBatch_SetTargetSpeaker(h, B - 1, 0)
        if k == 13:     # three rows leave the group of nine (9 -> 6: from a padded tile to quads of 4 + 2)
            for s in (17, 18, 19):
                a.BeatriceBatch_SetTargetSpeaker(h, s, 6)
        if k == 21:     # everybody of speaker 3 to speaker 2 (8 + 7 = 15: a padded tile)
            for s in range(B):
                if speaker_of[s] == 3:
                    a.BeatriceBatch_SetTargetSpeaker(h, s, 2)

    def hop_input(k):
        return audio[:, k]

    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    settings(batch)
    got = run_tick(bv, batch, steps, hop_input, change=change)
    batch.close()
    m.close()
    sample, want = oracle_leg(bv, oracle, model_dir, B, hop_input, steps, settings, change, list(range(B)))
    assert np.abs(got).max() > 1e-2
    bad = [(k, s) for k in range(steps) for s in range(B) if not np.array_equal(got[k, s], want[k, s])]
    assert not bad, "first differing (step, stream): %s of %d, max-abs %g" % (bad[:4], len(bad), np.abs(got - want).max())
