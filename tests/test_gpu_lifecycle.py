"""Object lifecycle: contexts and batches can be created and destroyed for ever
without leaking device memory (the reference host destroys and re-creates its four contexts on every
ResetContext, processor_core_2.cc:258-266)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_bytes():
    """hipMemGetInfo through the HIP runtime the product library itself is linked against"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipDeviceSynchronize() == 0
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_contexts_do_not_leak(bv, product, model_dir):
    m = bv.Models(product, model_dir)
    x = bv.synth_audio(160 * 2, seed=3)

    def cycle():
        s = bv.Stream1(m, speaker=1, vq_k=2)
        for h in range(2):
            s.hop(x[h * 160:(h + 1) * 160])
        s.close()

    for _ in range(3):
        cycle()  # warm up allocator pools and the codebook cache
    before = _free_bytes()
    for _ in range(40):
        cycle()
    after = _free_bytes()
    m.close()
    print("free device memory: %.1f MB -> %.1f MB" % (before / 2**20, after / 2**20))
    assert before - after <= 8 << 20


def test_batches_do_not_leak(bv, product, model_dir):
    m = bv.Models(product, model_dir)
    n = m.tables.n_speakers
    w = np.array([0.5, 0.3, 0.2], np.float32)

    def cycle(i):
        batch = bv.Batch(m, 48, hops_per_step=(1, 2, 4, 8)[i % 4])
        assert batch.a.BeatriceBatch_MorphSpeaker(batch.h, n, bv.fptr(w), n, i) == 0
        batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, 0, n)
        x = np.zeros((48, batch.H * 160), np.float32)
        batch.convert(x)
        batch.convert(x)
        batch.close()

    for i in range(4):
        cycle(i)
    before = _free_bytes()
    for i in range(24):
        cycle(i)
    after = _free_bytes()
    m.close()
    print("free device memory: %.1f MB -> %.1f MB" % (before / 2**20, after / 2**20))
    assert before - after <= 8 << 20


def test_wrapper_bindings_around_the_ticks_do_not_leak(bv, product, model_dir):
    """The resident-block wrappers around the tick pipeline (uniform clocks at 1 / 2 / 4 hops per step, clocks per stream): bound, run,
    unbound or destroyed while still bound, for ever."""
    import ctypes as C
    from tick_driver import Hip
    m = bv.Models(product, model_dir)
    hip = Hip()
    B, block = 8, 441

    def cycle(i):
        H = (1, 2, 4, 1)[i % 4]
        ragged = i % 4 == 3
        batch = bv.Batch(m, B, hops_per_step=H)
        a, h = batch.a, batch.h
        if ragged:
            assert a.BeatriceBatch_ConfigureWrapperRates(h, (C.c_double * B)(*([44100.0, 48000.0] * (B // 2)))) == 0
            slots = a.BeatriceBatch_TickStages(h) + 4
        else:
            assert a.BeatriceBatch_ConfigureWrapper(h, 44100.0) == 0
            slots = a.BeatriceBatch_ResidentBlocksDelayFor(h, block) + 4
        d_in, d_out = hip.malloc(slots * B * 480 * 4), hip.malloc(slots * B * 480 * 4)
        assert hip.lib.hipMemset(d_in, 0, C.c_size_t(slots * B * 480 * 4)) == 0
        if ragged:
            assert a.BeatriceBatch_BindResidentBlocksRagged(h, d_in, d_out, 1, 480, slots) == 0
            for _ in range(6):
                assert a.BeatriceBatch_ProcessBlocksRaggedDevice(h, (C.c_int * B)(*([441, 480] * (B // 2)))) == 0
        else:
            assert a.BeatriceBatch_BindResidentBlocks(h, d_in, d_out, 1, block, slots) == 0
            for _ in range(6):
                assert a.BeatriceBatch_ProcessBlocksDevice(h, None, None, 1, block) == 0
        assert a.BeatriceBatch_Synchronize(h) == 0
        if i % 2 == 0:   # every other batch is destroyed while still bound
            assert (a.BeatriceBatch_BindResidentBlocksRagged(h, None, None, 0, 0, 0) if ragged else a.BeatriceBatch_BindResidentBlocks(h, None, None, 0, 0, 0)) == 0
        batch.close()
        hip.free(d_in)
        hip.free(d_out)

    for i in range(4):
        cycle(i)
    before = _free_bytes()
    for i in range(16):
        cycle(i)
    after = _free_bytes()
    m.close()
    print("free device memory: %.1f MB -> %.1f MB" % (before / 2**20, after / 2**20))
    assert before - after <= 8 << 20


def test_unhealthy_objects_fail_softly(bv, product, model_dir):
    """Calls on objects that could not be built must not crash: error code from the batch, zeros from the
    void per-hop calls ('Errors')."""
    a = bv.bind_batch(product)
    empty_phone, empty_pitch = a.CreatePhoneExtractor(), a.CreatePitchEstimator()      # never loaded
    empty_wave, empty_embed = a.CreateWaveformGenerator(), a.CreateEmbeddingSetter()
    b = a.BeatriceBatch_Create(empty_phone, empty_pitch, empty_wave, empty_embed, 4, 2)
    assert a.BeatriceBatch_IsHealthy(b) == 0
    x = np.ones((4, 160), np.float32)
    y = np.ones((4, 240), np.float32)
    assert a.BeatriceBatch_ConvertFrames(b, bv.fptr(x), bv.fptr(y)) == -2
    assert a.BeatriceBatch_SetTargetSpeaker(b, 0, 0) == -2
    a.BeatriceBatch_Destroy(b)
    pc = a.CreatePhoneContext1()
    out = np.ones(bv.PHONE_CH, np.float32)
    a.ExtractPhone1(empty_phone, bv.fptr(x[0]), bv.fptr(out), pc)   # model not loaded: silence, no crash
    assert not out.any()
    a.DestroyPhoneContext1(pc)
    a.DestroyPhoneExtractor(empty_phone); a.DestroyPitchEstimator(empty_pitch)
    a.DestroyWaveformGenerator(empty_wave); a.DestroyEmbeddingSetter(empty_embed)
