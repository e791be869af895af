"""Real-time contract of the 1-stream ABI: the per-hop calls and the setters the reference
host issues on its audio thread (morph mode calls SetCodebook every hop, processor_core_2.cc:118-121) must not
allocate device memory, and a codebook table rewritten in place must not be served from a stale device copy."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model_dir8(tmp_path_factory):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_model
    d = str(tmp_path_factory.mktemp("model8rt"))
    make_model.make_model(d, n_speakers=8)
    return d


def _free_bytes():
    import torch
    return torch.cuda.mem_get_info()[0]


def test_setcodebook_per_hop_no_allocation_and_lru(bv, oracle, product, model_dir8):
    """14 distinct codebook pointers cycled hop by hop (more than the context's pool of 12, so slots are recycled):
    free device memory stays on a plateau once the contexts exist, and PCM equals the oracle's."""
    hops = 40
    x = bv.synth_audio(160 * hops, seed=31)
    outs, free_trace = {}, []
    for name, abi in (("oracle", oracle), ("hip", product)):
        m = bv.Models(abi, model_dir8)
        t = m.tables
        books = [t.codebooks[i] for i in range(9)] + [np.ascontiguousarray(t.codebooks[i % 8][::-1].copy()) for i in range(5)]
        st = bv.Stream1(m, speaker=1, vq_k=3)
        out = np.zeros((hops, bv.OUT_HOP), np.float32)
        for h in range(hops):
            st.a.SetCodebook(st.pc, bv.fptr(books[(h * 5) % len(books)]))
            if h % 9 == 4:   # the other audio-thread setters of morph mode
                st.a.SetAdditiveSpeakerEmbedding(m.embed, bv.fptr(t.additive[h % 8]), st.ec, st.wc)
                st.a.RegisterKeyValueSpeakerEmbedding(m.embed, bv.fptr(t.kv[h % 8]), st.ec)
                st.kv_count = 0
            out[h] = st.hop(x[h * 160:(h + 1) * 160])
            if name == "hip":
                free_trace.append(_free_bytes())
        st.close()
        m.close()
        outs[name] = out
    assert np.abs(outs["hip"]).max() > 0.01
    assert np.array_equal(outs["oracle"], outs["hip"]), "max-abs %g" % np.abs(outs["oracle"] - outs["hip"]).max()
    assert len(set(free_trace)) == 1, "free device memory moved during hops: %s" % sorted(set(free_trace))


def test_codebook_rewritten_in_place_is_reuploaded(bv, oracle, product, model_dir8):
    """Same host pointer, new contents (a host that reloads a model into the same storage): the next SetCodebook must
    pick the new table up."""
    hops = 12
    x = bv.synth_audio(160 * hops, seed=32)
    outs = {}
    for name, abi in (("oracle", oracle), ("hip", product)):
        m = bv.Models(abi, model_dir8)
        t = m.tables
        table = np.ascontiguousarray(t.codebooks[2].copy())
        st = bv.Stream1(m, speaker=0, vq_k=2)
        out = np.zeros((hops, bv.OUT_HOP), np.float32)
        for h in range(hops):
            if h == 6:
                table[:] = t.codebooks[5]
            st.a.SetCodebook(st.pc, bv.fptr(table))
            out[h] = st.hop(x[h * 160:(h + 1) * 160])
        st.close()
        m.close()
        outs[name] = out
    assert np.array_equal(outs["oracle"], outs["hip"])
    assert not np.array_equal(outs["hip"][5], outs["hip"][7])


def test_codebook_edited_outside_the_fingerprint_needs_invalidate(bv, oracle, product, model_dir8):
    """SetCodebook's identity test is address + a 96-word sample (a per-hop call cannot hash 256 KB): an in-place edit that
    touches none of the sampled words is served from the device copy until the caller says so with
    BeatriceHip_InvalidateCodebook (off the audio thread) -- after which the PCM equals the oracle's with the edited table."""
    import ctypes as C
    n = 512 * 128
    sampled = set(range(16)) | set(n - 1 - i for i in range(16)) | set((i * 1021 + 389) % n for i in range(64))
    hops = 10
    x = bv.synth_audio(160 * hops, seed=33)
    abi_b = bv.bind_batch(product)
    outs = {}
    for name, abi in (("oracle", oracle), ("hip", product)):
        m = bv.Models(abi, model_dir8)
        t = m.tables
        table = np.ascontiguousarray(t.codebooks[3].copy())
        other = t.codebooks[6].reshape(-1)
        st = bv.Stream1(m, speaker=0, vq_k=2)
        out = np.zeros((hops, bv.OUT_HOP), np.float32)
        for h in range(hops):
            if h == 5:   # every word the fingerprint does NOT look at takes the other speaker's value
                flat = table.reshape(-1)
                keep = {i: flat[i] for i in sampled}
                flat[:] = other
                for i, v in keep.items():
                    flat[i] = v
                if name == "hip":
                    abi_b.BeatriceHip_InvalidateCodebook(st.pc, table.ctypes.data_as(C.c_void_p))
            st.a.SetCodebook(st.pc, bv.fptr(table))
            out[h] = st.hop(x[h * 160:(h + 1) * 160])
        st.close()
        m.close()
        outs[name] = out
    assert np.array_equal(outs["oracle"], outs["hip"]), "max-abs %g" % np.abs(outs["oracle"] - outs["hip"]).max()
    assert not np.array_equal(outs["hip"][4], outs["hip"][6])


def test_reloading_parameters_into_the_same_model_objects(bv, oracle, product, model_dir, tmp_path):
    """Read*Parameters on model objects that contexts have already run with (the device blobs are freed and re-allocated):
    the contexts' captured hop graphs are keyed on the blob, so the next hop re-captures instead of replaying kernels that
    read freed memory, and a k-NN toggle afterwards finds its variant captured already.  Same calls on the oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_model
    d2 = str(tmp_path / "other")
    make_model.make_model(d2, n_speakers=3, seed=0x51C0)
    hops = 18
    x = bv.synth_audio(160 * hops, seed=77)
    outs = {}
    for name, abi in (("oracle", oracle), ("hip", product)):
        m = bv.Models(abi, model_dir)
        st = bv.Stream1(m, speaker=1, vq_k=2)
        out = []
        for h in range(hops):
            if h == 6:   # new parameters into the SAME objects; tables and conditioning re-applied as a host does after LoadModel
                for obj, fn, fname in ((m.phone, abi.ReadPhoneExtractorParameters, "phone_extractor.bin"),
                                       (m.pitch, abi.ReadPitchEstimatorParameters, "pitch_estimator.bin"),
                                       (m.wave, abi.ReadWaveformGeneratorParameters, "waveform_generator.bin"),
                                       (m.embed, abi.ReadEmbeddingSetterParameters, "embedding_setter.bin")):
                    assert fn(obj, os.path.join(d2, fname).encode()) == 0
                m.tables = bv.SpeakerTables(abi, d2)
                st.set_target_speaker(2)
                while st.set_kv_block():
                    pass
                st.set_formant_index(4)
            if h == 11:
                abi.SetVQNumNeighbors(st.pc, 0)     # the other graph variant
            if h == 14:
                abi.SetVQNumNeighbors(st.pc, 3)
            out.append(st.hop(x[h * 160:(h + 1) * 160]))
        outs[name] = np.stack(out)
        st.close()
        m.close()
    dev = float(np.abs(outs["hip"] - outs["oracle"]).max())
    print("parameters reloaded into live model objects: max-abs %g" % dev)
    assert np.abs(outs["hip"][7:]).max() > 1e-3
    assert dev <= 1e-4


def test_team_launch_timeout_recovers(bv, oracle, product, model_dir):
    """The 1-stream calls run each module's convolutions as ONE launch of a team of workgroups that wait for each other; where the
    team cannot become resident its bounded waits give up.  The call that hit the timeout returns zeros (beatrice.h's contract for
    any internal failure), the context's rings restart from silence and every later call runs one launch per layer -- it does not
    stay silent for good.  The waveform generator has a finite memory, so a few hops after the (injected) timeout the
    samples equal those of an uninterrupted oracle stream bit for bit again."""
    bv.bind_batch(product)
    hops, at = 70, 9
    x = bv.synth_audio(160 * hops, seed=77)
    mo = bv.Models(oracle, model_dir)
    so = bv.Stream1(mo, speaker=1, vq_k=2)
    want = np.array([so.hop(x[i * 160:(i + 1) * 160]) for i in range(hops)])
    so.close(); mo.close()
    m = bv.Models(product, model_dir)
    s = bv.Stream1(m, speaker=1, vq_k=2)
    got = []
    for i in range(hops):
        if i == at:
            assert product.BeatriceHip_InjectTeamTimeout(s.wc) == 0
        got.append(s.hop(x[i * 160:(i + 1) * 160]))
    assert product.BeatriceHip_InjectTeamTimeout(s.wc) == -1      # the context has left the team launch for good
    s.close(); m.close()
    got = np.array(got)
    assert np.array_equal(got[:at], want[:at])
    assert not got[at].any()                                       # the failed call: zeros
    assert np.abs(got[at + 1:at + 4]).max() > 0                    # ... and sound again right after
    settled = at + 40                                              # (dilated convolutions reach 16 frames back, the tail a few more)
    assert np.array_equal(got[settled:], want[settled:]), "max-abs %g" % np.abs(got[settled:] - want[settled:]).max()


@pytest.mark.parametrize("which", ["phone", "pitch"])
def test_team_launch_timeout_recovers_in_the_other_modules(bv, oracle, product, model_dir, which):
    """The same recovery in ExtractPhone1 and EstimatePitch1 (only the waveform context had an injection test): the call that hit
    the (injected) timeout returns zeros, the context restarts FROM SILENCE -- rings, GRU state, the pitch estimator's previous bin -- and runs one
    launch per layer from then on.  Both modules remember for ever (a GRU), so the yardstick after the timeout is a NEW oracle context that starts
    with the hop after it: bit for bit from its first hop on."""
    bv.bind_batch(product)
    hops, at = 30, 8
    x = bv.synth_audio(160 * hops, seed=78)

    def run(abi, models, inject):
        pc, tc = abi.CreatePhoneContext1(), abi.CreatePitchContext1()
        t = models.tables
        abi.SetCodebook(pc, bv.fptr(t.codebooks[1]))
        abi.SetVQNumNeighbors(pc, 2)
        abi.SetMinQuantizedPitch(tc, 1)
        abi.SetMaxQuantizedPitch(tc, 383)
        out = []
        for i in range(hops):
            if inject is not None and i == inject:
                if which == "phone":
                    assert product.BeatriceHip_InjectTeamTimeoutPhone(pc) == 0
                else:
                    assert product.BeatriceHip_InjectTeamTimeoutPitch(tc) == 0
            h = np.ascontiguousarray(x[i * 160:(i + 1) * 160])
            if which == "phone":
                v = np.zeros(bv.PHONE_CH, np.float32)
                abi.ExtractPhone1(models.phone, bv.fptr(h), bv.fptr(v), pc)
                out.append(v)
            else:
                q, f = np.zeros(1, np.int32), np.zeros(4, np.float32)
                abi.EstimatePitch1(models.pitch, bv.fptr(h), bv.iptr(q), bv.fptr(f), tc)
                out.append(np.concatenate([q.astype(np.float32), f]))
        if inject is not None:
            assert (product.BeatriceHip_InjectTeamTimeoutPhone(pc) if which == "phone" else product.BeatriceHip_InjectTeamTimeoutPitch(tc)) == -1   # per-layer launches for good
        abi.DestroyPhoneContext1(pc)
        abi.DestroyPitchContext1(tc)
        return np.array(out)

    m = bv.Models(product, model_dir)
    got = run(product, m, at)
    m.close()
    mo = bv.Models(oracle, model_dir)
    want_before = run(oracle, mo, None)
    x_keep, x = x, x[160 * (at + 1):]
    hops_keep, hops = hops, hops - at - 1
    want_after = run(oracle, mo, None)          # a new context that starts with the hop after the timeout
    x, hops = x_keep, hops_keep
    mo.close()
    assert np.array_equal(got[:at], want_before[:at])
    if which == "phone":
        assert not got[at].any()                                    # the failed call: zeros
    else:
        assert got[at][0] == 1.0 and not got[at][1:].any()           # ... the pitch estimator's: the lowest valid bin, zero features (csrc/abi.hip)
    assert np.abs(got[at + 1:]).max() > 0
    assert np.array_equal(got[at + 1:], want_after), "max-abs %g" % np.abs(got[at + 1:] - want_after).max()


def test_team_launch_timeout_in_a_one_stream_batch(bv, oracle, product, model_dir):
    """A batch of ONE stream runs its in-order chain on the modules' team launches; a timeout there voids the steps since the last
    synchronisation (ConvertFrames reports -2 once and hands out zeros), the module whose team gave up -- here the waveform generator's -- restarts
    from silence on the per-layer launches, the other modules' state stays, and the batch stays usable: once the waveform generator's finite memory
    (dilated convolutions, the tail's histories) has passed, the samples equal the UNINTERRUPTED oracle stream's bit for bit again."""
    a = bv.bind_batch(product)
    hops, at = 90, 6
    x = bv.synth_audio(160 * hops, seed=79)
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, 1)
    got = []
    for i in range(hops):
        if i == at:
            rc = a.BeatriceBatch_InjectTeamTimeout(batch.h)
            if rc == -1:
                pytest.skip("this device runs the one-stream batch without team launches")
            out = np.zeros((1, 240), np.float32)
            assert a.BeatriceBatch_ConvertFrames(batch.h, bv.fptr(np.ascontiguousarray(x[i * 160:(i + 1) * 160])), bv.fptr(out)) == -2
            assert not out.any()
            got.append(out[0])
            continue
        got.append(batch.convert(x[None, i * 160:(i + 1) * 160])[0])
    assert a.BeatriceBatch_InjectTeamTimeout(batch.h) == -1
    assert a.BeatriceBatch_IsHealthy(batch.h)
    batch.close(); m.close()
    got = np.array(got)
    mo = bv.Models(oracle, model_dir)
    so = bv.Stream1(mo, speaker=0)
    want = np.array([so.hop(x[i * 160:(i + 1) * 160]) for i in range(hops)])
    so.close(); mo.close()
    assert np.array_equal(got[:at], want[:at])
    assert np.abs(got[at + 1:at + 4]).max() > 0                    # sound again right after
    settled = at + 45
    assert np.array_equal(got[settled:], want[settled:]), "max-abs %g" % np.abs(got[settled:] - want[settled:]).max()
