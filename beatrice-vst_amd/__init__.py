"""beatrice-vst_amd -- MI355X-native Beatrice 2 (rc.0) per-hop voice-conversion path.

The product is the C-ABI shared library ``csrc/libbeatrice_hip.so`` (hand-written HIP for gfx950,
declared in ``include/beatrice_abi.h`` + ``include/beatrice_batch.h``).  This Python module is
plumbing only: ctypes prototypes for that ABI, used by the tests, ``bench.py`` and
``__graft_entry__.py``.  The directory name contains a hyphen, so import it by path:

    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("beatrice_vst_amd", ".../beatrice-vst_amd/__init__.py")

There is deliberately NO CPU fallback here: ``load_product()`` raises if the HIP library is
missing or cannot be loaded.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
# (BEATRICE_HIP_LIB: another build of the same library, for A/B measurements of kernel variants)
PRODUCT_LIB = os.environ.get("BEATRICE_HIP_LIB") or os.path.join(HERE, "csrc", "libbeatrice_hip.so")

IN_HOP, OUT_HOP = 160, 240
PHONE_CH, HID, PITCH_BINS = 128, 256, 448
CODEBOOK, KV_LEN, KV_CH, N_BLOCKS = 512, 384, 128, 4

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)
_vp = C.c_void_p

# name -> (restype, argtypes) for the rc0 generation (include/beatrice_abi.h)
_RC0 = {
    "CreatePhoneExtractor": (_vp, []), "DestroyPhoneExtractor": (None, [_vp]),
    "CreatePhoneContext1": (_vp, []), "DestroyPhoneContext1": (None, [_vp]),
    "CreatePitchEstimator": (_vp, []), "DestroyPitchEstimator": (None, [_vp]),
    "CreatePitchContext1": (_vp, []), "DestroyPitchContext1": (None, [_vp]),
    "CreateWaveformGenerator": (_vp, []), "DestroyWaveformGenerator": (None, [_vp]),
    "CreateWaveformContext1": (_vp, []), "DestroyWaveformContext1": (None, [_vp]),
    "CreateEmbeddingSetter": (_vp, []), "DestroyEmbeddingSetter": (None, [_vp]),
    "CreateEmbeddingContext": (_vp, []), "DestroyEmbeddingContext": (None, [_vp]),
    "ReadPhoneExtractorParameters": (C.c_int, [_vp, C.c_char_p]),
    "ReadPitchEstimatorParameters": (C.c_int, [_vp, C.c_char_p]),
    "ReadWaveformGeneratorParameters": (C.c_int, [_vp, C.c_char_p]),
    "ReadEmbeddingSetterParameters": (C.c_int, [_vp, C.c_char_p]),
    "ReadNSpeakers": (C.c_int, [C.c_char_p, _i32p]),
    "ReadSpeakerEmbeddings": (C.c_int, [C.c_char_p, _f32p, _f32p, _f32p, _f32p]),
    "SetVQNumNeighbors": (None, [_vp, C.c_int]),
    "SetMinQuantizedPitch": (None, [_vp, C.c_int]),
    "SetMaxQuantizedPitch": (None, [_vp, C.c_int]),
    "ExtractPhone1": (None, [_vp, _f32p, _f32p, _vp]),
    "EstimatePitch1": (None, [_vp, _f32p, _i32p, _f32p, _vp]),
    "GenerateWaveform1": (None, [_vp, _f32p, _i32p, _f32p, _f32p, _vp]),
    "SetCodebook": (None, [_vp, _f32p]),
    "SetAdditiveSpeakerEmbedding": (None, [_vp, _f32p, _vp, _vp]),
    "SetFormantShiftEmbedding": (None, [_vp, _f32p, _vp, _vp]),
    "RegisterKeyValueSpeakerEmbedding": (None, [_vp, _f32p, _vp]),
    "SetKeyValueSpeakerEmbedding": (None, [_vp, C.c_int, _vp, _vp]),
}
ABI_SYMBOLS_RC0 = ["Beatrice20rc0_" + n for n in _RC0]
_LEGACY = ["CreatePhoneExtractor", "DestroyPhoneExtractor", "CreatePhoneContext1",
           "DestroyPhoneContext1", "CreatePitchEstimator", "DestroyPitchEstimator",
           "CreatePitchContext1", "DestroyPitchContext1", "CreateWaveformGenerator",
           "DestroyWaveformGenerator", "CreateWaveformContext1", "DestroyWaveformContext1",
           "ReadPhoneExtractorParameters", "ReadPitchEstimatorParameters",
           "ReadWaveformGeneratorParameters", "ReadNSpeakers", "ReadSpeakerEmbeddings",
           "ExtractPhone1", "SetMinQuantizedPitch", "SetMaxQuantizedPitch", "EstimatePitch1",
           "GenerateWaveform1"]
ABI_SYMBOLS_LEGACY = [g + "_" + n for g in ("Beatrice20a2", "Beatrice20b1") for n in _LEGACY]


# legacy generations (include/beatrice_abi.h: Beatrice20a2_*, Beatrice20b1_*): 256-d phone vector, 384 pitch bins, the speaker
# vector handed to GenerateWaveform1 per hop (reference lib/beatricelib/beatrice.h:39-203)
LEGACY_PHONE_CH, LEGACY_PITCH_BINS = 256, 384
_LEGACY_SIG = {
    "CreatePhoneExtractor": (_vp, []), "DestroyPhoneExtractor": (None, [_vp]),
    "CreatePhoneContext1": (_vp, []), "DestroyPhoneContext1": (None, [_vp]),
    "CreatePitchEstimator": (_vp, []), "DestroyPitchEstimator": (None, [_vp]),
    "CreatePitchContext1": (_vp, []), "DestroyPitchContext1": (None, [_vp]),
    "CreateWaveformGenerator": (_vp, []), "DestroyWaveformGenerator": (None, [_vp]),
    "CreateWaveformContext1": (_vp, []), "DestroyWaveformContext1": (None, [_vp]),
    "ReadPhoneExtractorParameters": (C.c_int, [_vp, C.c_char_p]),
    "ReadPitchEstimatorParameters": (C.c_int, [_vp, C.c_char_p]),
    "ReadWaveformGeneratorParameters": (C.c_int, [_vp, C.c_char_p]),
    "ReadNSpeakers": (C.c_int, [C.c_char_p, _i32p]),
    "ReadSpeakerEmbeddings": (C.c_int, [C.c_char_p, _f32p]),
    "ExtractPhone1": (None, [_vp, _f32p, _f32p, _vp]),
    "SetMinQuantizedPitch": (None, [_vp, C.c_int]),
    "SetMaxQuantizedPitch": (None, [_vp, C.c_int]),
    "EstimatePitch1": (None, [_vp, _f32p, _i32p, _f32p, _vp]),
    "GenerateWaveform1": (None, [_vp, _f32p, _i32p, _f32p, _f32p, _f32p, _vp]),
}


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f32p)


def iptr(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_i32p)


class Abi:
    """Typed view of any shared library exporting the Beatrice20rc0_* C-ABI."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError("beatrice library not built: %s" % path)
        self.path = path
        self.lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        for name, (res, args) in _RC0.items():
            fn = getattr(self.lib, "Beatrice20rc0_" + name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)


class AbiLegacy:
    """Typed view of a library's Beatrice20a2_* or Beatrice20b1_* entry points (generation = "20a2" | "20b1")."""

    def __init__(self, path, generation="20b1"):
        if not os.path.exists(path):
            raise FileNotFoundError("beatrice library not built: %s" % path)
        self.path, self.generation = path, generation
        self.lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        for name, (res, args) in _LEGACY_SIG.items():
            fn = getattr(self.lib, "Beatrice%s_%s" % (generation, name))
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)


class ModelsLegacy:
    """The five files of a legacy package, read as the reference's ProcessorCore0 / 1 read them
    (reference src/common/processor_core_1.cc:165-216)."""

    def __init__(self, abi, model_dir):
        self.abi = abi
        self.phone, self.pitch, self.wave = abi.CreatePhoneExtractor(), abi.CreatePitchEstimator(), abi.CreateWaveformGenerator()
        for obj, fn, name in ((self.phone, abi.ReadPhoneExtractorParameters, "phone_extractor.bin"),
                              (self.pitch, abi.ReadPitchEstimatorParameters, "pitch_estimator.bin"),
                              (self.wave, abi.ReadWaveformGeneratorParameters, "waveform_generator.bin")):
            err = fn(obj, os.path.join(model_dir, name).encode())
            if err:
                raise RuntimeError("%s: Beatrice_ErrorCode %d" % (name, err))
        n = C.c_int(0)
        path = os.path.join(model_dir, "speaker_embeddings.bin").encode()
        err = abi.ReadNSpeakers(path, C.byref(n))
        if err:
            raise RuntimeError("ReadNSpeakers error %d" % err)
        self.n_speakers = n.value
        self.speakers = np.zeros((self.n_speakers + 1, HID), np.float32)      # + the morph slot, as the reference keeps it
        err = abi.ReadSpeakerEmbeddings(path, fptr(self.speakers))
        self.formant = np.zeros((9, HID), np.float32)
        err = err or abi.ReadSpeakerEmbeddings(os.path.join(model_dir, "formant_shift_embeddings.bin").encode(), fptr(self.formant))
        if err:
            raise RuntimeError("ReadSpeakerEmbeddings error %d" % err)

    def close(self):
        a = self.abi
        a.DestroyPhoneExtractor(self.phone)
        a.DestroyPitchEstimator(self.pitch)
        a.DestroyWaveformGenerator(self.wave)


class StreamLegacy:
    """One stream through a legacy generation's per-hop protocol (reference src/common/processor_core_1.cc:50-143):
    phone, pitch, the host's pitch transform (clamped to [1, 383]), speaker vector = speaker + formant-shift row."""

    def __init__(self, models, speaker=0, formant_index=4, min_q=1, max_q=LEGACY_PITCH_BINS - 1):
        self.m, self.a = models, models.abi
        a = self.a
        self.pc, self.tc, self.wc = a.CreatePhoneContext1(), a.CreatePitchContext1(), a.CreateWaveformContext1()
        self.speaker, self.formant_index = speaker, formant_index
        self.pitch_params = {}
        a.SetMinQuantizedPitch(self.tc, min_q)
        a.SetMaxQuantizedPitch(self.tc, max_q)

    def hop(self, x160, return_all=False):
        a, m = self.a, self.m
        x = np.ascontiguousarray(x160, np.float32)
        phone = np.zeros(LEGACY_PHONE_CH, np.float32)
        a.ExtractPhone1(m.phone, fptr(x), fptr(phone), self.pc)
        q = np.zeros(1, np.int32)
        feat = np.zeros(4, np.float32)
        a.EstimatePitch1(m.pitch, fptr(x), iptr(q), fptr(feat), self.tc)
        q2 = np.array([min(pitch_transform(int(q[0]), **self.pitch_params), LEGACY_PITCH_BINS - 1)], np.int32)
        spk = np.ascontiguousarray(m.speakers[self.speaker] + m.formant[self.formant_index])      # float32 add, as the host does
        out = np.zeros(OUT_HOP, np.float32)
        a.GenerateWaveform1(m.wave, fptr(phone), iptr(q2), fptr(feat), fptr(spk), fptr(out), self.wc)
        if return_all:
            return out, phone, int(q[0]), feat, int(q2[0]), spk
        return out

    def close(self):
        a = self.a
        a.DestroyPhoneContext1(self.pc)
        a.DestroyPitchContext1(self.tc)
        a.DestroyWaveformContext1(self.wc)


def load_product():
    """The HIP library.  Raises (never falls back) when it is not built or not loadable."""
    return Abi(PRODUCT_LIB)


class SpeakerTables:
    """Caller-owned embedding tables, shaped as the reference host keeps them
    (reference src/common/processor_core_2.cc:335-351: n_speakers+1 slots, last = morph result)."""

    def __init__(self, abi, model_dir):
        path = os.path.join(model_dir, "speaker_embeddings.bin").encode()
        n = C.c_int(0)
        err = abi.ReadNSpeakers(path, C.byref(n))
        if err:
            raise RuntimeError("ReadNSpeakers error %d" % err)
        self.n_speakers = n.value
        s = self.n_speakers + 1
        self.codebooks = np.zeros((s, CODEBOOK, PHONE_CH), np.float32)
        self.additive = np.zeros((s, HID), np.float32)
        self.formant = np.zeros((9, HID), np.float32)
        self.kv = np.zeros((s, KV_LEN, KV_CH), np.float32)
        err = abi.ReadSpeakerEmbeddings(path, fptr(self.codebooks), fptr(self.additive),
                                        fptr(self.formant), fptr(self.kv))
        if err:
            raise RuntimeError("ReadSpeakerEmbeddings error %d" % err)


class Models:
    def __init__(self, abi, model_dir):
        self.abi = abi
        self.phone = abi.CreatePhoneExtractor()
        self.pitch = abi.CreatePitchEstimator()
        self.wave = abi.CreateWaveformGenerator()
        self.embed = abi.CreateEmbeddingSetter()
        for obj, fn, name in ((self.phone, abi.ReadPhoneExtractorParameters, "phone_extractor.bin"),
                              (self.pitch, abi.ReadPitchEstimatorParameters, "pitch_estimator.bin"),
                              (self.wave, abi.ReadWaveformGeneratorParameters, "waveform_generator.bin"),
                              (self.embed, abi.ReadEmbeddingSetterParameters, "embedding_setter.bin")):
            err = fn(obj, os.path.join(model_dir, name).encode())
            if err:
                raise RuntimeError("%s: Beatrice_ErrorCode %d" % (name, err))
        self.tables = SpeakerTables(abi, model_dir)

    def close(self):
        a = self.abi
        a.DestroyPhoneExtractor(self.phone)
        a.DestroyPitchEstimator(self.pitch)
        a.DestroyWaveformGenerator(self.wave)
        a.DestroyEmbeddingSetter(self.embed)


def pitch_transform(q, avg=52.0, intonation=1.0, shift=0.0, correction=0.0, ctype=0):
    """Host pitch math between EstimatePitch1 and GenerateWaveform1, in double precision
    (restates reference src/common/processor_core_2.cc:190-252)."""
    import math
    per = 8.0
    t = avg + (float(q) - avg) * intonation + per * shift
    if correction != 0.0:
        if ctype == 0:
            near = (math.floor(t / per) + 0.5) * per
            d = (t - near) * (2.0 / per)
            t = near if abs(d) < 1e-4 else near + d * math.pow(abs(d), -correction) * (per / 2.0)
        else:
            # C++ std::round: half away from zero
            tt = t / per
            near = (math.floor(tt + 0.5) if tt >= 0 else -math.floor(-tt + 0.5)) * per
            d = (t - near) * (2.0 / per)
            if correction > 1 - 1e-4:
                t = near
            elif d >= 0.0:
                t = near + math.pow(d, 1.0 / (1.0 - correction)) * (per / 2.0)
            else:
                t = near - math.pow(-d, 1.0 / (1.0 - correction)) * (per / 2.0)
    r = math.floor(t + 0.5) if t >= 0 else -math.floor(-t + 0.5)
    return int(min(max(int(r), 1), PITCH_BINS - 1))


class Stream1:
    """One stream through the 1-stream C-ABI with the reference's per-hop call protocol
    (reference src/common/processor_core_2.cc:181-255 and :431-466)."""

    def __init__(self, models, speaker=0, formant_index=4, vq_k=0, min_q=1, max_q=383):
        self.m, a = models, models.abi
        self.a = a
        self.pc, self.tc = a.CreatePhoneContext1(), a.CreatePitchContext1()
        self.wc, self.ec = a.CreateWaveformContext1(), a.CreateEmbeddingContext()
        self.kv_count = N_BLOCKS
        self.pitch_params = {}
        self.set_target_speaker(speaker)
        while self.set_kv_block():
            pass
        self.set_formant_index(formant_index)
        a.SetMinQuantizedPitch(self.tc, min_q)
        a.SetMaxQuantizedPitch(self.tc, max_q)
        a.SetVQNumNeighbors(self.pc, vq_k)

    def set_target_speaker(self, s):
        t, a = self.m.tables, self.a
        a.SetCodebook(self.pc, fptr(t.codebooks[s]))
        a.SetAdditiveSpeakerEmbedding(self.m.embed, fptr(t.additive[s]), self.ec, self.wc)
        a.RegisterKeyValueSpeakerEmbedding(self.m.embed, fptr(t.kv[s]), self.ec)
        self.kv_count = 0

    def set_formant_index(self, idx):
        self.a.SetFormantShiftEmbedding(self.m.embed, fptr(self.m.tables.formant[idx]), self.ec, self.wc)

    def set_kv_block(self):
        if self.kv_count < N_BLOCKS:
            self.a.SetKeyValueSpeakerEmbedding(self.m.embed, self.kv_count, self.ec, self.wc)
            self.kv_count += 1
            return True
        return False

    def hop(self, x160, return_all=False):
        a = self.a
        x = np.ascontiguousarray(x160, np.float32)
        self.set_kv_block()
        phone = np.zeros(PHONE_CH, np.float32)
        a.ExtractPhone1(self.m.phone, fptr(x), fptr(phone), self.pc)
        q = np.zeros(1, np.int32)
        feat = np.zeros(4, np.float32)
        a.EstimatePitch1(self.m.pitch, fptr(x), iptr(q), fptr(feat), self.tc)
        q2 = np.array([pitch_transform(int(q[0]), **self.pitch_params)], np.int32)
        out = np.zeros(OUT_HOP, np.float32)
        a.GenerateWaveform1(self.m.wave, fptr(phone), iptr(q2), fptr(feat), fptr(out), self.wc)
        if return_all:
            return out, phone, int(q[0]), feat, int(q2[0])
        return out

    def close(self):
        a = self.a
        a.DestroyPhoneContext1(self.pc)
        a.DestroyPitchContext1(self.tc)
        a.DestroyWaveformContext1(self.wc)
        a.DestroyEmbeddingContext(self.ec)


def synth_audio(n_samples, seed=0, sr=16000, silence_gap=False):
    """Deterministic voiced-like test signal: gliding band-limited saw,
    4 Hz syllable envelope, low-level noise.  silence_gap: 0.5 s of digital silence from 0.1 s on (the survey's
    "10 % of the streams contain 0.5 s digital silence gaps"; the harness converts them like anything else)."""
    rng = np.random.Generator(np.random.PCG64(0xBEA7 + seed))
    t = np.arange(n_samples) / sr
    period = 2.0 + 2.0 * rng.random()
    f0 = 90.0 * (400.0 / 90.0) ** (0.5 - 0.5 * np.cos(2 * np.pi * t / period + rng.random() * 6.28))
    ph = 2 * np.pi * np.cumsum(f0) / sr
    x = np.zeros(n_samples)
    for h in range(1, 20):
        x += np.sin(h * ph) / h * (h * f0 < 0.45 * sr)
    env = 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * t + rng.random() * 6.28)
    x = 0.3 * x / 1.8 * env + 0.0316 * 0.3 * rng.standard_normal(n_samples)
    if silence_gap:
        x[int(0.1 * sr):int(0.6 * sr)] = 0.0
    return x.astype(np.float32)


# ------------------------------------------------------------------------------------------------
# Batched extension ABI (include/beatrice_batch.h)
# ------------------------------------------------------------------------------------------------
_BATCH = {
    "BeatriceHip_LoadPhoneExtractorFromMemory": (C.c_int, [_vp, _vp, C.c_size_t]),
    "BeatriceHip_LoadPitchEstimatorFromMemory": (C.c_int, [_vp, _vp, C.c_size_t]),
    "BeatriceHip_LoadWaveformGeneratorFromMemory": (C.c_int, [_vp, _vp, C.c_size_t]),
    "BeatriceHip_LoadEmbeddingSetterFromMemory": (C.c_int, [_vp, _vp, C.c_size_t]),
    "BeatriceHip_SetDevice": (C.c_int, [C.c_int]),
    "BeatriceHip_GetDevice": (C.c_int, []),
    "BeatriceBatch_Device": (C.c_int, [_vp]),
    "BeatriceHip_MathSelfTest": (C.c_longlong, [C.c_int, C.POINTER(C.c_uint)]),
    "BeatriceHip_InvalidateCodebook": (None, [_vp, _vp]),
    "BeatriceHip_InjectTeamTimeout": (C.c_int, [_vp]),
    "BeatriceHip_InjectTeamTimeoutPhone": (C.c_int, [_vp]),
    "BeatriceHip_InjectTeamTimeoutPitch": (C.c_int, [_vp]),
    "BeatriceHip_PitchSpeculation": (C.c_int, [_vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "BeatriceBatch_InjectTeamTimeout": (C.c_int, [_vp]),
    "BeatriceHip_ModelBlob": (C.c_int, [C.c_int, _vp, C.c_int, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "BeatriceHip_ModelBlobReady": (C.c_int, [C.c_int, _vp]),
    "BeatriceBatch_Create": (_vp, [_vp, _vp, _vp, _vp, C.c_int, C.c_int]),
    "BeatriceBatch_CreateBlock": (_vp, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    "BeatriceBatch_HopsPerStep": (C.c_int, [_vp]),
    "BeatriceBatch_StateBytes": (C.c_size_t, [_vp]),
    "BeatriceBatch_Destroy": (None, [_vp]),
    "BeatriceBatch_IsHealthy": (C.c_int, [_vp]),
    "BeatriceBatch_NumStreams": (C.c_int, [_vp]),
    "BeatriceBatch_SetSpeakerTables": (C.c_int, [_vp, C.c_int, _f32p, _f32p, _f32p, _f32p]),
    "BeatriceBatch_BindResidentIO48k": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int]),
    "BeatriceBatch_EnableHostStreaming": (C.c_int, [_vp, C.c_int]),
    "BeatriceBatch_HostStreamDelay": (C.c_int, [_vp]),
    "BeatriceBatch_StreamFrames": (C.c_int, [_vp, _f32p, _f32p]),
    "BeatriceBatch_StreamFlush": (C.c_int, [_vp, _f32p]),
    "BeatriceBatch_SpeakerTablesDevice": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "BeatriceBatch_ProjectSpeakerTables": (C.c_int, [_vp, C.c_int]),
    "BeatriceBatch_UpdateSpeaker": (C.c_int, [_vp, C.c_int, _f32p, _f32p, _f32p]),
    "BeatriceBatch_SeedLottery": (C.c_int, [_vp, C.c_int, C.c_uint]),
    "BeatriceBatch_MorphSpeaker": (C.c_int, [_vp, C.c_int, _f32p, C.c_int, C.c_uint]),
    "BeatriceBatch_EnableSilentBlockRule": (C.c_int, [_vp, C.c_int]),
    "BeatriceBatch_SetSilentStreams": (C.c_int, [_vp, C.c_char_p]),
    "BeatriceBatch_BindResidentBlocks": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    "BeatriceBatch_ResidentBlocksDelay": (C.c_int, [_vp]),
    "BeatriceBatch_ResidentBlocksDelayFor": (C.c_int, [_vp, C.c_int]),
    "BeatriceBatch_ResidentBlocksOwed": (C.c_int, [_vp]),
    "BeatriceBatch_FlushResidentBlocks": (C.c_int, [_vp]),
    "BeatriceBatch_BindResidentBlocksRagged": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    "BeatriceBatch_ProcessBlocksRaggedDevice": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "BeatriceBatch_MorphSpeakerStaged": (C.c_int, [_vp, C.c_int, C.c_int, _f32p, C.c_int, C.c_uint]),
    "BeatriceBatch_GetSpeakerEmbeddings": (C.c_int, [_vp, C.c_int, _f32p, _f32p]),
    "BeatriceBatch_SetTargetSpeaker": (C.c_int, [_vp, C.c_int, C.c_int]),
    "BeatriceBatch_SetTargetSpeakers": (C.c_int, [_vp, C.c_int, _i32p, _i32p]),
    "BeatriceBatch_FlushSpeaker": (C.c_int, [_vp, C.c_int]),
    "BeatriceBatch_SetFormantShift": (C.c_int, [_vp, C.c_int, C.c_double]),
    "BeatriceBatch_SetVQNumNeighbors": (C.c_int, [_vp, C.c_int, C.c_int]),
    "BeatriceBatch_SetMinSourcePitch": (C.c_int, [_vp, C.c_int, C.c_double]),
    "BeatriceBatch_SetMaxSourcePitch": (C.c_int, [_vp, C.c_int, C.c_double]),
    "BeatriceBatch_SetPitchShift": (C.c_int, [_vp, C.c_int, C.c_double]),
    "BeatriceBatch_SetAverageSourcePitch": (C.c_int, [_vp, C.c_int, C.c_double]),
    "BeatriceBatch_SetIntonationIntensity": (C.c_int, [_vp, C.c_int, C.c_double]),
    "BeatriceBatch_SetPitchCorrection": (C.c_int, [_vp, C.c_int, C.c_double]),
    "BeatriceBatch_SetPitchCorrectionType": (C.c_int, [_vp, C.c_int, C.c_int]),
    "BeatriceBatch_ResetStream": (C.c_int, [_vp, C.c_int]),
    "BeatriceBatch_ConvertFrames": (C.c_int, [_vp, _f32p, _f32p]),
    "BeatriceBatch_ConvertFramesDevice": (C.c_int, [_vp, _vp, _vp]),
    "BeatriceBatch_ConvertBlocks48k": (C.c_int, [_vp, _f32p, _f32p, C.c_int]),
    "BeatriceBatch_ConvertBlocks48kDevice": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "BeatriceBatch_ConfigureWrapper": (C.c_int, [_vp, C.c_double]),
    "BeatriceBatch_SetInputGain": (C.c_int, [_vp, C.c_int, C.c_double]),
    "BeatriceBatch_SetOutputGain": (C.c_int, [_vp, C.c_int, C.c_double]),
    "BeatriceBatch_ProcessBlocks": (C.c_int, [_vp, _f32p, _f32p, C.c_int, C.c_int]),
    "BeatriceBatch_ProcessBlocksDevice": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int]),
    "BeatriceBatch_MaxWrapperBlock": (C.c_int, [_vp]),
    "BeatriceBatch_ConfigureWrapperRates": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "BeatriceBatch_ProcessBlocksRagged": (C.c_int, [_vp, _vp, _vp, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "BeatriceBatch_Synchronize": (C.c_int, [_vp]),
    "BeatriceBatch_BindResidentIO": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "BeatriceBatch_SetStream": (C.c_int, [_vp, _vp]),
    "BeatriceBatch_GetStream": (_vp, [_vp]),
    "BeatriceBatch_EnableGraph": (C.c_int, [_vp, C.c_int]),
    "BeatriceBatch_EnablePipelining": (C.c_int, [_vp, C.c_int]),
    "BeatriceBatch_GetWaveStream": (_vp, [_vp]),
    "BeatriceBatch_EnableTickPipeline": (C.c_int, [_vp, C.c_int]),
    "BeatriceBatch_TickStages": (C.c_int, [_vp]),
    "BeatriceBatch_TimeTickLaunch": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "BeatriceBatch_Prepare": (C.c_int, [_vp]),
    "BeatriceBatch_DeviceInput": (_vp, [_vp]),
    "BeatriceBatch_DeviceOutput": (_vp, [_vp]),
    "BeatriceBatch_GetIntermediates": (C.c_int, [_vp, _f32p, _i32p, _i32p, _f32p]),
    "BeatriceBatch_TimeSteps": (C.c_int, [_vp, C.c_int, _f32p]),
    "BeatriceBatch_ProfileKernels": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p, _i32p,
                                               C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}
ABI_SYMBOLS_BATCH = list(_BATCH)


def bind_batch(abi):
    """Attach typed prototypes of the batched extension to an Abi (product library only)."""
    if getattr(abi, "_batch_bound", False):
        return abi
    for name, (res, args) in _BATCH.items():
        if os.environ.get("BEATRICE_HIP_LIB") and not hasattr(abi.lib, name):
            continue   # an older build of the library loaded for an A/B measurement: it simply lacks the newer entry points
        fn = getattr(abi.lib, name)
        fn.restype, fn.argtypes = res, args
        setattr(abi, name, fn)
    abi._batch_bound = True
    return abi


class Batch:
    """B concurrent streams on one GPU through include/beatrice_batch.h.

    hops_per_step = 1: one 10 ms hop per step (real time).  2 or 4: block mode, every step converts that
    many consecutive hops per stream (in [B][H*160] -> out [B][H*240]), same results as single hops."""

    def __init__(self, models, n_streams, max_speakers=None, hops_per_step=1, upload_tables=True):
        self.m = models
        self.a = bind_batch(models.abi)
        t = models.tables if upload_tables else None
        self.B = n_streams
        self.H = hops_per_step
        if t is None and max_speakers is None:
            raise ValueError("Batch(upload_tables=False) needs max_speakers: without host tables the table size is unknown")
        ms = max_speakers or (t.n_speakers + 1)
        if hops_per_step == 1:
            self.h = self.a.BeatriceBatch_Create(models.phone, models.pitch, models.wave, models.embed, n_streams, ms)
        else:
            self.h = self.a.BeatriceBatch_CreateBlock(models.phone, models.pitch, models.wave, models.embed, n_streams, ms,
                                                      hops_per_step)
        if not self.a.BeatriceBatch_IsHealthy(self.h):
            raise RuntimeError("BeatriceBatch_Create failed (no GPU / HIP error)")
        if not upload_tables:  # the caller fills the device tables itself (shard.share_speaker_tables)
            return
        self._check(self.a.BeatriceBatch_SetSpeakerTables(self.h, t.n_speakers + 1, fptr(t.codebooks), fptr(t.additive),
                                                          fptr(t.formant), fptr(t.kv)))
        self.apply_defaults()

    def apply_defaults(self):
        # reference host defaults (processor_core_2.h:103-113): speaker 0 with all K/V blocks installed
        self._check(self.a.BeatriceBatch_SetTargetSpeaker(self.h, -1, 0))
        self._check(self.a.BeatriceBatch_FlushSpeaker(self.h, -1))
        self._check(self.a.BeatriceBatch_SetMinSourcePitch(self.h, -1, 33.125))
        self._check(self.a.BeatriceBatch_SetMaxSourcePitch(self.h, -1, 80.875))

    @staticmethod
    def _check(rc):
        if rc != 0:
            raise RuntimeError("BeatriceBatch call failed: %d" % rc)

    def convert(self, x):
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape == (self.B, self.H * IN_HOP)
        out = np.zeros((self.B, self.H * OUT_HOP), np.float32)
        self._check(self.a.BeatriceBatch_ConvertFrames(self.h, fptr(x), fptr(out)))
        return out

    def convert48k(self, x, channels):
        """x: [B][channels][480] planar float @48 kHz -> same shape (device-side wrapper, configs[4])."""
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape == (self.B, channels, 480)
        out = np.zeros_like(x)
        self._check(self.a.BeatriceBatch_ConvertBlocks48k(self.h, fptr(x), fptr(out), channels))
        return out

    def intermediates(self):
        shape = (self.B,) if self.H == 1 else (self.B, self.H)
        phone = np.zeros(shape + (PHONE_CH,), np.float32)
        q_raw = np.zeros(shape, np.int32)
        q = np.zeros(shape, np.int32)
        feat = np.zeros(shape + (4,), np.float32)
        self._check(self.a.BeatriceBatch_GetIntermediates(self.h, fptr(phone), iptr(q_raw), iptr(q), fptr(feat)))
        return phone, q_raw, q, feat

    def time_steps(self, steps):
        ms = C.c_float(0)
        self._check(self.a.BeatriceBatch_TimeSteps(self.h, steps, C.byref(ms)))
        return ms.value

    def profile_kernels(self, repeats=10, max_entries=96):
        """[{name, launches, mean_us, flops, bytes}] for one hop (BeatriceBatch_ProfileKernels)."""
        names = C.create_string_buffer(64 * max_entries)
        launches = (C.c_int * max_entries)()
        us = (C.c_double * max_entries)()
        fl = (C.c_double * max_entries)()
        by = (C.c_double * max_entries)()
        n = self.a.BeatriceBatch_ProfileKernels(self.h, repeats, max_entries, names, launches, us, fl, by)
        if n < 0:
            raise RuntimeError("BeatriceBatch_ProfileKernels failed: %d" % n)
        rows = []
        for i in range(n):
            nm = names.raw[64 * i:64 * (i + 1)].split(b"\0")[0].decode()
            rows.append(dict(name=nm, launches=launches[i], mean_us=us[i], flops=fl[i], bytes=by[i]))
        return rows

    def close(self):
        if self.h:
            self.a.BeatriceBatch_Destroy(self.h)
            self.h = None
