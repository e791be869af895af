"""Host layer above the core: model package reader (TOML subset + ModelConfig), version dispatch with the silent
"unloaded" fallback, parameter fan-out, and the preset / project-state wire format -- this project's counterparts of the
reference's ProcessorProxy, ModelConfig, ParameterSchema and ParameterState (reference src/common/processor_proxy.{h,cc},
model_config.h, parameter_schema.cc, parameter_state.cc).  CPU only: the C++ host layer is built against the oracle core.

Pinning: the reference's own reader needs toml11 (absent submodule, no stand-ins), so these checks are against
hand-derived known answers from the reference's source text: error classes per failure (processor_proxy.h:76-99),
record layout and byte order of the state blob (parameter_state.cc:68-147), defaults of the table
(parameter_schema.cc:51-477), the morph weight formula (voice_morph_state.h:50-85)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import hostlib

OK, FILE_OPEN, TOO_SMALL, TOML_SYNTAX, INVALID_CONFIG, SPEAKER_RANGE, NOT_LOADED, UNKNOWN = 0, 1, 2, 5, 6, 7, 9, 12
K_MODEL, K_VOICE, K_FORMANT, K_PITCH_SHIFT, K_IN_GAIN, K_OUT_GAIN, K_VQ = 1, 2, 3, 4, 7, 8, 14
K_CURSOR_X, K_FALLOFF, K_MARKER_COUNT, K_MARKER_VOICE0, K_MARKER_X0, K_MARKER_Y0, K_TARGET_PITCH0 = 15, 17, 18, 19, 27, 35, 100
_f32p = C.POINTER(C.c_float)


class Proxy:
    def __init__(self, sample_rate=48000.0):
        L = self.lib = C.CDLL(hostlib.HOST_ON_ORACLE)
        L.BeatriceProxy_Create.restype = C.c_void_p
        for name, args in (("Destroy", []), ("SetSampleRate", [C.c_double]), ("LoadModel", [C.c_char_p]), ("SetNumber", [C.c_int, C.c_double]),
                           ("SetInt", [C.c_int, C.c_int]), ("SetString", [C.c_int, C.c_char_p]), ("Process", [_f32p, _f32p, C.c_int]),
                           ("ResetContext", []), ("CoreVersion", []), ("VoiceCount", []), ("GetKind", [C.c_int]), ("GetNumber", [C.c_int]),
                           ("GetString", [C.c_int, C.c_char_p, C.c_int]), ("WriteState", [C.c_char_p, C.c_int]), ("ReadState", [C.c_char_p, C.c_int]),
                           ("MorphWeights", [_f32p]), ("ProcessChannels", [_f32p, _f32p, _f32p, _f32p, C.c_int])):
            getattr(L, "BeatriceProxy_" + name).argtypes = [C.c_void_p] + args
        L.BeatriceProxy_GetNumber.restype = C.c_double
        self.h = L.BeatriceProxy_Create()
        assert self.call("SetSampleRate", sample_rate) == OK

    def call(self, name, *a):
        return getattr(self.lib, "BeatriceProxy_" + name)(self.h, *a)

    def process(self, x, block=480):
        x = np.ascontiguousarray(x, np.float32)
        out = np.full_like(x, 7.0)
        codes = []
        for pos in range(0, len(x), block):
            n = min(block, len(x) - pos)
            codes.append(self.call("Process", x[pos:pos + n].ctypes.data_as(_f32p), out[pos:pos + n].ctypes.data_as(_f32p), n))
        return out, codes

    def state(self):
        n = self.call("WriteState", None, 0)
        buf = C.create_string_buffer(n)
        assert self.call("WriteState", buf, n) == n
        return buf.raw

    def close(self):
        self.call("Destroy")


def _toml(model_dir):
    return os.path.join(model_dir, "model.toml").encode()


def _signal(n, seed=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n) / 48000.0
    return (0.3 * np.sin(2 * np.pi * 180.0 * t) + 0.02 * rng.standard_normal(n)).astype(np.float32)


def test_load_by_toml_and_process_equals_core(built, model_dir):
    x = _signal(480 * 12)
    p = Proxy()
    assert p.call("CoreVersion") == -1
    silent, codes = p.process(x[:960])
    assert not silent.any() and set(codes) == {NOT_LOADED}          # unloaded core: zeros (processor_core.h:95-104)
    assert p.call("LoadModel", _toml(model_dir)) == OK
    assert p.call("CoreVersion") == 2 and p.call("VoiceCount") == 3
    assert p.call("SetInt", K_VOICE, 1) == OK
    assert p.call("SetNumber", K_VQ, 2.6) == OK                      # rounds to 3 (parameter_schema.cc: VQ lambda)
    assert p.call("SetNumber", K_FORMANT, 1.0) == OK
    got, codes = p.process(x)
    assert set(codes) == {OK}
    h = hostlib.Host(hostlib.HOST_ON_ORACLE, 48000)
    assert h.load(model_dir) == 0
    h.call("SetTargetSpeaker", 1); h.call("SetVQNumNeighbors", 3); h.call("SetFormantShift", 1.0)
    want, _ = h.process(x, 480)
    h.close()
    assert np.abs(want).max() > 1e-3 and np.array_equal(got, want)
    assert p.call("SetInt", K_VOICE, 300) == SPEAKER_RANGE           # stored all the same, like the reference
    assert int(p.call("GetNumber", K_VOICE)) == 300
    p.close()


def _write_pkg(tmp_path, model_dir, text):
    d = tmp_path / ("pkg%d" % len(list(tmp_path.iterdir())))
    d.mkdir()
    for f in os.listdir(model_dir):
        if f.endswith(".bin"):
            os.symlink(os.path.join(model_dir, f), d / f)
    (d / "model.toml").write_text(text)
    return str(d / "model.toml").encode()




@pytest.mark.parametrize("edit,want", [
    (lambda t: t, OK),
    (lambda t: t.replace('version = "2.0.0-rc.0"', 'version = "9.9.9"'), INVALID_CONFIG),            # unknown generation
    (lambda t: t.replace('version = "2.0.0-rc.0"', 'version = "2.0.0-beta.1"'), 4),                   # a legacy core on rc.0 files: kInvalidFileSize
    (lambda t: t.replace('version = "2.0.0-rc.0"', 'version = "2.0.0-alpha.2"'), 4),
    (lambda t: t.replace("average_pitch = 52.0", "average_pitch = 52", 1), INVALID_CONFIG),           # integer is not a float
    (lambda t: t.replace("average_pitch = 52.0", "average_pitch = 200.0", 1), INVALID_CONFIG),
    (lambda t: t.replace("average_pitch = 52.0", "average_pitch = nan", 1), INVALID_CONFIG),
    (lambda t: t.replace('description = "synthetic speaker 1"\n', ""), INVALID_CONFIG),               # missing key
    (lambda t: t.replace("[voice.1]", "[voice.7]").replace("[voice.1.portrait]", "[voice.7.portrait]"), INVALID_CONFIG),  # not contiguous
    (lambda t: t.replace("[voice.2]", "[voice.300]").replace("[voice.2.portrait]", "[voice.300.portrait]"), INVALID_CONFIG),
    (lambda t: t.replace("[voice.2]", "[voice.x]").replace("[voice.2.portrait]", "[voice.x.portrait]"), INVALID_CONFIG),
    (lambda t: t.replace("[model]", "[model"), TOML_SYNTAX),
    (lambda t: t + '\nname = "dup"\nname = "dup"\n', TOML_SYNTAX),
    (lambda t: t.replace('name = "spk0"', 'name = "spk0'), TOML_SYNTAX),
    (lambda t: "# comment only\n", INVALID_CONFIG),                                                    # no [voice] table at all
    (lambda t: t.replace('name = "spk0"', "name = 'sp\\u006b0' # literal string, no escapes").replace('[voice.0]\n', '[voice.0]\n# a comment\n'), OK),
])
def test_model_config_failure_classes(built, model_dir, tmp_path, edit, want):
    good = open(os.path.join(model_dir, "model.toml")).read()
    p = Proxy()
    assert p.call("LoadModel", _toml(model_dir)) == OK                # something loaded first ...
    rc = p.call("LoadModel", _write_pkg(tmp_path, model_dir, edit(good)))
    assert rc == want
    assert p.call("CoreVersion") == (2 if want == OK else -1)          # ... any failure leaves the unloaded core
    if want != OK:
        out, codes = p.process(_signal(960))
        assert not out.any() and set(codes) == {NOT_LOADED}
    p.close()


def test_missing_files(built, model_dir, tmp_path):
    p = Proxy()
    assert p.call("LoadModel", b"/nonexistent/model.toml") == FILE_OPEN
    assert p.call("LoadModel", b"") == OK and p.call("CoreVersion") == -1   # empty path: unload, no error
    d = tmp_path / "only_toml"
    d.mkdir()
    (d / "model.toml").write_text(open(os.path.join(model_dir, "model.toml")).read())
    assert p.call("LoadModel", str(d / "model.toml").encode()) == FILE_OPEN  # the .bin files are missing
    assert p.call("CoreVersion") == -1
    p.close()


def _records(blob):
    out, at = [], 0
    while at < len(blob):
        pid, typ = struct.unpack_from("<hi", blob, at)
        at += 6
        if typ == 0:
            val = struct.unpack_from("<i", blob, at)[0]; at += 4
        elif typ == 1:
            val = struct.unpack_from("<d", blob, at)[0]; at += 8
        else:
            n = struct.unpack_from("<i", blob, at)[0]; at += 4
            val = blob[at:at + n]; at += n
        out.append((pid, typ, val))
    return out


def test_state_blob_layout_and_defaults(built):
    p = Proxy()
    blob = p.state()
    p.close()
    # the first records byte for byte: kModel = empty string, kVoice = int 0, kFormantShift = double 0.0
    assert blob[:10] == bytes.fromhex("0100" "02000000" "00000000")
    assert blob[10:20] == bytes.fromhex("0200" "00000000" "00000000")
    assert blob[20:34] == bytes.fromhex("0300" "01000000") + struct.pack("<d", 0.0)
    rec = _records(blob)
    ids = [r[0] for r in rec]
    assert ids == sorted(ids) and len(ids) == 14 + 4 + 3 * 8 + 257        # std::map order; the whole table
    d = {r[0]: r for r in rec}
    assert d[5][1:] == (1, 52.0) and d[9][1:] == (1, 1.0) and d[12][1:] == (1, 33.125) and d[13][1:] == (1, 80.875)
    assert d[6][1] == 0 and d[11][1] == 0 and d[14][1:] == (1, 0.0)
    assert d[K_FALLOFF][2] == 2.0 and d[K_MARKER_COUNT][2] == 4.0
    assert [d[K_MARKER_VOICE0 + i][2] for i in range(8)] == [0, 1, 2, 3, 0, 0, 0, 0]
    assert d[K_MARKER_X0][2] == float(np.float32(0.18)) and d[K_MARKER_Y0 + 3][2] == float(np.float32(0.82))   # float literals widened
    assert all(d[K_TARGET_PITCH0 + i][1:] == (1, 60.0) for i in range(257))


def test_state_round_trip_reloads_model(built, model_dir):
    x = _signal(480 * 10, seed=5)
    a = Proxy()
    assert a.call("SetString", K_MODEL, _toml(model_dir)) == OK          # loading through the parameter, as the VST does
    assert a.call("CoreVersion") == 2
    for pid, v in ((K_PITCH_SHIFT, 3.5), (K_IN_GAIN, -6.0), (K_OUT_GAIN, 2.0), (K_VQ, 4.0), (K_CURSOR_X, 0.3)):
        assert a.call("SetNumber", pid, v) == OK
    assert a.call("SetInt", K_VOICE, 2) == OK
    blob = a.state()
    want, _ = a.process(x)
    b = Proxy()
    assert b.call("ReadState", blob, len(blob)) == OK
    assert b.call("CoreVersion") == 2 and b.state() == blob
    buf = C.create_string_buffer(4096)
    assert b.call("GetString", K_MODEL, buf, 4096) == len(_toml(model_dir)) and buf.value == _toml(model_dir)
    got, _ = b.process(x)
    assert np.abs(want).max() > 1e-3 and np.array_equal(got, want)
    # truncated blob: what was read is kept, the rest are defaults, error = file too small
    c = Proxy()
    cut = blob.index(struct.pack("<hi", K_IN_GAIN, 1)) + 9
    assert c.call("ReadState", blob[:cut], cut) == TOO_SMALL
    assert c.call("GetNumber", K_PITCH_SHIFT) == 3.5 and c.call("GetNumber", K_IN_GAIN) == 0.0
    assert c.call("ReadState", b"", 0) == TOO_SMALL
    bad = struct.pack("<hii", K_VOICE, 5, 0)
    assert c.call("ReadState", bad, len(bad)) == UNKNOWN
    for q in (a, b, c):
        q.close()


def test_morph_weights_formula(built):
    """weights = normalised 1 / (d^2 + 0.0008)^falloff per marker, summed per voice (voice_morph_state.h:50-85), float32."""
    p = Proxy()
    p.call("SetNumber", K_CURSOR_X, 0.31)
    p.call("SetNumber", K_CURSOR_X + 1, 0.64)
    p.call("SetNumber", K_FALLOFF, 1.5)
    p.call("SetNumber", K_MARKER_VOICE0 + 3, 1.0)      # markers 1 and 3 both point at voice 1
    got = np.zeros(256, np.float32)
    p.call("MorphWeights", got.ctypes.data_as(_f32p))
    f = np.float32
    mx, my = [f(0.18), f(0.82), f(0.5), f(0.5)], [f(0.5), f(0.5), f(0.18), f(0.82)]
    cx, cy = f(0.31), f(0.64)
    raw = []
    for i in range(4):
        dx, dy = f(cx - mx[i]), f(cy - my[i])
        raw.append(f(1.0) / f(np.power(f(f(dx * dx) + f(dy * dy)) + f(0.0008), f(1.5))))
    total = f(0)
    for r in raw:
        total = f(total + r)
    mw = [f(r / total) for r in raw]
    want = np.zeros(256, np.float32)
    want[0], want[1], want[2] = mw[0], f(mw[1] + mw[3]), mw[2]
    assert np.allclose(got, want, rtol=2e-6, atol=0) and abs(float(got.sum()) - 1.0) < 1e-6
    p.call("SetNumber", K_FALLOFF, 0.0)                # falloff 0: equal marker weights
    p.call("MorphWeights", got.ctypes.data_as(_f32p))
    assert got[0] == f(0.25) and got[1] == f(0.5) and got[2] == f(0.25)
    p.close()


def test_vst_shell_block_handling(built, model_dir):
    """ProcessChannels = the block handling of the reference's VST shell (src/vst/processor.cc:183-225): stereo is
    down-mixed (L + R) * 0.5, an all-zero block is not converted at all -- the core's state and its 10 ms FIFO stand
    still, so the remaining blocks come out as if the silent block had never been there -- and a second output channel
    is a copy.  Known answers: the same proxy driven through Process with the equivalent mono input."""
    rng = np.random.default_rng(11)
    n, blocks = 512, 9
    L_ = (0.2 * rng.standard_normal(n * blocks)).astype(np.float32)
    R_ = (0.2 * rng.standard_normal(n * blocks)).astype(np.float32)
    silent_at = {3, 4, 7}
    for k in silent_at:
        L_[k * n:(k + 1) * n] = 0.0
        R_[k * n:(k + 1) * n] = 0.0
    mono = ((L_ + R_) * np.float32(0.5)).astype(np.float32)      # (float add, float multiply: the shell's two roundings)

    def fresh():
        p = Proxy(44100.0)
        assert p.call("LoadModel", os.path.join(model_dir, "model.toml").encode()) == OK
        return p

    p = fresh()
    out0 = np.full(n * blocks, 5.0, np.float32)
    out1 = np.full(n * blocks, 6.0, np.float32)
    flags = []
    for k in range(blocks):
        sl = slice(k * n, (k + 1) * n)
        flags.append(p.call("ProcessChannels", L_[sl].ctypes.data_as(_f32p), R_[sl].ctypes.data_as(_f32p),
                            out0[sl].ctypes.data_as(_f32p), out1[sl].ctypes.data_as(_f32p), n))
    assert flags == [1 if k in silent_at else 0 for k in range(blocks)]
    assert np.array_equal(out0, out1)
    for k in silent_at:
        assert not out0[k * n:(k + 1) * n].any()
    # reference run: only the non-silent blocks, through Process, mono
    q = fresh()
    keep = [k for k in range(blocks) if k not in silent_at]
    want, codes = q.process(np.concatenate([mono[k * n:(k + 1) * n] for k in keep]), block=n)
    assert all(c == OK for c in codes)
    got = np.concatenate([out0[k * n:(k + 1) * n] for k in keep])
    assert np.abs(want).max() > 1e-3 and np.array_equal(got, want)
    # mono input, no second output channel
    r = fresh()
    o = np.zeros(n, np.float32)
    assert r.call("ProcessChannels", mono[:n].ctypes.data_as(_f32p), None, o.ctypes.data_as(_f32p), None, n) == 0
    assert np.array_equal(o, want[:n])


def test_wrongly_typed_state_records_cannot_poison_the_table(built, model_dir):
    """A state blob is untrusted input.  Records of KNOWN ids whose type differs from the schema's (a morph cursor typed
    int, a marker typed string, the voice typed float64) are dropped -- the defaults stay -- so no later typed read can
    throw across the C boundary; unknown ids keep whatever they carry; the typed setters refuse a wrong kind."""
    p = Proxy()
    assert p.call("LoadModel", _toml(model_dir)) == OK
    path = _toml(model_dir)
    blob = b"".join([
        struct.pack("<hii", K_MODEL, 2, len(path)) + path,             # (a state without the model path unloads the core)
        struct.pack("<hii", K_VOICE, 0, 2),                            # fine: voice 2
        struct.pack("<hid", K_VOICE, 1, 1.0),                          # wrong: the voice is an int (dropped; 2 stays)
        struct.pack("<hii", K_CURSOR_X, 0, 1),                         # wrong: a morph number typed int
        struct.pack("<hii", K_MARKER_X0 + 1, 2, 3) + b"abc",           # wrong: a morph number typed string
        struct.pack("<hid", K_FALLOFF, 1, 3.0),                        # fine, and its sync reads ALL the morph ids
        struct.pack("<hii", 777, 2, 2) + b"zz",                        # unknown id: kept
    ])
    assert p.call("ReadState", blob, len(blob)) == OK
    assert p.call("GetKind", K_VOICE) == 0 and p.call("GetNumber", K_VOICE) == 2.0
    assert p.call("GetKind", K_CURSOR_X) == 1 and p.call("GetNumber", K_CURSOR_X) == 0.5
    assert p.call("GetKind", K_MARKER_X0 + 1) == 1 and abs(p.call("GetNumber", K_MARKER_X0 + 1) - 0.82) < 1e-6
    assert p.call("GetNumber", K_FALLOFF) == 3.0
    buf = C.create_string_buffer(8)
    assert p.call("GetString", 777, buf, 8) == 2 and buf.value == b"zz"
    w = np.zeros(256, np.float32)
    p.call("MorphWeights", w.ctypes.data_as(_f32p))                    # used to terminate the process on a poisoned table
    assert abs(float(w.sum()) - 1.0) < 1e-5
    # getters on ids that hold nothing: no throw, neutral answers
    assert p.call("GetKind", 9999) == -1 and p.call("GetNumber", 9999) == 0.0 and p.call("GetString", 9999, buf, 8) == -1
    # the typed setters refuse the wrong kind for a known id and leave the value alone
    assert p.call("SetInt", K_CURSOR_X, 1) == UNKNOWN and p.call("GetNumber", K_CURSOR_X) == 0.5
    assert p.call("SetNumber", K_VOICE, 1.0) == UNKNOWN and p.call("GetNumber", K_VOICE) == 2.0
    assert p.call("SetString", K_FORMANT, b"x") == UNKNOWN
    assert p.call("SetNumber", K_CURSOR_X, 0.25) == OK
    x = _signal(480 * 4)
    out, codes = p.process(x)
    assert set(codes) == {OK} and np.isfinite(out).all()
    p.close()


def test_deeply_nested_toml_is_a_syntax_error_not_a_crash(built, model_dir, tmp_path):
    """toml_subset parses arrays and inline tables by recursion; a hostile model.toml must end as kTOMLSyntaxError (the
    silent unloaded core), not as a stack overflow."""
    import shutil
    d = tmp_path / "deep"
    shutil.copytree(model_dir, d)
    text = open(os.path.join(model_dir, "model.toml")).read()
    for opener, closer in (("[", "]"), ("{ a = ", " }")):
        (d / "model.toml").write_text(text + "\nevil = " + opener * 200000 + "1" + closer * 200000 + "\n")
        p = Proxy()
        assert p.call("LoadModel", str(d / "model.toml").encode()) == TOML_SYNTAX
        assert p.call("CoreVersion") == -1
        p.close()
    (d / "model.toml").write_text(text + "\nfine = " + "[" * 40 + "1" + "]" * 40 + "\n")      # 40 levels are fine
    p = Proxy()
    assert p.call("LoadModel", str(d / "model.toml").encode()) == OK
    p.close()
