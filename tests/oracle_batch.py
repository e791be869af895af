"""The ORACLE behind the batched interface, for a sample of a batch's streams (test infrastructure).

`OracleBatch` looks like `bv.Batch` to a test script -- `.a.BeatriceBatch_Set*(h, stream, ...)`, `.h`, `convert(x)` --
but every sampled stream is an independent `Stream1` on oracle/libbeatrice_oracle.so driven through the reference
host's per-hop protocol (reference src/common/processor_core_2.cc:50-256: one pending K/V block installed per hop
`:179-181`, then ExtractPhone1 / EstimatePitch1 / the double-precision pitch transform `:190-252` / GenerateWaveform1;
setters as `:431-590`; ResetContext `:258-291` = fresh contexts + every setting re-applied with all four K/V blocks).
A throughput-mode test can therefore run the SAME `settings()` / `change()` script on the product batch and on this
object and compare samples directly -- HIP vs oracle, not HIP vs HIP.  Calls that address streams outside the sample
are accepted and ignored (the streams of a batch never exchange anything)."""
import numpy as np


def _midi_to_bin(note):
    # processor_core_2.cc:561-583: (note - 33) * 8 rounded half away from zero, clamped to [1, 447]
    note = min(max(float(note), 0.0), 128.0)
    r = (note - 33.0) * 8.0
    q = int(np.floor(r + 0.5)) if r >= 0 else -int(np.floor(-r + 0.5))
    return min(max(q, 1), 447)


class _Calls:
    """The subset of include/beatrice_batch.h's per-stream setters that test scripts use; `h` is ignored."""

    def __init__(self, owner):
        self.o = owner

    def _each(self, stream, fn):
        o = self.o
        if stream < -1 or stream >= o.B:
            return -1
        for s in (o.sample if stream < 0 else [stream]):
            if s in o.st:
                fn(o.st[s])
        return 0

    def BeatriceBatch_SetTargetSpeaker(self, h, stream, speaker):
        def f(c):
            c["speaker"] = speaker
            c["s1"].set_target_speaker(speaker)   # codebook + additive now, K/V registered: blocks follow one per hop
        return self._each(stream, f)

    def BeatriceBatch_FlushSpeaker(self, h, stream):
        def f(c):
            while c["s1"].set_kv_block():
                pass
        return self._each(stream, f)

    def BeatriceBatch_SetFormantShift(self, h, stream, shift):
        shift = min(max(float(shift), -2.0), 2.0)
        r = shift * 2.0 + 4.0
        idx = int(np.floor(r + 0.5))   # (r >= 0 here: std::round = half away from zero)

        def f(c):
            c["formant"] = idx
            c["s1"].set_formant_index(idx)
        return self._each(stream, f)

    def BeatriceBatch_SetVQNumNeighbors(self, h, stream, k):
        k = min(max(int(k), 0), 8)

        def f(c):
            c["vq_k"] = k
            c["s1"].a.SetVQNumNeighbors(c["s1"].pc, k)
        return self._each(stream, f)

    def BeatriceBatch_SetMinSourcePitch(self, h, stream, note):
        q = _midi_to_bin(note)

        def f(c):
            c["min_q"] = q
            c["s1"].a.SetMinQuantizedPitch(c["s1"].tc, q)
        return self._each(stream, f)

    def BeatriceBatch_SetMaxSourcePitch(self, h, stream, note):
        q = _midi_to_bin(note)

        def f(c):
            c["max_q"] = q
            c["s1"].a.SetMaxQuantizedPitch(c["s1"].tc, q)
        return self._each(stream, f)

    def _pitch(self, stream, key, v):
        def f(c):
            c["pitch"][key] = v
            c["s1"].pitch_params = dict(c["pitch"])
        return self._each(stream, f)

    def BeatriceBatch_SetPitchShift(self, h, stream, v):
        return self._pitch(stream, "shift", min(max(float(v), -24.0), 24.0))

    def BeatriceBatch_SetAverageSourcePitch(self, h, stream, v):
        return self._pitch(stream, "avg", min(max(float(v), 0.0), 128.0))

    def BeatriceBatch_SetIntonationIntensity(self, h, stream, v):
        return self._pitch(stream, "intonation", float(v))

    def BeatriceBatch_SetPitchCorrection(self, h, stream, v):
        return self._pitch(stream, "correction", min(max(float(v), 0.0), 1.0))

    def BeatriceBatch_SetPitchCorrectionType(self, h, stream, t):
        if t not in (0, 1):
            return -1
        return self._pitch(stream, "ctype", int(t))

    def BeatriceBatch_ResetStream(self, h, stream):
        return self._each(stream, lambda c: self.o._fresh(c))


class OracleBatch:
    def __init__(self, bv, oracle_abi, model_dir, n_streams, sample=None, models=None):
        self.bv = bv
        self.B = n_streams
        self.sample = sorted(set(range(n_streams) if sample is None else [s for s in sample if 0 <= s < n_streams]))
        self._own_models = models is None
        self.m = models if models is not None else bv.Models(oracle_abi, model_dir)
        self.h = None
        self.a = _Calls(self)
        self.st = {}
        for s in self.sample:
            c = dict(speaker=0, formant=4, vq_k=0, min_q=1, max_q=bv.PITCH_BINS - 1, pitch={}, s1=None)
            self.st[s] = c
            self._fresh(c, first=True)
        # bv.Batch.apply_defaults: speaker 0 with every K/V block installed, default pitch search range
        self.a.BeatriceBatch_SetMinSourcePitch(None, -1, 33.125)
        self.a.BeatriceBatch_SetMaxSourcePitch(None, -1, 80.875)

    def _fresh(self, c, first=False):
        """New contexts with the stream's settings re-applied and all four K/V blocks of its target speaker installed."""
        if c["s1"] is not None:
            c["s1"].close()
        c["s1"] = self.bv.Stream1(self.m, speaker=c["speaker"], formant_index=c["formant"], vq_k=c["vq_k"], min_q=c["min_q"], max_q=c["max_q"])
        c["s1"].pitch_params = dict(c["pitch"])

    def convert(self, x):
        """x: [B][160] (rows of non-sampled streams are ignored) -> {stream: [240]} for the sampled streams."""
        return {s: self.st[s]["s1"].hop(x[s]) for s in self.sample}

    def convert_rows(self, x):
        """The sampled streams' samples as an array [len(sample)][240], in the order of `self.sample`."""
        out = self.convert(x)
        return np.stack([out[s] for s in self.sample])

    def close(self):
        for c in self.st.values():
            if c["s1"] is not None:
                c["s1"].close()
                c["s1"] = None
        if self._own_models:
            self.m.close()


def pick_streams(B, n=8):
    """A spread of stream indices that covers the corners of every row tiling (first / last row of 16- and 32-row tiles,
    the ragged last tile) plus the streams the test scripts single out (0, 1, 2)."""
    want = [0, B - 1, 15, 16, 1, 2, 31, 32, B // 2, B - 17, 17, 33, B - 2]
    got = []
    for s in want:
        if 0 <= s < B and s not in got:
            got.append(s)
    return sorted(got[:max(n, 3)]) if B > n else list(range(B))


class _Recorder:
    """Runs a test script once without converting anything and notes which streams it addresses."""

    def __init__(self, B):
        self.B, self.h, self.touched = B, None, []
        self.a = self

    def __getattr__(self, name):
        if not name.startswith("BeatriceBatch_"):
            raise AttributeError(name)

        def call(h, stream, *args):
            if isinstance(stream, int) and 0 <= stream < self.B and stream not in self.touched:
                self.touched.append(stream)
            return 0
        return call


def scripted_streams(B, total, change, limit=6):
    """The first `limit` streams a `change(batch, k)` script addresses individually over `total` steps."""
    rec = _Recorder(B)
    for k in range(total):
        change(rec, k)
    return rec.touched[:limit]


def oracle_leg(bv, oracle_abi, model_dir, B, hop_input, total, settings, change, sample):
    """Samples of the sampled streams from independent oracle streams driven by the test's own script:
    `settings(batch)` once, then per step `change(batch, k)` and one hop of `hop_input(k)` ([B][160]).
    Returns (sorted sample, array [total][len(sample)][240])."""
    ob = OracleBatch(bv, oracle_abi, model_dir, B, sample=sample)
    settings(ob)
    out = np.zeros((total, len(ob.sample), bv.OUT_HOP), np.float32)
    for k in range(total):
        change(ob, k)
        out[k] = ob.convert_rows(hop_input(k))
    ob.close()
    return ob.sample, out
