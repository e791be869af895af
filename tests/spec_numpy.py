"""An INDEPENDENT restatement of MODEL_SPEC.md sections 3-4 in numpy float64, written from the spec's text (not from
oracle/beatrice_oracle.c) and deliberately in a different form: OFFLINE, whole-utterance tensors instead of per-hop
streaming state, library FFT / tanh / exp instead of the spec's float32 polynomial sequences, matrix products instead
of segmented FMA chains.  tests/test_cpu_spec_crosscheck.py compares the oracle's hop-by-hop float32 results with it;
agreement (to float32 rounding) says two independently written readings of the spec coincide -- gate order of the GRU,
the causal index formula, the polyphase column order of the transposed convolutions, the attention scaling -- and that
streaming equals offline.  Test infrastructure only."""
import os
import struct

import numpy as np

IN_HOP, OUT_HOP, HID, PHONE_CH, BINS = 160, 240, 256, 128, 448
CODEBOOK, KV_LEN, KV_CH, N_BLOCKS, FFT_N = 512, 384, 128, 4, 1024


def _read(path, kind):
    raw = open(path, "rb").read()
    magic, k, version, n = struct.unpack("<IIII", raw[:16])
    assert magic == 0x43525442 and k == kind and version == 1 and len(raw) == 16 + 4 * n
    return np.frombuffer(raw, "<f4", offset=16).astype(np.float64)


class _Cursor:
    def __init__(self, flat):
        self.flat, self.pos = flat, 0

    def take(self, *shape):
        n = int(np.prod(shape))
        out = self.flat[self.pos:self.pos + n].reshape(shape)
        self.pos += n
        return out

    def done(self):
        assert self.pos == self.flat.size


# ---- MODEL_SPEC 2.1 (mathematical definitions; the float32 polynomial forms approximate these) --------------
def gelu(x):
    return 0.5 * x * (1.0 + np.tanh(0.7978845608 * (x + 0.044715 * x ** 3)))


def lrelu(x):
    return np.where(x > 0, x, 0.1 * x)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ---- MODEL_SPEC 3.1 --------------------------------------------------------------------------------------------
def conv(x, w, b, k, stride=1, dil=1, pre=None):
    """x [T_in][cin] -> [T_in // stride][cout]; output frame t reads input frames (t+1)*stride-1-(k-1-j)*dil,
    frames before the start are zero; weight rows are tap*cin + c."""
    t_in, cin = x.shape
    if pre is not None:
        x = pre(x)
    t_out = t_in // stride
    y = np.tile(b, (t_out, 1))
    for j in range(k):
        idx = (np.arange(t_out) + 1) * stride - 1 - (k - 1 - j) * dil
        tap = np.where((idx >= 0)[:, None], x[np.clip(idx, 0, None)], 0.0)
        y = y + tap @ w[j * cin:(j + 1) * cin]
    return y


def conv_transpose(x, w, b, rate, pre):
    """ConvT(cin -> cout, rate r) in the spec's polyphase form: Conv(cin -> r*cout, k=2); output frame t*r + rho is
    columns rho*cout .. rho*cout+cout-1 of row t (the bias is stored expanded)."""
    rows = conv(x, w, b, 2, pre=pre)
    return rows.reshape(x.shape[0] * rate, -1)


# ---- MODEL_SPEC 3.2 --------------------------------------------------------------------------------------------
def gru(xs, wih, whh, bih, bhh):
    hdim = whh.shape[0]
    h = np.zeros(hdim)
    out = []
    for x in xs:
        gi, gh = x @ wih + bih, h @ whh + bhh
        r = sigmoid(gi[:hdim] + gh[:hdim])
        z = sigmoid(gi[hdim:2 * hdim] + gh[hdim:2 * hdim])
        n = np.tanh(gi[2 * hdim:] + r * gh[2 * hdim:])
        h = n + z * (h - n)
        out.append(h)
    return np.array(out)


# ---- MODEL_SPEC 4.1 --------------------------------------------------------------------------------------------
class PhoneExtractor:
    FRONT = ((10, 1, 64, 5), (8, 64, 128, 4), (4, 128, 256, 2), (4, 256, 256, 2), (4, 256, 256, 2))

    def __init__(self, model_dir, out_ch=PHONE_CH, kind=1):   # (legacy generations, MODEL_SPEC 6.1: out_ch 256, file kind 11)
        c = _Cursor(_read(os.path.join(model_dir, "phone_extractor.bin"), kind))
        self.front = [(c.take(k * cin, cout), c.take(cout), k, s) for k, cin, cout, s in self.FRONT]
        self.res = [(c.take(5 * 256, 256), c.take(256)) for _ in range(4)]
        self.wih, self.whh, self.bih, self.bhh = c.take(256, 768), c.take(256, 768), c.take(768), c.take(768)
        self.wo, self.bo = c.take(256, out_ch), c.take(out_ch)
        c.done()

    def __call__(self, audio, codebook=None, k=0):
        x = np.asarray(audio, np.float64).reshape(-1, 1)
        for w, b, ksz, s in self.front:
            x = gelu(conv(x, w, b, ksz, stride=s))
        for w, b in self.res:
            x = x + gelu(conv(x, w, b, 5))
        x = gru(x, self.wih, self.whh, self.bih, self.bhh) @ self.wo + self.bo
        if k > 0 and codebook is not None:
            cb = np.asarray(codebook, np.float64)
            d = (cb * cb).sum(1)[None, :] - 2.0 * (x @ cb.T)
            order = np.argsort(d, axis=1, kind="stable")[:, :k]
            return cb[order].sum(1) / k, x, d
        return x, x, None


# ---- MODEL_SPEC 4.2 --------------------------------------------------------------------------------------------
class PitchEstimator:
    def __init__(self, model_dir, bins=BINS, kind=2):   # (legacy generations, MODEL_SPEC 6.2: 384 bins, file kind 12)
        self.bins = bins
        c = _Cursor(_read(os.path.join(model_dir, "pitch_estimator.bin"), kind))
        self.window, self.twiddle = c.take(FFT_N), c.take(FFT_N // 2, 2)
        self.p1 = (c.take(3 * 512, 128), c.take(128))
        self.res = [(c.take(3 * 128, 128), c.take(128)) for _ in range(2)]
        self.wih, self.whh, self.bih, self.bhh = c.take(128, 384), c.take(128, 384), c.take(384), c.take(384)
        self.wo, self.bo = c.take(128, bins), c.take(bins)
        self.v, self.vb = c.take(128), c.take(1)
        c.done()

    def __call__(self, audio, lo=1, hi=None):
        BINS = self.bins
        hi = BINS - 1 if hi is None else hi
        audio = np.asarray(audio, np.float64)
        n = audio.size // IN_HOP
        padded = np.concatenate([np.zeros(FFT_N - IN_HOP), audio])
        frames = np.stack([padded[t * IN_HOP:t * IN_HOP + FFT_N] for t in range(n)]) * self.window
        spec = np.fft.fft(frames, axis=1)[:, :512]
        feat = 0.5 * np.log(spec.real ** 2 + spec.imag ** 2 + 1e-5)
        x = gelu(conv(feat, *self.p1, 3))
        for w, b in self.res:
            x = x + gelu(conv(x, w, b, 3))
        h = gru(x, self.wih, self.whh, self.bih, self.bhh)
        logits = h @ self.wo + self.bo
        lo, hi = max(1, min(BINS - 1, lo)), max(1, min(BINS - 1, hi))
        hi = max(hi, lo)
        bins = lo + np.argmax(logits[:, lo:hi + 1], axis=1)
        e = np.exp(logits - logits.max(1, keepdims=True))
        f0 = e[np.arange(n), bins] / e.sum(1)
        f1 = 0.1 * np.log((audio.reshape(n, IN_HOP) ** 2).sum(1) / 160.0 + 1e-8)
        f2 = np.clip((bins - np.concatenate([[0], bins[:-1]])) / 8.0, -1.0, 1.0)
        f3 = sigmoid(h @ self.v + self.vb[0])
        return bins, np.stack([f0, f1, f2, f3], axis=1), logits


# ---- MODEL_SPEC 4.3 + 4.4 ----------------------------------------------------------------------------------------
class WaveformGenerator:
    UP = ((256, 128, 5), (128, 64, 4), (64, 32, 4), (32, 16, 3))

    def __init__(self, model_dir):
        c = _Cursor(_read(os.path.join(model_dir, "waveform_generator.bin"), 3))
        self.wi, self.bi = c.take(PHONE_CH, HID), c.take(HID)
        self.pitch_emb, self.wf = c.take(BINS, HID), c.take(4, HID)
        self.blocks = [dict(c1=(c.take(3 * HID, HID), c.take(HID)), c2=(c.take(HID, HID), c.take(HID)),
                            q=(c.take(HID, HID), c.take(HID)), o=(c.take(HID, HID), c.take(HID))) for _ in range(N_BLOCKS)]
        self.up = []
        for cin, cout, r in self.UP:
            self.up.append(dict(t=(c.take(2 * cin, r * cout), c.take(r * cout)), a=(c.take(3 * cout, cout), c.take(cout)),
                                b=(c.take(3 * cout, cout), c.take(cout)), r=r))
        self.wfin, self.bfin = c.take(7 * 16, 1), c.take(1)
        c.done()
        e = _Cursor(_read(os.path.join(model_dir, "embedding_setter.bin"), 4))
        self.add = (e.take(HID, HID), e.take(HID))
        self.frm = (e.take(HID, HID), e.take(HID))
        self.kv = [dict(k=(e.take(KV_CH, HID), e.take(HID)), v=(e.take(KV_CH, HID), e.take(HID))) for _ in range(N_BLOCKS)]
        e.done()

    def __call__(self, phone, bins, feat, additive, formant, kv_raw):
        phone, feat = np.asarray(phone, np.float64), np.asarray(feat, np.float64)
        a = np.asarray(additive, np.float64) @ self.add[0] + self.add[1]
        f = np.asarray(formant, np.float64) @ self.frm[0] + self.frm[1]
        kv_raw = np.asarray(kv_raw, np.float64)
        e = (self.pitch_emb[np.asarray(bins)] + feat @ self.wf) + (a + f)
        x = (phone @ self.wi + self.bi) + e
        for i, (blk, d) in enumerate(zip(self.blocks, (1, 2, 4, 8))):
            keys = kv_raw @ self.kv[i]["k"][0] + self.kv[i]["k"][1]      # [384][256]
            vals = kv_raw @ self.kv[i]["v"][0] + self.kv[i]["v"][1]
            h = gelu(conv(x, *blk["c1"], 3, dil=d))
            xa = x + (h @ blk["c2"][0] + blk["c2"][1])
            q = xa @ blk["q"][0] + blk["q"][1]
            s = (q @ keys.T) * 0.0625
            p = np.exp(s - s.max(1, keepdims=True))
            o = (p @ vals) / p.sum(1, keepdims=True)
            x = xa + (o @ blk["o"][0] + blk["o"][1])
        y = x
        for st in self.up:
            y = conv_transpose(y, *st["t"], st["r"], pre=lrelu)
            y = y + conv(y, *st["a"], 3, dil=1, pre=lrelu)
            y = y + conv(y, *st["b"], 3, dil=3, pre=lrelu)
        return np.tanh(conv(y, self.wfin, self.bfin, 7, pre=lrelu))[:, 0]


# ---- MODEL_SPEC 6.3: the waveform generator of the legacy generations ------------------------------------------------
class LegacyWaveformGenerator:
    def __init__(self, model_dir):
        c = _Cursor(_read(os.path.join(model_dir, "waveform_generator.bin"), 13))
        self.wi, self.bi = c.take(256, HID), c.take(HID)
        self.pitch_emb, self.wf = c.take(384, HID), c.take(4, HID)
        self.blocks = [dict(c1=(c.take(3 * HID, HID), c.take(HID)), c2=(c.take(HID, HID), c.take(HID))) for _ in range(N_BLOCKS)]
        self.up = []
        for cin, cout, r in WaveformGenerator.UP:
            self.up.append(dict(t=(c.take(2 * cin, r * cout), c.take(r * cout)), a=(c.take(3 * cout, cout), c.take(cout)),
                                b=(c.take(3 * cout, cout), c.take(cout)), r=r))
        self.wfin, self.bfin = c.take(7 * 16, 1), c.take(1)
        c.done()

    def __call__(self, phone, bins, feat, speaker):
        """speaker: [frames][256] (the vector the host hands over per hop: speaker + formant-shift embedding)"""
        phone, feat = np.asarray(phone, np.float64), np.asarray(feat, np.float64)
        e = (self.pitch_emb[np.asarray(bins)] + feat @ self.wf) + np.asarray(speaker, np.float64)
        x = (phone @ self.wi + self.bi) + e
        for blk, d in zip(self.blocks, (1, 2, 4, 8)):
            h = gelu(conv(x, *blk["c1"], 3, dil=d))
            x = x + (h @ blk["c2"][0] + blk["c2"][1])
        y = x
        for st in self.up:
            y = conv_transpose(y, *st["t"], st["r"], pre=lrelu)
            y = y + conv(y, *st["a"], 3, dil=1, pre=lrelu)
            y = y + conv(y, *st["b"], 3, dil=3, pre=lrelu)
        return np.tanh(conv(y, self.wfin, self.bfin, 7, pre=lrelu))[:, 0]


def read_rows(path):
    """[n][256] embedding rows of a legacy package (speaker_embeddings.bin, formant_shift_embeddings.bin)"""
    return _read(path, 15).reshape(-1, HID)
