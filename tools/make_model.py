#!/usr/bin/env python3
"""Deterministic synthetic model package for the frozen MODEL_SPEC (see MODEL_SPEC.md).

The reference's model packages (`phone_extractor.bin`, `pitch_estimator.bin`,
`waveform_generator.bin`, `embedding_setter.bin`, `speaker_embeddings.bin`; file names from
reference src/common/processor_core_2.cc:301-346) are produced by a closed trainer and their on-disk
layout is unknown.  This tool writes OUR layout: a 16-byte header
(`BTRC`, kind, version, n_floats) followed by little-endian float32 tensors in the fixed order
listed in MODEL_SPEC.md §5.  Weights are a counter-based splitmix64 stream, so the same seed
gives the same bytes on every machine and numpy version.

Usage: python tools/make_model.py OUT_DIR [--speakers N] [--seed S]
"""
import argparse
import os
import struct

import numpy as np

MAGIC = 0x43525442  # 'BTRC' little endian
VERSION = 1
KIND = {"phone_extractor": 1, "pitch_estimator": 2, "waveform_generator": 3,
        "embedding_setter": 4, "speaker_embeddings": 5}

IN_HOP, OUT_HOP = 160, 240
PHONE_CH, HID, PITCH_BINS = 128, 256, 448
CODEBOOK, KV_LEN, KV_CH, N_BLOCKS = 512, 384, 128, 4
FFT_N = 1024
UP_RATES = (5, 4, 4, 3)
UP_CH = (256, 128, 64, 32, 16)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


class Stream:
    """Counter-based uniform generator: tensor t, element i -> U[-1, 1)."""

    def __init__(self, seed):
        self.seed = np.uint64(seed)
        self.tensor_id = 0

    def uniform(self, n):
        self.tensor_id += 1
        with np.errstate(over="ignore"):
            base = _splitmix64(np.array([self.seed ^ np.uint64(self.tensor_id * 0x1000003)],
                                        dtype=np.uint64))[0]
            idx = np.arange(n, dtype=np.uint64) + base
            bits = _splitmix64(idx) >> np.uint64(40)  # 24 random bits
        return (bits.astype(np.float64) / float(1 << 23) - 1.0).astype(np.float32)


def dense(rs, k, n, gain=1.0):
    """[K][N] weight, U(-a, a) with a = gain*sqrt(3/K) (unit-variance-preserving at gain 1)."""
    a = gain * np.sqrt(3.0 / k)
    return (rs.uniform(k * n) * np.float32(a)).astype(np.float32)


def bias(rs, n, scale=0.02):
    return (rs.uniform(n) * np.float32(scale)).astype(np.float32)


def write_file(path, kind, tensors):
    payload = np.concatenate([np.asarray(t, dtype=np.float32).ravel() for t in tensors])
    with open(path, "wb") as f:
        f.write(struct.pack("<IIII", MAGIC, KIND[kind], VERSION, payload.size))
        f.write(payload.astype("<f4").tobytes())
    return payload.size


def phone_extractor(rs):
    t = []
    t += [dense(rs, 10, 64, 6.0), bias(rs, 64)]                       # F1 k10 s5 1->64
    t += [dense(rs, 8 * 64, 128, 1.6), bias(rs, 128)]                 # F2 k8 s4
    t += [dense(rs, 4 * 128, 256, 1.6), bias(rs, 256)]                # F3 k4 s2
    t += [dense(rs, 4 * 256, 256, 1.6), bias(rs, 256)]                # F4 k4 s2
    t += [dense(rs, 4 * 256, 256, 1.6), bias(rs, 256)]                # F5 k4 s2
    for _ in range(4):                                                # RB k5
        t += [dense(rs, 5 * 256, 256, 1.0), bias(rs, 256)]
    t += [dense(rs, 256, 768, 1.0), dense(rs, 256, 768, 0.9), bias(rs, 768, 0.1), bias(rs, 768, 0.1)]
    t += [dense(rs, 256, PHONE_CH, 1.5), bias(rs, PHONE_CH)]
    return t


def pitch_estimator(rs):
    i = np.arange(FFT_N, dtype=np.float64)
    window = (0.5 - 0.5 * np.cos(2.0 * np.pi * i / FFT_N)).astype(np.float32)
    k = np.arange(FFT_N // 2, dtype=np.float64)
    tw = np.stack([np.cos(2.0 * np.pi * k / FFT_N), -np.sin(2.0 * np.pi * k / FFT_N)], axis=1)
    t = [window, tw.astype(np.float32)]
    t += [dense(rs, 3 * 512, 128, 0.5), bias(rs, 128)]                # P1
    t += [dense(rs, 3 * 128, 128, 1.0), bias(rs, 128)]                # P2
    t += [dense(rs, 3 * 128, 128, 1.0), bias(rs, 128)]                # P3
    t += [dense(rs, 128, 384, 1.0), dense(rs, 128, 384, 0.9), bias(rs, 384, 0.1), bias(rs, 384, 0.1)]
    t += [dense(rs, 128, PITCH_BINS, 6.0), bias(rs, PITCH_BINS, 0.5)]
    t += [dense(rs, 128, 1, 2.0), bias(rs, 1, 0.1)]                   # voicing
    return t


def waveform_generator(rs):
    t = [dense(rs, PHONE_CH, HID, 1.0), bias(rs, HID)]
    t += [(rs.uniform(PITCH_BINS * HID) * np.float32(0.5)).astype(np.float32)]  # pitch embedding
    t += [dense(rs, 4, HID, 0.5)]
    for _ in range(N_BLOCKS):
        t += [dense(rs, 3 * HID, HID, 1.4), bias(rs, HID)]            # dilated conv
        t += [dense(rs, HID, HID, 0.7), bias(rs, HID)]                # 1x1
        t += [dense(rs, HID, HID, 1.0), bias(rs, HID)]                # q
        t += [dense(rs, HID, HID, 0.7), bias(rs, HID)]                # o
    for s in range(4):
        cin, cout, r = UP_CH[s], UP_CH[s + 1], UP_RATES[s]
        t += [dense(rs, 2 * cin, r * cout, 1.0), np.tile(bias(rs, cout), r)]  # polyphase ConvT
        t += [dense(rs, 3 * cout, cout, 0.7), bias(rs, cout)]         # res d=1
        t += [dense(rs, 3 * cout, cout, 0.7), bias(rs, cout)]         # res d=3
    t += [dense(rs, 7 * 16, 1, 0.45), bias(rs, 1, 0.0)]
    return t


def embedding_setter(rs):
    t = [dense(rs, HID, HID, 0.5), bias(rs, HID), dense(rs, HID, HID, 0.5), bias(rs, HID)]
    for _ in range(N_BLOCKS):
        t += [dense(rs, KV_CH, HID, 1.0), bias(rs, HID), dense(rs, KV_CH, HID, 1.0), bias(rs, HID)]
    return t


def unit_rows(rs, rows, dim):
    x = rs.uniform(rows * dim).reshape(rows, dim).astype(np.float64)
    x *= np.sqrt(dim) / np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def speaker_embeddings(rs, n_speakers):
    t = [unit_rows(rs, 9, HID)]
    for _ in range(n_speakers):
        t += [unit_rows(rs, CODEBOOK, PHONE_CH) * np.float32(1.0),
              unit_rows(rs, 1, HID), unit_rows(rs, KV_LEN, KV_CH)]
    return t


# ---- legacy generations (MODEL_SPEC section 6): 2.0.0-beta.1 (Beatrice20b1_*) and 2.0.0-alpha.2 (Beatrice20a2_*) ----------
LEGACY_PHONE_CH, LEGACY_PITCH_BINS = 256, 384
LEGACY_KIND = {"phone_extractor": 11, "pitch_estimator": 12, "waveform_generator": 13, "embedding_rows": 15}
LEGACY_VERSION = {"2.0.0-beta.1": 1, "2.0.0-alpha.2": 0}


def legacy_phone_extractor(rs):
    t = phone_extractor(rs)[:-2]                                      # front end, residual blocks, GRU as rc.0
    t += [dense(rs, 256, LEGACY_PHONE_CH, 1.5), bias(rs, LEGACY_PHONE_CH)]
    return t


def legacy_pitch_estimator(rs):
    t = pitch_estimator(rs)
    head = [dense(rs, 128, LEGACY_PITCH_BINS, 6.0), bias(rs, LEGACY_PITCH_BINS, 0.5)]
    return t[:-4] + head + t[-2:]                                     # logits over 384 bins; voicing vector as rc.0


def legacy_waveform_generator(rs):
    t = [dense(rs, LEGACY_PHONE_CH, HID, 1.0), bias(rs, HID)]
    t += [(rs.uniform(LEGACY_PITCH_BINS * HID) * np.float32(0.5)).astype(np.float32)]
    t += [dense(rs, 4, HID, 0.5)]
    for _ in range(N_BLOCKS):
        t += [dense(rs, 3 * HID, HID, 1.4), bias(rs, HID)]            # dilated conv
        t += [dense(rs, HID, HID, 0.7), bias(rs, HID)]                # 1x1 (no attention half in these generations)
    for s in range(4):
        cin, cout, r = UP_CH[s], UP_CH[s + 1], UP_RATES[s]
        t += [dense(rs, 2 * cin, r * cout, 1.0), np.tile(bias(rs, cout), r)]
        t += [dense(rs, 3 * cout, cout, 0.7), bias(rs, cout)]
        t += [dense(rs, 3 * cout, cout, 0.7), bias(rs, cout)]
    t += [dense(rs, 7 * 16, 1, 0.45), bias(rs, 1, 0.0)]
    return t


def write_legacy_file(path, kind, tensors):
    payload = np.concatenate([np.asarray(t, dtype=np.float32).ravel() for t in tensors])
    with open(path, "wb") as f:
        f.write(struct.pack("<IIII", MAGIC, LEGACY_KIND[kind], VERSION, payload.size))
        f.write(payload.astype("<f4").tobytes())
    return payload.size


def make_model_legacy(out_dir, n_speakers=2, seed=0x20B1, version="2.0.0-beta.1"):
    """A model package of a legacy generation: the five files the reference's ProcessorCore0 / ProcessorCore1 read
    (reference src/common/processor_core_1.cc:165-216): phone_extractor.bin, pitch_estimator.bin, waveform_generator.bin,
    speaker_embeddings.bin ([n][256]) and formant_shift_embeddings.bin ([9][256], read with the same reader)."""
    assert version in LEGACY_VERSION
    os.makedirs(out_dir, exist_ok=True)
    sizes = {}
    for name, fn in (("phone_extractor", legacy_phone_extractor), ("pitch_estimator", legacy_pitch_estimator),
                     ("waveform_generator", legacy_waveform_generator)):
        rs = Stream(seed + LEGACY_KIND[name])
        sizes[name] = write_legacy_file(os.path.join(out_dir, name + ".bin"), name, fn(rs))
    rs = Stream(seed + LEGACY_KIND["embedding_rows"])
    sizes["speaker_embeddings"] = write_legacy_file(os.path.join(out_dir, "speaker_embeddings.bin"), "embedding_rows",
                                                    [unit_rows(rs, n_speakers, HID)])
    sizes["formant_shift_embeddings"] = write_legacy_file(os.path.join(out_dir, "formant_shift_embeddings.bin"), "embedding_rows",
                                                          [unit_rows(rs, 9, HID) * np.float32(0.25)])
    with open(os.path.join(out_dir, "model.toml"), "w") as f:
        f.write('[model]\nversion = "%s"\nname = "synthetic-legacy-%x"\n' % (version, seed))
        f.write('description = "deterministic synthetic weights for MODEL_SPEC section 6"\n')
        for s in range(n_speakers):
            f.write('\n[voice.%d]\nname = "spk%d"\ndescription = "synthetic speaker %d"\naverage_pitch = 52.0\n' % (s, s, s))
            f.write('[voice.%d.portrait]\npath = ""\ndescription = ""\n' % s)
    return sizes


def make_model(out_dir, n_speakers=2, seed=0x20C0):
    os.makedirs(out_dir, exist_ok=True)
    sizes = {}
    for name, fn in (("phone_extractor", phone_extractor), ("pitch_estimator", pitch_estimator),
                     ("waveform_generator", waveform_generator),
                     ("embedding_setter", embedding_setter)):
        rs = Stream(seed + KIND[name])
        sizes[name] = write_file(os.path.join(out_dir, name + ".bin"), name, fn(rs))
    rs = Stream(seed + KIND["speaker_embeddings"])
    sizes["speaker_embeddings"] = write_file(os.path.join(out_dir, "speaker_embeddings.bin"),
                                             "speaker_embeddings", speaker_embeddings(rs, n_speakers))
    with open(os.path.join(out_dir, "model.toml"), "w") as f:
        f.write('[model]\nversion = "2.0.0-rc.0"\nname = "synthetic-%x"\n' % seed)
        f.write('description = "deterministic synthetic weights for MODEL_SPEC v%d"\n' % VERSION)
        # every key the reference's ModelConfig reader asks for (reference src/common/model_config.h:73-136):
        # voice ids from 0, contiguous; name, description, average_pitch in [0, 128], portrait.path, portrait.description
        for s in range(n_speakers):
            f.write('\n[voice.%d]\nname = "spk%d"\ndescription = "synthetic speaker %d"\naverage_pitch = 52.0\n' % (s, s, s))
            f.write('[voice.%d.portrait]\npath = ""\ndescription = ""\n' % s)
    return sizes


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--speakers", type=int, default=2)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x20C0)
    ap.add_argument("--legacy", choices=sorted(LEGACY_VERSION), default=None, help="write a legacy-generation package instead")
    a = ap.parse_args()
    sizes = make_model_legacy(a.out_dir, a.speakers, a.seed, a.legacy) if a.legacy else make_model(a.out_dir, a.speakers, a.seed)
    for k, v in sizes.items():
        print("%-20s %9d floats" % (k, v))
