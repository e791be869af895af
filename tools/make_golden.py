#!/usr/bin/env python3
"""Mint the golden fixtures under tests/golden/.

  wrapper_*.npz  : produced by the REFERENCE's own DSP headers (oracle/_ref/libref_wrapper.so, built
                   from /root/reference/src/common/{resample,gain}.h by oracle/Makefile).  Data only.
  core_selftest.npz : produced by this repo's oracle for the frozen MODEL_SPEC + synthetic weights
                   (a regression pin of our own spec; NOT reference-derived -- the reference's neural
                   core is closed).

Run in the container that has /root/reference:  make -C oracle && python tools/make_golden.py
"""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tools"))
import wrapperlib  # noqa: E402
import make_model  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
RATES = [16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000, 192000]


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = wrapperlib.ref_wrapper()
    if ref is None:
        raise SystemExit("oracle/_ref/libref_wrapper.so missing: run `make -C oracle` where /root/reference exists")
    # G1: whole Process() chain with the deterministic stub hop, 0.1 s per host rate, block 441
    g1 = {}
    for sr in RATES:
        x = wrapperlib.test_signal(int(0.1 * sr), sr, seed=sr)
        g1["in_%d" % sr] = x
        g1["out_%d" % sr] = ref.run_chain(sr, x, 441)
    np.savez_compressed(os.path.join(GOLD, "wrapper_chain.npz"), **g1)
    # G2: Stern-Brocot fractions
    ratios = [hi / lo for hi in RATES + [48000, 11025, 8000, 47999, 50000, 384000] for lo in (48000.0,)]
    ratios = sorted(set([r if r >= 1 else 1.0 / r for r in ratios] + [1.0, 1.5, 2.0, 3.0, np.pi, np.e, 1000.0 / 999.0, 7.25]))
    fr = np.array([ref.fraction(r) for r in ratios], np.int32)
    np.savez_compressed(os.path.join(GOLD, "wrapper_fraction.npz"), ratio=np.array(ratios, np.float64), frac=fr)
    # G3: gain ramps: +6, -60, +20, 0 dB steps
    g3 = {}
    for sr in (16000, 48000, 96000):
        n = int(0.12 * sr)
        x = wrapperlib.test_signal(n, sr, seed=7 * sr)
        ev = [(0, 6.0), (n // 4, -60.0), (n // 2, 20.0), (3 * n // 4, 0.0)]
        g3["in_%d" % sr] = x
        g3["out_%d" % sr] = ref.gain_trace(sr, x, ev)
        g3["ev_%d" % sr] = np.array(ev, np.float64)
    # gains through the chain as well (input and output gain moving during streaming)
    x = wrapperlib.test_signal(4800, 48000, seed=99)
    g3["chain_in"] = x
    g3["chain_out"] = ref.run_chain(48000, x, 480, in_gain_events=[(0, -6.0), (2400, 3.0)], out_gain_events=[(960, 6.0)])
    np.savez_compressed(os.path.join(GOLD, "wrapper_gain.npz"), **g3)

    # core self-test (our own spec): 16 hops, speaker switch at hop 6, VQ k=2
    import importlib.util
    spec = importlib.util.spec_from_file_location("beatrice_vst_amd", os.path.join(REPO, "beatrice-vst_amd", "__init__.py"))
    bv = importlib.util.module_from_spec(spec)
    sys.modules["beatrice_vst_amd"] = bv
    spec.loader.exec_module(bv)
    oracle = bv.Abi(os.path.join(REPO, "oracle", "libbeatrice_oracle.so"))
    with tempfile.TemporaryDirectory() as d:
        make_model.make_model(d, n_speakers=3)
        m = bv.Models(oracle, d)
        s = bv.Stream1(m, speaker=0, vq_k=2)
        x = bv.synth_audio(160 * 16, seed=42)
        outs, phones, qs, feats = [], [], [], []
        for i in range(16):
            if i == 6:
                s.set_target_speaker(2)
            o, ph, q, f, _ = s.hop(x[i * 160:(i + 1) * 160], return_all=True)
            outs.append(o); phones.append(ph); qs.append(q); feats.append(f)
        s.close()
        m.close()
    np.savez_compressed(os.path.join(GOLD, "core_selftest.npz"), audio=x, out=np.array(outs), phone=np.array(phones),
                        q=np.array(qs, np.int32), feat=np.array(feats), model_seed=np.array([0x20C0]), speakers=np.array([3]))
    for f in sorted(os.listdir(GOLD)):
        print("%-24s %7d bytes" % (f, os.path.getsize(os.path.join(GOLD, f))))


if __name__ == "__main__":
    main()
