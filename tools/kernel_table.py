#!/usr/bin/env python3
"""Print the per-kernel table of one hop (BeatriceBatch_ProfileKernels) for a given batch size.
Usage (on a GPU box): python tools/kernel_table.py [B] [repeats]"""
import importlib.util
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("beatrice_vst_amd", os.path.join(REPO, "beatrice-vst_amd", "__init__.py"))
bv = importlib.util.module_from_spec(spec)
sys.modules["beatrice_vst_amd"] = bv
spec.loader.exec_module(bv)
sys.path.insert(0, os.path.join(REPO, "tools"))
import make_model  # noqa: E402
import numpy as np  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
product = bv.bind_batch(bv.load_product())
with tempfile.TemporaryDirectory() as d:
    make_model.make_model(d, n_speakers=1)
    m = bv.Models(product, d)
    batch = bv.Batch(m, B)
    x = np.stack([bv.synth_audio(160 * 8, seed=s) for s in range(B)]).reshape(B, 8, 160)
    for h in range(8):
        batch.convert(x[:, h])
    ms = batch.time_steps(100)
    rows = batch.profile_kernels(repeats=reps)
    tot = sum(r["mean_us"] * r["launches"] for r in rows)
    print("B=%d  graph step %.1f us  (%.0f frames/s);  sum of bracketed kernel times %.1f us over %d launches"
          % (B, ms * 10, B * 100 / (ms * 1e-3), tot, sum(r["launches"] for r in rows)))
    print("%-20s %3s %9s %9s %9s %8s" % ("kernel", "n", "us/launch", "GFLOP/s", "GB/s", "MFLOP"))
    for r in rows:
        print("%-20s %3d %9.2f %9.0f %9.0f %8.1f" % (r["name"], r["launches"], r["mean_us"],
                                                      r["flops"] / r["mean_us"] / 1e3, r["bytes"] / r["mean_us"] / 1e3,
                                                      r["flops"] / 1e6))
    batch.close()
    m.close()
