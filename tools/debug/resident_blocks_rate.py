"""Throughput of the any-rate wrapper AROUND the tick pipeline (BeatriceBatch_BindResidentBlocks) at 1 / 2 / 4 hops per step, and of its
per-stream-clock form (BeatriceBatch_BindResidentBlocksRagged, one hop per step): resident host-rate blocks, `calls` calls after a
warm-up, drain inside the timed region.  Prints one JSON line per case (frames = model hops of 10 ms that came out)."""
import ctypes as C
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import make_model  # noqa: E402
from tick_driver import Hip  # noqa: E402

bv = importlib.import_module("beatrice-vst_amd")
product = bv.bind_batch(bv.load_product())
tmp = tempfile.TemporaryDirectory()
make_model.make_model(tmp.name, n_speakers=1)
m = bv.Models(product, tmp.name)
hip = Hip()
B = int(os.environ.get("STREAMS", "256"))
CALLS = int(os.environ.get("CALLS", "600"))


def uniform(sr, block, H):
    batch = bv.Batch(m, B, hops_per_step=H)
    a, h = batch.a, batch.h
    assert a.BeatriceBatch_ConfigureWrapper(h, float(sr)) == 0
    delay = a.BeatriceBatch_ResidentBlocksDelayFor(h, block)
    slots = delay + 8
    rng = np.random.default_rng(7)
    x = (0.1 * rng.standard_normal((slots, B, 1, block))).astype(np.float32)
    d_in, d_out = hip.malloc(x.nbytes), hip.malloc(x.nbytes)
    hip.h2d(d_in, x)
    assert a.BeatriceBatch_BindResidentBlocks(h, d_in, d_out, 1, block, slots) == 0
    for _ in range(3 * delay):
        assert a.BeatriceBatch_ProcessBlocksDevice(h, None, None, 1, block) == 0
    assert a.BeatriceBatch_Synchronize(h) == 0
    t0 = time.perf_counter()
    for _ in range(CALLS):
        assert a.BeatriceBatch_ProcessBlocksDevice(h, None, None, 1, block) == 0
    assert a.BeatriceBatch_Synchronize(h) == 0
    dt = time.perf_counter() - t0
    hops = CALLS * block * 100.0 / sr          # model hops per stream in the timed calls
    print(json.dumps({"mode": "resident blocks around the ticks", "rate": sr, "block": block, "streams": B, "hops_per_step": H, "calls": CALLS,
                      "delay_calls": delay, "ms_per_call": round(1e3 * dt / CALLS, 4), "frames_per_s": round(B * hops / dt, 1)}))
    assert a.BeatriceBatch_BindResidentBlocks(h, None, None, 0, 0, 0) == 0
    batch.close()
    hip.free(d_in)
    hip.free(d_out)


def ragged():
    rates = [(44100.0, 441), (48000.0, 480), (96000.0, 960), (32000.0, 320)]
    batch = bv.Batch(m, B)
    a, h = batch.a, batch.h
    rs = [rates[s % 4][0] for s in range(B)]
    ns = [rates[s % 4][1] for s in range(B)]
    assert a.BeatriceBatch_ConfigureWrapperRates(h, (C.c_double * B)(*rs)) == 0
    stages = a.BeatriceBatch_TickStages(h)
    slots, cap = stages + 8, max(ns)
    rng = np.random.default_rng(7)
    x = (0.1 * rng.standard_normal((slots, B, cap))).astype(np.float32)
    d_in, d_out = hip.malloc(x.nbytes), hip.malloc(x.nbytes)
    hip.h2d(d_in, x)
    assert a.BeatriceBatch_BindResidentBlocksRagged(h, d_in, d_out, 1, cap, slots) == 0
    n_arr = (C.c_int * B)(*ns)
    for _ in range(3 * stages):
        assert a.BeatriceBatch_ProcessBlocksRaggedDevice(h, n_arr) == 0
    assert a.BeatriceBatch_Synchronize(h) == 0
    t0 = time.perf_counter()
    for _ in range(CALLS):
        assert a.BeatriceBatch_ProcessBlocksRaggedDevice(h, n_arr) == 0
    assert a.BeatriceBatch_Synchronize(h) == 0
    dt = time.perf_counter() - t0
    print(json.dumps({"mode": "resident blocks around the ticks, clocks per stream (44.1 / 48 / 96 / 32 kHz, 10 ms blocks)", "streams": B, "hops_per_step": 1,
                      "calls": CALLS, "ms_per_call": round(1e3 * dt / CALLS, 4), "frames_per_s": round(B * CALLS / dt, 1)}))
    assert a.BeatriceBatch_BindResidentBlocksRagged(h, None, None, 0, 0, 0) == 0
    batch.close()
    hip.free(d_in)
    hip.free(d_out)


if os.environ.get("BLOCK"):   # e.g. BLOCK=64 RATE=48000: DAW-sized blocks (several calls per model hop)
    for H in (1, 2, 4):
        uniform(int(os.environ.get("RATE", "48000")), int(os.environ["BLOCK"]), H)
    sys.exit(0)
if os.environ.get("ONLY_H"):   # one case (for a kernel trace)
    uniform(44100, 441, int(os.environ["ONLY_H"]))
    sys.exit(0)
for H in (1, 2, 4):
    uniform(44100, 441, H)
uniform(48000, 480, 4)
ragged()
