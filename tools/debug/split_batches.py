"""Experiment: 256 streams as G independent tick-pipelined batches of 256/G streams, each on its own HIP stream,
ticks enqueued round-robin.  Do two launches overlap each other's ragged ends?   usage: split_batches.py G [steps]"""
import importlib
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

torch.cuda.init()
bv = importlib.import_module("beatrice-vst_amd")
import make_model  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
total = 256
B = total // G
product = bv.bind_batch(bv.load_product())
tmp = tempfile.TemporaryDirectory()
make_model.make_model(tmp.name, n_speakers=1)
m = bv.Models(product, tmp.name)
n_cycle = 64
batches, keep = [], []
for g in range(G):
    b = bv.Batch(m, B, max_speakers=2)
    audio = np.stack([bv.synth_audio(160 * n_cycle, seed=g * 1000 + s) for s in range(B)])
    d_in = torch.from_numpy(np.ascontiguousarray(audio.reshape(B, n_cycle, 160).transpose(1, 0, 2))).cuda()
    d_out = torch.zeros((n_cycle, B, 240), dtype=torch.float32, device="cuda")
    assert product.BeatriceBatch_BindResidentIO(b.h, d_in.data_ptr(), d_out.data_ptr(), n_cycle) == 0
    assert product.BeatriceBatch_EnableTickPipeline(b.h, 1) == 0
    batches.append(b)
    keep.append((d_in, d_out))


def run(n):
    for _ in range(n):
        for b in batches:
            product.BeatriceBatch_ConvertFramesDevice(b.h, None, None)
    for b in batches:
        product.BeatriceBatch_Synchronize(b.h)


run(60)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("G=%d x %d streams: %.4f ms per step of all, %.2f M frames/s, rms %.4f" % (G, B, dt / steps * 1e3, total * steps / dt / 1e6,
                                                                        float(keep[0][1].pow(2).mean().sqrt())))
