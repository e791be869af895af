"""Two (or L) tick pipelines of B / L streams each on their own HIP streams, fed alternately: do their launches overlap
(the tail of one lane's tick under the head of the other's)?   usage: two_lanes.py [B_total] [lanes]"""
import ctypes, importlib, os, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np
import torch
torch.cuda.init()
import make_model
bv = importlib.import_module("beatrice-vst_amd")
product = bv.bind_batch(bv.load_product())
tmp = tempfile.TemporaryDirectory(); make_model.make_model(tmp.name, n_speakers=1)
m = bv.Models(product, tmp.name)
BT = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = 64
B = BT // L
lanes = []
for l in range(L):
    batch = bv.Batch(m, B)
    d_in = torch.randn((n, B, 160), device="cuda") * 0.1
    d_out = torch.zeros((n, B, 240), device="cuda")
    assert product.BeatriceBatch_BindResidentIO(batch.h, d_in.data_ptr(), d_out.data_ptr(), n) == 0
    assert product.BeatriceBatch_EnableTickPipeline(batch.h, 1) == 0
    lanes.append((batch, d_in, d_out))
for _ in range(60):
    for batch, _, _ in lanes:
        product.BeatriceBatch_ConvertFramesDevice(batch.h, None, None)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(400):
        for batch, _, _ in lanes:
            product.BeatriceBatch_ConvertFramesDevice(batch.h, None, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 400 * 1e6
    print("%d lanes x %d streams: %.2f us per step of %d streams = %.3f M frames/s" % (L, B, dt, BT, BT / dt))
