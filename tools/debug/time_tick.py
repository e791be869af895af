"""BeatriceBatch_TimeTickLaunch called repeatedly (is its figure stable, does it match the loop's period?)"""
import ctypes, importlib, os, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np
import torch
torch.cuda.init()
import make_model
bv = importlib.import_module("beatrice-vst_amd")
product = bv.bind_batch(bv.load_product())
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1   # speakers (fourth argument): stream s on speaker s mod S, k-NN 4 when S > 1 (configs[3]'s shape, no switches)
tmp = tempfile.TemporaryDirectory(); make_model.make_model(tmp.name, n_speakers=S)
m = bv.Models(product, tmp.name)
B, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256), 64
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # hops per step (third argument; the second: "ragged" or "-")
batch = bv.Batch(m, B, hops_per_step=H)
if S > 1:
    for s_ in range(B):
        product.BeatriceBatch_SetTargetSpeaker(batch.h, s_, s_ % S)
    product.BeatriceBatch_FlushSpeaker(batch.h, -1)
    product.BeatriceBatch_SetVQNumNeighbors(batch.h, -1, 4)
d_in = torch.randn((n, B, H * 160), device="cuda") * 0.1
d_out = torch.zeros((n, B, H * 240), device="cuda")
assert product.BeatriceBatch_BindResidentIO(batch.h, d_in.data_ptr(), d_out.data_ptr(), n) == 0
assert product.BeatriceBatch_EnableTickPipeline(batch.h, 1) == 0
if len(sys.argv) > 2 and sys.argv[2] == "ragged":   # the second instance of the launch: one step with a tenth of the streams sitting it out
    assert product.BeatriceBatch_EnableSilentBlockRule(batch.h, 1) == 0
    assert product.BeatriceBatch_SetSilentStreams(batch.h, bytes(1 if s % 10 == 3 else 0 for s in range(B))) == 0
for _ in range(60):
    product.BeatriceBatch_ConvertFramesDevice(batch.h, None, None)
us, fl, by = ctypes.c_float(0), ctypes.c_double(0), ctypes.c_double(0)
for ticks in (48, 48, 64, 64, 16):
    product.BeatriceBatch_TimeTickLaunch(batch.h, ticks, ctypes.byref(us), ctypes.byref(fl), ctypes.byref(by))
    print("TimeTickLaunch(%d): %.2f us" % (ticks, us.value))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(400):
    product.BeatriceBatch_ConvertFramesDevice(batch.h, None, None)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 400 * 1e6
print("loop without drain: %.2f us per tick of %d hop(s) x %d streams = %.3f M frames/s" % (dt, H, B, H * B / dt))
if os.environ.get("BEATRICE_HIP_TICK_TRACE"):   # the drain dumps the per-workgroup timeline of the last full tick
    product.BeatriceBatch_Synchronize(batch.h)
