"""Feeds N steps into the tick pipeline and drains; meant to run under rocprofv3 --kernel-trace (tools/debug/fill_drain.sh)."""
import importlib, os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
torch.cuda.init()
import make_model
bv = importlib.import_module("beatrice-vst_amd")
product = bv.bind_batch(bv.load_product())
tmp = tempfile.TemporaryDirectory(); make_model.make_model(tmp.name, n_speakers=1)
m = bv.Models(product, tmp.name)
B, n, steps = 256, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 20
H = int(os.environ.get("FD_HOPS", "2"))   # hops per step
batch = bv.Batch(m, B, hops_per_step=H)
d_in = torch.randn((n, B, H * 160), device="cuda") * 0.1
d_out = torch.zeros((n, B, H * 240), device="cuda")
assert product.BeatriceBatch_BindResidentIO(batch.h, d_in.data_ptr(), d_out.data_ptr(), n) == 0
assert product.BeatriceBatch_EnableTickPipeline(batch.h, 1) == 0
x = torch.randn((4096, 4096), device="cuda")
for rep in range(3):
    for _ in range(30): y = x @ x          # keep the clocks up between the runs
    torch.cuda.synchronize()
    import time
    ts = [time.perf_counter()]
    for _ in range(steps):
        product.BeatriceBatch_ConvertFramesDevice(batch.h, None, None); ts.append(time.perf_counter())
    product.BeatriceBatch_Synchronize(batch.h); ts.append(time.perf_counter())
    if os.environ.get("FD_HOSTTIMES") and rep == 2:
        print("host us per feed:", " ".join("%.0f" % ((b - a) * 1e6) for a, b in zip(ts[:-2], ts[1:-1])), "| drain %.0f | whole %.0f" % ((ts[-1] - ts[-2]) * 1e6, (ts[-1] - ts[0]) * 1e6), flush=True)
