"""A 20-step run (fill + drain) of the bench's workload after the product's own steady work (the clocks a running server has): median of 9.
python tools/debug/short_run_rate.py [hops per step = 4] [steps = 20]"""
import importlib, os, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
torch.cuda.init()
import make_model
bv = importlib.import_module("beatrice-vst_amd")
product = bv.bind_batch(bv.load_product())
tmp = tempfile.TemporaryDirectory(); make_model.make_model(tmp.name, n_speakers=1)
m = bv.Models(product, tmp.name)
B, n = 256, 64
H = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
batch = bv.Batch(m, B, hops_per_step=H)
d_in = torch.randn((n, B, H * 160), device="cuda") * 0.1
d_out = torch.zeros((n, B, H * 240), device="cuda")
assert product.BeatriceBatch_BindResidentIO(batch.h, d_in.data_ptr(), d_out.data_ptr(), n) == 0
assert product.BeatriceBatch_EnableTickPipeline(batch.h, 1) == 0
def feed(k):
    for _ in range(k): product.BeatriceBatch_ConvertFramesDevice(batch.h, None, None)
    product.BeatriceBatch_Synchronize(batch.h)
ts = []
for rep in range(9):
    feed(300); feed(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); feed(steps); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print("%s: %d steps + drain at %d hops per step: median %.3f ms (%.3f .. %.3f) = %.3f M frames/s" % (os.environ.get("TAG", ""), steps, H, ts[4], ts[0], ts[-1], B * H * steps / ts[4] / 1e3), flush=True)
