"""What the GPU did right before a 20-step run decides how fast its partly filled launches go (clocks): the same 20 steps + drain, timed by the host,
after different preludes.  python tools/debug/clock_state_probe.py [hops per step = 4]"""
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
B, n, steps = 256, 64, 20
H = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = bv.Batch(m, B, hops_per_step=H)
d_in = torch.randn((n, B, H * 160), device="cuda") * 0.1
d_out = torch.zeros((n, B, H * 240), device="cuda")
assert product.BeatriceBatch_BindResidentIO(batch.h, d_in.data_ptr(), d_out.data_ptr(), n) == 0
assert product.BeatriceBatch_EnableTickPipeline(batch.h, 1) == 0
x = torch.randn((4096, 4096), device="cuda")
w = torch.randn((2048, 2048), device="cuda")

def feed(k):
    for _ in range(k): product.BeatriceBatch_ConvertFramesDevice(batch.h, None, None)
    product.BeatriceBatch_Synchronize(batch.h)

def big(k=30):
    global x
    for _ in range(k): y = x @ x
    torch.cuda.synchronize()

def benchwarm(ms=300.0):
    global w
    t_end = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(8): w = (w @ w).clamp_(-1.0, 1.0)
        torch.cuda.synchronize()

preludes = {
    "big matmuls": lambda: big(),
    "big matmuls, 5 steps + drain": lambda: (big(), feed(5)),
    "bench warm (2048 matmul + clamp, sync every 8)": lambda: benchwarm(),
    "bench warm, 5 steps + drain": lambda: (benchwarm(), feed(5)),
    "5 steps + drain, big matmuls": lambda: (feed(5), big()),
    "5 steps + drain, bench warm": lambda: (feed(5), benchwarm()),
    "300 full steps + drain": lambda: feed(300),
    "300 full steps + drain, 5 steps + drain": lambda: (feed(300), feed(5)),
    "idle 50 ms": lambda: time.sleep(0.05),
    "idle 50 ms, 5 steps + drain": lambda: (time.sleep(0.05), feed(5)),
}
feed(40)
for name, pre in preludes.items():
    ts = []
    for rep in range(5):
        pre(); torch.cuda.synchronize()
        t0 = time.perf_counter(); feed(steps); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("%-50s 20 steps + drain: median %.3f ms (%.3f .. %.3f) = %.2f M frames/s" % (name, ts[2], ts[0], ts[-1], B * H * steps / ts[2] / 1e3), flush=True)
