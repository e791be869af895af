"""In-order chain at a large batch (bench.py's `saturation`), for A/B builds: BEATRICE_HIP_LIB=... python tools/debug/sat.py [streams]"""
import importlib, os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
torch.cuda.init()
import bench, make_model
bv = importlib.import_module("beatrice-vst_amd")
product = bv.bind_batch(bv.load_product())
tmp = tempfile.TemporaryDirectory()
make_model.make_model(tmp.name, n_speakers=1)
m = bv.Models(product, tmp.name)
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
r = bench.saturation(bv, m, product, streams=streams)
print(os.environ.get("BEATRICE_HIP_LIB", "default"), r)
batch = bv.Batch(m, streams)
rows = batch.profile_kernels(repeats=3)
for x in sorted(rows, key=lambda x: -x["mean_us"] * x["launches"])[:14]:
    print("  %-22s n %d  %7.1f us  %6.1f TF" % (x["name"], x["launches"], x["mean_us"], x["flops"] / x["mean_us"] / 1e6))
