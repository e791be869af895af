import importlib, os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch; torch.cuda.init()
import bench, make_model
bv = importlib.import_module("beatrice-vst_amd")
product = bv.bind_batch(bv.load_product())
tmp = tempfile.TemporaryDirectory(); make_model.make_model(tmp.name, n_speakers=1)
m = bv.Models(product, tmp.name)
print(os.environ.get("BEATRICE_HIP_HS_COPIES"), bench.host_buffer_rate(bv, m, product, 256)["streamed_through_tick_pipeline"])
