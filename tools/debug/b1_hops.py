"""Drive the 1-stream C-ABI for a few hundred hops (profiling aid; see b1_prof.sh)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402  (initialised before the product library: both bundle a HIP runtime)

torch.cuda.init()
bv = importlib.import_module("beatrice-vst_amd")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from make_model import make_model  # noqa: E402

hops = int(sys.argv[1]) if len(sys.argv) > 1 else 600
model_dir = "/tmp/b1_model"
make_model(model_dir, n_speakers=1)
product = bv.load_product()
m = bv.Models(product, model_dir)
s = bv.Stream1(m, speaker=0)
x = bv.synth_audio(160 * 64, seed=5)
lat = []
for i in range(hops):
    t0 = time.perf_counter()
    s.hop(x[(i % 64) * 160:(i % 64 + 1) * 160])
    lat.append(time.perf_counter() - t0)
lat = np.array(lat[hops // 2:]) * 1e6
print("p50 %.1f us  p99 %.1f us  mean %.1f us" % (np.percentile(lat, 50), np.percentile(lat, 99), lat.mean()))
s.close()
m.close()
