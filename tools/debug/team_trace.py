"""Per-stage timeline of the waveform team launch (workgroup 0): BEATRICE_HIP_TEAM_TRACE=1 python tools/debug/team_trace.py"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
bv = importlib.import_module("beatrice-vst_amd")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from make_model import make_model
model_dir = "/tmp/b1_model"; make_model(model_dir, n_speakers=1)
product = bv.load_product()
m = bv.Models(product, model_dir); s = bv.Stream1(m, speaker=0)
x = bv.synth_audio(160 * 64, seed=5)
for i in range(300): s.hop(x[(i % 64) * 160:(i % 64 + 1) * 160])
buf = (ctypes.c_ulonglong * 1024)()
product.lib.BeatriceHip_TeamTraceDump.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = product.lib.BeatriceHip_TeamTraceDump(buf, 1024)
t = np.array(buf[:max(n, 2)], dtype=np.int64)
print("stamps", n, "total cycles", t[-1] - t[0])
d = np.diff(t)
names = ["inputs", "weights->LDS", "chains", "epilogue"]
# stamps per tile: entered, (inputs), weights, chains, published -> first tile has 5, later tiles 4
print("per stamp deltas (cycles):", d.tolist()[:200])
s.close(); m.close()
