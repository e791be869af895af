import sys, os, importlib.util
sys.path.insert(0, "tests")
import conftest
import test_gpu_resident_io as t
bv = conftest._load_pkg()
product = bv.load_product()
import tempfile
sys.path.insert(0, "tools"); import make_model
with tempfile.TemporaryDirectory() as d:
    make_model.make_model(d, n_speakers=3)
    for (B, H, slots, steps, depth) in [(48, 1, 16, 1500, 4), (64, 1, 7, 1200, 3), (16, 2, 5, 600, 4), (200, 1, 12, 400, 2)]:
        t.test_resident_io_matches_host_buffers(bv, product, d, B, H, slots, steps, depth)
print("soak ok")
