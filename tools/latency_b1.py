import importlib.util, os, sys, json, tempfile
sys.argv=["bench.py"]
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
bv = b.load_pkg(); product = bv.bind_batch(bv.load_product())
sys.path.insert(0,"tools"); import make_model
with tempfile.TemporaryDirectory() as d:
    make_model.make_model(d, n_speakers=1)
    print(b.latency_b1(bv, product, d))
