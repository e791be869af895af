#!/usr/bin/env python3
"""Do independent per-hop chains overlap on the GPU?  N BeatriceBatch objects (own stream + own
hipGraph each) of B/N streams are stepped round-robin from one host thread."""
import importlib.util, os, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("beatrice_vst_amd", os.path.join(REPO, "beatrice-vst_amd", "__init__.py"))
bv = importlib.util.module_from_spec(spec); sys.modules["beatrice_vst_amd"] = bv; spec.loader.exec_module(bv)
sys.path.insert(0, os.path.join(REPO, "tools"))
import make_model
product = bv.bind_batch(bv.load_product())
total = int(sys.argv[1]) if len(sys.argv) > 1 else 256
with tempfile.TemporaryDirectory() as d:
    make_model.make_model(d, n_speakers=1)
    m = bv.Models(product, d)
    for n in (1, 2, 4, 8):
        batches = [bv.Batch(m, total // n) for _ in range(n)]
        for _ in range(20):
            for b in batches:
                product.BeatriceBatch_ConvertFramesDevice(b.h, None, None)
        for b in batches:
            product.BeatriceBatch_Synchronize(b.h)
        steps = 200
        t0 = time.perf_counter()
        for _ in range(steps):
            for b in batches:
                product.BeatriceBatch_ConvertFramesDevice(b.h, None, None)
        for b in batches:
            product.BeatriceBatch_Synchronize(b.h)
        dt = time.perf_counter() - t0
        print("%d chains x %3d streams: %.1f us per step of all %d streams -> %.0f frames/s" %
              (n, total // n, dt / steps * 1e6, total, total * steps / dt))
        for b in batches:
            b.close()
    m.close()
