#!/usr/bin/env python3
"""frames/s over batch size x hops per step x pipeline depth (BeatriceBatch_TimeSteps on resident buffers).
Run with GPU_MAX_HW_QUEUES=8 for depth 4."""
import importlib.util, os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("beatrice_vst_amd", os.path.join(REPO, "beatrice-vst_amd", "__init__.py"))
bv = importlib.util.module_from_spec(spec); sys.modules["beatrice_vst_amd"] = bv; spec.loader.exec_module(bv)
sys.path.insert(0, os.path.join(REPO, "tools"))
import make_model
product = bv.bind_batch(bv.load_product())
with tempfile.TemporaryDirectory() as d:
    make_model.make_model(d, n_speakers=1)
    m = bv.Models(product, d)
    for B in (256, 512, 1024):
        for H in (1, 2, 4, 8):
            row = []
            for depth in (0, 2, 3, 4):
                batch = bv.Batch(m, B, hops_per_step=H)
                product.BeatriceBatch_EnablePipelining(batch.h, depth)
                batch.time_steps(20)
                ms = batch.time_steps(100)
                row.append("%8.0f" % (B * H * 100 / (ms * 1e-3)))
                batch.close()
            print("B=%4d H=%d  frames/s at depth 0/2/3/4: %s" % (B, H, " ".join(row)), flush=True)
    m.close()
