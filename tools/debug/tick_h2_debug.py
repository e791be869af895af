"""tick mode at 2 hops per step against the in-order chain at 2 hops per step: outputs and front-end intermediates step by step"""
import importlib, os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import make_model
from test_gpu_resident_io import Hip
bv = importlib.import_module("beatrice-vst_amd")
product = bv.bind_batch(bv.load_product())
tmp = tempfile.TemporaryDirectory(); make_model.make_model(tmp.name, n_speakers=3)
m = bv.Models(product, tmp.name)
B, H = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
audio = np.stack([bv.synth_audio(160 * H * N, seed=8100 + s) for s in range(B)])
hip = Hip()
for n in range(1, N + 1):
    ref = bv.Batch(m, B, hops_per_step=H)
    for k in range(n):
        want = ref.convert(audio[:, k * H * 160:(k + 1) * H * 160])
    wi = [x.copy() for x in ref.intermediates()]
    ref.close()
    batch = bv.Batch(m, B, hops_per_step=H)
    a, h = batch.a, batch.h
    slots = a.BeatriceBatch_TickStages(h) + 5
    d_in, d_out = hip.malloc(slots * B * H * 160 * 4), hip.malloc(slots * B * H * 240 * 4)
    assert a.BeatriceBatch_BindResidentIO(h, d_in, d_out, slots) == 0
    assert a.BeatriceBatch_EnableTickPipeline(h, 1) == 0
    buf = np.zeros((slots, B, H * 160), np.float32)
    for k in range(n):
        buf[k] = audio[:, k * H * 160:(k + 1) * H * 160]
    hip.h2d(d_in, buf)
    for k in range(n):
        assert a.BeatriceBatch_ConvertFramesDevice(h, None, None) == 0
    assert a.BeatriceBatch_Synchronize(h) == 0
    out = np.zeros((slots, B, H * 240), np.float32)
    hip.d2h(out, d_out)
    gi = batch.intermediates()
    got = out[n - 1]
    names = ["phone", "q_raw", "q", "feat"]
    msg = []
    for nm, x, y in zip(names, wi, gi):
        msg.append("%s %s" % (nm, "ok" if np.array_equal(x, y) else "DIFF rows %s" % sorted(set(np.argwhere(x != y)[:, :2].reshape(-1, 2)[:, 1].tolist()))[:4]))
    d = np.argwhere(want != got)
    msg.append("out " + ("ok" if d.size == 0 else "DIFF streams %s hop-halves %s first sample %d" % (sorted(set(d[:, 0].tolist()))[:6], sorted(set((d[:, 1] // 240).tolist())), d[:, 1].min())))
    print("after %d steps: %s" % (n, "; ".join(msg)))
    batch.close(); hip.free(d_in); hip.free(d_out)
