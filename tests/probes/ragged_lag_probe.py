"""which lag (steps a stream sat out before the batch leaves tick mode) breaks the hand-over to the in-order chain?  tools/debug/ragged_lag_probe.py [H]"""
import importlib, os, sys, tempfile
REPO = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, make_model
from tick_driver import run_tick
bv = importlib.import_module("beatrice-vst_amd")
product = bv.bind_batch(bv.load_product()); a = product
oracle = bv.Abi(os.path.join(REPO, "oracle", "libbeatrice_oracle.so"))
tmp = tempfile.TemporaryDirectory(); make_model.make_model(tmp.name, n_speakers=3)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B, steps, tail = 3, 30, 3
x = np.stack([bv.synth_audio(160 * H * (steps + tail), seed=7300 + s) for s in range(B)]).reshape(B, steps + tail, H * 160)
mo = bv.Models(oracle, tmp.name)
for L in list(range(1, 21)):
    out = set(range(steps - L, steps))
    so = bv.Stream1(mo, speaker=0)
    for k in range(steps):
        if k not in out:
            for hh in range(H): so.hop(x[1, k, hh * 160:(hh + 1) * 160])
    want = np.concatenate([so.hop(x[1, steps, hh * 160:(hh + 1) * 160]) for hh in range(H)])
    so.close()
    m = bv.Models(product, tmp.name); batch = bv.Batch(m, B, hops_per_step=H); h = batch.h
    on = []
    def change(b_, k):
        if not on:
            assert a.BeatriceBatch_EnableSilentBlockRule(h, 1) == 0; on.append(1)
        if k in out: assert a.BeatriceBatch_SetSilentStreams(h, bytes([0, 1, 0])) == 0
    run_tick(bv, batch, steps, lambda k: x[:, k], change=change, chunk=steps)
    assert a.BeatriceBatch_EnableSilentBlockRule(h, 0) == 0
    got = batch.convert(np.ascontiguousarray(x[:, steps]))[1]
    batch.close(); m.close()
    print("H %d lag %2d: %s (max-abs %.3g)" % (H, L, "ok" if np.array_equal(got, want) else "DIFFERS", float(np.abs(got - want).max())))
