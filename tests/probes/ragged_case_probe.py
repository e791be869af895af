"""the failing shape of tests/test_gpu_tick_ragged.py at several hops per step, with knobs: STEPS, KNN (0/1), SWITCH (0/1), CHUNK"""
import importlib, os, sys, tempfile
REPO = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, make_model
from tick_driver import run_tick
bv = importlib.import_module("beatrice-vst_amd")
product = bv.bind_batch(bv.load_product()); a = product
oracle = bv.Abi(os.path.join(REPO, "oracle", "libbeatrice_oracle.so"))
tmp = tempfile.TemporaryDirectory(); make_model.make_model(tmp.name, n_speakers=3)
H, B, tail = 4, 3, 3
steps = int(os.environ.get("STEPS", "36")); knn = int(os.environ.get("KNN", "1")); sw = int(os.environ.get("SWITCH", "1")); chunk = int(os.environ.get("CHUNK", "13"))
x = np.stack([bv.synth_audio(160 * H * (steps + tail), seed=7300 + s) for s in range(B)]).reshape(B, steps + tail, H * 160)
out = {0: set(), 1: {5, 6, 7, 30}, 2: {0, 1, 2} | set(range(20, 36))}
switch = {2: (19, 0)} if sw else {}
mo = bv.Models(oracle, tmp.name)
want = {}
for s in range(B):
    so = bv.Stream1(mo, speaker=s % 3, vq_k=(s % 3) if knn else 0)
    for k in range(steps):
        if s in switch and switch[s][0] == k: so.set_target_speaker(switch[s][1])
        if k not in out[s]:
            for hh in range(H): so.hop(x[s, k, hh * 160:(hh + 1) * 160])
    want[s] = [np.concatenate([so.hop(x[s, steps + t, hh * 160:(hh + 1) * 160]) for hh in range(H)]) for t in range(tail)]
    so.close()
m = bv.Models(product, tmp.name); batch = bv.Batch(m, B, hops_per_step=H); h = batch.h
for s in range(B):
    a.BeatriceBatch_SetTargetSpeaker(h, s, s % 3); a.BeatriceBatch_SetVQNumNeighbors(h, s, (s % 3) if knn else 0)
a.BeatriceBatch_FlushSpeaker(h, -1)
on = []
def change(b_, k):
    if not on:
        assert a.BeatriceBatch_EnableSilentBlockRule(h, 1) == 0; on.append(1)
    for s in switch:
        if switch[s][0] == k: a.BeatriceBatch_SetTargetSpeaker(h, s, switch[s][1])
    fl = bytes(1 if k in out[s] else 0 for s in range(B))
    if any(fl): assert a.BeatriceBatch_SetSilentStreams(h, fl) == 0
run_tick(bv, batch, steps, lambda k: x[:, k], change=change, chunk=chunk)
assert a.BeatriceBatch_EnableSilentBlockRule(h, 0) == 0
for t in range(tail):
    got = batch.convert(np.ascontiguousarray(x[:, steps + t]))
    print("steps %d knn %d switch %d chunk %d tail step %d:" % (steps, knn, sw, chunk, t), ["ok" if np.array_equal(got[s], want[s][t]) else "DIFF %.3g" % float(np.abs(got[s] - want[s][t]).max()) for s in range(B)])
batch.close(); m.close()
