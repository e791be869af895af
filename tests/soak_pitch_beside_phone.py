"""Soak of the pitch hop beside the phone call: N plugin instances on N threads, every call against the oracle's for the same call sequence; every 97th hop of a thread
deviates (other samples for the pitch call), contexts are destroyed and re-created now and then while the other threads run.  python tests/soak_pitch_beside_phone.py [threads=8] [hops=1500]   (test infrastructure: loads the oracle as the checker; not collected by pytest)"""
import importlib, os, sys, tempfile, threading
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import make_model
bv = importlib.import_module("beatrice-vst_amd")
from test_gpu_pitch_beside_phone import Pair, stats
product = bv.bind_batch(bv.load_product())
oracle = bv.Abi(os.path.join(REPO, "oracle", "libbeatrice_oracle.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hops = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
tmp = tempfile.TemporaryDirectory(); make_model.make_model(tmp.name, n_speakers=3)
xs = [bv.synth_audio(160 * hops, seed=900 + t) for t in range(n)]
ys = [bv.synth_audio(160 * hops, seed=990 + t) for t in range(n)]
res = {}
for abi in (product, oracle):
    m = bv.Models(abi, tmp.name)
    outs = [[] for _ in range(n)]
    st = [None] * n
    def work(t):
        p = Pair(bv, abi, m, speaker=t % 3, k=2 * (t % 3))
        for i in range(hops):
            h = xs[t][i * 160:(i + 1) * 160]
            outs[t].append(p.phone(h))
            if i % 97 == 50 + t:
                outs[t].append(p.pitch(ys[t][i * 160:(i + 1) * 160]))
            else:
                outs[t].append(p.pitch(h))
            if i % 400 == 200 + 10 * t:
                p.new_pitch_context()
            if i % 700 == 350 + 10 * t:
                p.close(); p = Pair(bv, abi, m, speaker=t % 3, k=2 * (t % 3))
        if abi is product:
            st[t] = stats(product, p.tc)
        p.close()
    if abi is product:
        th = [threading.Thread(target=work, args=(t,)) for t in range(n)]
        [t.start() for t in th]; [t.join() for t in th]
        print("product done:", st, flush=True)
    else:
        for t in range(n): work(t)
    m.close()
    res[abi is product] = outs
bad = 0
for t in range(n):
    for i, (g, w) in enumerate(zip(res[True][t], res[False][t])):
        if not np.array_equal(g, w):
            bad += 1
            if bad < 5: print("thread %d call %d differs: max-abs %g" % (t, i, np.abs(g - w).max()))
print("calls compared: %d, differing: %d" % (sum(len(o) for o in res[True]), bad))
sys.exit(1 if bad else 0)
