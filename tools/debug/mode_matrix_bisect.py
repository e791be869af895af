"""which refused entry point perturbs a mode?  tools/debug/mode_matrix_bisect.py <mode> <H>  (runs tests/test_gpu_mode_matrix.py's walker with ONE probe at a time)"""
import importlib, os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import make_model, wrapperlib
bv = importlib.import_module("beatrice-vst_amd")
import test_gpu_mode_matrix as mm
product = bv.bind_batch(bv.load_product())
tmp = tempfile.TemporaryDirectory(); make_model.make_model(tmp.name, n_speakers=3)
mode, H = sys.argv[1], int(sys.argv[2])
B, CH = mm.B, mm.CH
steps = 7 if not mode.startswith("resident_blocks") else (10 if H == 1 else 44)
x16 = np.stack([bv.synth_audio(160 * H * steps, seed=8800 + s) for s in range(B)]).reshape(B, steps, H * 160)
x48 = np.stack([wrapperlib.test_signal(480 * H * steps * CH, 48000, seed=8900 + s) for s in range(B)]).astype(np.float32).reshape(B, steps, H, CH, 480)
def run(names):
    c = mm.Ctx(bv, product, tmp.name, H)
    try:
        mm.enter(c, mode)
        got = []
        for k in range(steps):
            if k in (1, 4):
                for n in names:
                    rc = mm.PROBES[n](c)
                    if rc != -1: print("  %s returned %d" % (n, rc))
            y = mm.step(c, mode, k, x16[:, k], x48[:, k])
            if y is not None: got.append(np.array(y, copy=True))
        got += mm.finish(c, mode)
        return got
    finally:
        c.close()
control = run([])
again = run([])
print("control vs control:", all(np.array_equal(p, q) for p, q in zip(control, again)))
for n in mm.refused(mode, H):
    got = run([n])
    same = len(got) == len(control) and all(np.array_equal(p, q) for p, q in zip(got, control))
    print("%-34s %s" % (n, "same" if same else "DIFFERS"))
