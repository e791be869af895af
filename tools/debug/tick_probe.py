"""Debug aid: one stream batch through the tick pipeline vs the in-order chain, step by step."""
import os, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tools"))
import importlib.util
spec = importlib.util.spec_from_file_location("beatrice_vst_amd", os.path.join(REPO, "beatrice-vst_amd", "__init__.py"))
bv = importlib.util.module_from_spec(spec); sys.modules["beatrice_vst_amd"] = bv; spec.loader.exec_module(bv)
import make_model
from test_gpu_resident_io import Hip
product = bv.bind_batch(bv.load_product())
d = tempfile.mkdtemp(); make_model.make_model(d, n_speakers=3)
m = bv.Models(product, d)
B, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 6
audio = np.stack([bv.synth_audio(160 * steps, seed=8000 + s) for s in range(B)])
hip = Hip()
MODE = os.environ.get("MODE", "")
def settings(batch):
    for s in range(B):
        if "spk" in MODE: batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % 3)
        if "vq" in MODE: batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, s % 3)
    batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)
def change(batch, k):
    if "chg" in MODE and k % 4 == 1:
        s = (7 * k) % B
        batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, (k + s) % 3)
    if "fmt" in MODE and k % 4 == 1:
        batch.a.BeatriceBatch_SetFormantShift(batch.h, (7 * k + 1) % B, float(k % 5) - 2.0)
    if "vqc" in MODE and k % 9 == 5:
        batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, (3 * k) % B, k % 5)
    if "rst" in MODE and k == 33:
        assert batch.a.BeatriceBatch_ResetStream(batch.h, 2 % B) == 0
    if "mnp" in MODE and k == 41:
        batch.a.BeatriceBatch_SetMinSourcePitch(batch.h, 0, 50.0)
    if "cor" in MODE and k == 41:
        batch.a.BeatriceBatch_SetPitchCorrection(batch.h, 1 % B, 0.6)
    if "pit" in MODE and k % 4 == 1:
        batch.a.BeatriceBatch_SetPitchShift(batch.h, (7 * k + 2) % B, float(k % 7) - 3.0)
ref_b = bv.Batch(m, B)
settings(ref_b)
ref, refi = [], []
for k in range(steps):
    change(ref_b, k)
    ref.append(ref_b.convert(audio[:, k * 160:(k + 1) * 160])); refi.append(ref_b.intermediates())
ref_b.close()
b = bv.Batch(m, B); a, h = b.a, b.h
settings(b)
slots = a.BeatriceBatch_TickStages(h) + 6
d_in, d_out = hip.malloc(slots * B * 160 * 4), hip.malloc(slots * B * 240 * 4)
assert a.BeatriceBatch_BindResidentIO(h, d_in, d_out, slots) == 0
assert a.BeatriceBatch_EnableTickPipeline(h, 1) == 0
buf = np.zeros((slots, B, 160), np.float32)
for k in range(steps): buf[k % slots] = audio[:, k * 160:(k + 1) * 160]
hip.h2d(d_in, buf)
for k in range(steps):
    change(b, k)
    assert a.BeatriceBatch_ConvertFramesDevice(h, None, None) == 0
    if os.environ.get("SYNC_EACH"):
        assert a.BeatriceBatch_Synchronize(h) == 0
        ph, qr, q, ft = b.intermediates()
        print("step", k, "phone", np.abs(ph - refi[k][0]).max(), "q_raw eq", np.array_equal(qr, refi[k][1]), "feat", np.abs(ft - refi[k][3]).max())
assert a.BeatriceBatch_Synchronize(h) == 0
out = np.zeros((slots, B, 240), np.float32); hip.d2h(out, d_out)
for k in range(steps):
    dd = np.abs(out[k % slots] - ref[k]).max(axis=1)
    if dd.max() > 0: print("step", k, "streams that differ", np.nonzero(dd)[0].tolist(), "max-abs", dd.max())
print("done", MODE)
b.close()
