"""examples/batch_convert: a C++ program on the batched C-ABI (model readers of the reference ABI, BeatriceBatch_*, host
streaming).  Its output file must equal what the Python-driven in-order chain gives for the same input, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_program_matches_python_driven_chain(bv, product, model_dir, tmp_path):
    exe = os.path.join(REPO, "examples", "batch_convert")
    assert os.path.exists(exe), "examples/batch_convert was not built (make -C beatrice-vst_amd)"
    B, hops, speaker, k = 40, 45, 2, 3
    audio = np.stack([bv.synth_audio(160 * hops, seed=6100 + s) for s in range(B)]).reshape(B, hops, 160)
    x = np.ascontiguousarray(audio.transpose(1, 0, 2))            # [hops][B][160]
    fin, fout = str(tmp_path / "in.f32"), str(tmp_path / "out.f32")
    x.tofile(fin)
    r = subprocess.run([exe, model_dir, str(B), str(hops), fin, fout, str(speaker), str(k)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.fromfile(fout, np.float32).reshape(hops, B, 240)
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, B)
    batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, -1, speaker)
    batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)
    batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, -1, k)
    want = np.stack([batch.convert(np.ascontiguousarray(x[h])) for h in range(hops)])
    batch.close()
    m.close()
    assert np.abs(want).max() > 0.05
    assert np.array_equal(got, want)
