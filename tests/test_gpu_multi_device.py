"""More than one GPU in the box (the driver's 8-GPU node, never the 1-GPU gpurun box): the two multi-GPU hosts run on real RCCL
with N = 2 -- examples/node_convert (one process, one thread per GPU, ncclBroadcast of blobs and tables on librccl) against a
single batch bit for bit, and `bench.py --gpus 2` (one process per GPU, torch.distributed backend nccl) to a well-formed line.
Skipped where hipGetDeviceCount() < 2; the same code paths are covered with N = 1 (test_gpu_cpp_example.py) and over gloo
(test_cpu_sharding_gloo.py, test_gpu_two_ranks_load_path.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_devices():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_n_devices() < 2, reason="one GPU in this box")


@needs_two
@pytest.mark.parametrize("placement", ["range", "speaker"])
def test_node_host_on_two_gpus_matches_single_batch(bv, product, model_dir, tmp_path, placement):
    exe = os.path.join(REPO, "examples", "node_convert")
    assert os.path.exists(exe)
    B, hops, k = 48, 36, 2
    audio = np.stack([bv.synth_audio(160 * hops, seed=6700 + s) for s in range(B)]).reshape(B, hops, 160)
    x = np.ascontiguousarray(audio.transpose(1, 0, 2))
    fin, fout = str(tmp_path / "in.f32"), str(tmp_path / "out.f32")
    x.tofile(fin)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, model_dir, "2", str(B), str(hops), fin, fout, "-1", str(k), placement], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["frames"] == B * hops and line["gpus"] == 2 and sum(line["streams_per_gpu"]) == B
    got = np.fromfile(fout, np.float32).reshape(hops, B, 240)
    m = bv.Models(product, model_dir)
    n = m.tables.n_speakers
    batch = bv.Batch(m, B)
    for s in range(B):
        batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % n)
    batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)
    batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, -1, k)
    want = np.stack([batch.convert(np.ascontiguousarray(x[h])) for h in range(hops)])
    batch.close()
    m.close()
    assert np.array_equal(got, want)


@needs_two
def test_bench_on_two_gpus_over_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5", "--no-extras"],
                       capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["streams_per_gpu"] == 256
