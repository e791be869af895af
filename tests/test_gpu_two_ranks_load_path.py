"""The multi-GPU load path on real device memory: two ranks (both on the one GPU of the test box, gloo carrying the
collectives -- RCCL refuses two ranks on one device) run beatrice-vst_amd/shard.py exactly as bench.py does: rank 0 reads
and packs the model files, rank 1 receives the PACKED DEVICE BLOBS in place (torch tensors aliasing the library's
allocations), marks them ready, receives the raw speaker tables the same way and projects them itself.  Both ranks then
convert their shard of the streams; the union must equal one process converting all of them, bit for bit."""
import importlib.util
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL, HOPS = 24, 6


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _convert(bv, product, m, handles_batch, lo, hi):
    batch = handles_batch
    for s in range(lo, hi):
        batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s - lo, s % 3)
        batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s - lo, s % 3)
    batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)
    audio = np.stack([bv.synth_audio(160 * HOPS, seed=7300 + s) for s in range(lo, hi)])
    return np.stack([batch.convert(np.ascontiguousarray(audio[:, h * 160:(h + 1) * 160])) for h in range(HOPS)])


def _worker(rank, world, port, model_dir, q):
    try:
        import torch
        import torch.distributed as dist
        torch.cuda.init()
        torch.cuda.set_device(0)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        bv = _load("beatrice_vst_amd_t", os.path.join(REPO, "beatrice-vst_amd", "__init__.py"))
        shard = _load("beatrice_shard_t", os.path.join(REPO, "beatrice-vst_amd", "shard.py"))
        product = bv.bind_batch(bv.load_product())

        class Loaded:
            pass
        m = Loaded()
        m.abi = product
        m.phone, m.pitch = product.CreatePhoneExtractor(), product.CreatePitchEstimator()
        m.wave, m.embed = product.CreateWaveformGenerator(), product.CreateEmbeddingSetter()
        moved, path = shard.load_models_from_rank0(product, {"phone": m.phone, "pitch": m.pitch, "wave": m.wave, "embed": m.embed},
                                                   model_dir if rank == 0 else "/nonexistent", rank, world, dist, torch)
        if rank == 0:
            m.tables = bv.SpeakerTables(product, model_dir)
        lo, hi = shard.stream_range(rank, world, TOTAL)
        batch = bv.Batch(m, hi - lo, max_speakers=4, upload_tables=(rank == 0))
        tmoved, tpath = shard.share_speaker_tables(product, batch.h, 4, rank, world, dist, torch, host_tables=m.tables if rank == 0 else None)
        if rank != 0:
            batch.apply_defaults()
        out = _convert(bv, product, m, batch, lo, hi)
        batch.close()
        q.put((rank, lo, hi, out.tobytes(), out.shape, moved, path, tmoved, tpath))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e)))


def test_two_ranks_share_packed_blobs_and_tables_in_place(bv, product, model_dir):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 26500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, model_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for r in results:
        assert r[1] != "error", r[2]
    m = bv.Models(product, model_dir)
    whole = _convert(bv, product, m, bv.Batch(m, TOTAL, max_speakers=4), 0, TOTAL)
    m.close()
    assert np.abs(whole).max() > 0.05
    covered = 0
    for rank, lo, hi, raw, shape, moved, path, tmoved, tpath in results:
        got = np.frombuffer(raw, np.float32).reshape(shape)
        assert np.array_equal(got, whole[:, lo:hi]), "rank %d's streams differ from the single-process batch" % rank
        assert path == "device blobs, in place" and tpath == "device tables, in place", (path, tpath)
        assert moved > 20e6 and tmoved > 1e6      # the packed parameters (22 MB) and the raw tables really travelled
        covered += hi - lo
    assert covered == TOTAL


def test_bench_with_two_ranks_on_this_box():
    """`python bench.py --gpus 2` end to end (rank spawn, load path, sharded timing, one JSON line from rank 0), with gloo
    carrying the collectives so that both ranks can share this box's GPU."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_PORT=str(28300 + os.getpid() % 1000))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--collectives", "gloo", "--streams", "64",
                          "--steps", "40", "--warmup", "5", "--no-extras", "--device-warm-ms", "0"],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["streams_per_gpu"] == 64
    assert line["value"] > 1e4 and line["output_rms"] > 0.01
    assert "device blobs, in place" in line["config"]["parallelism"]
