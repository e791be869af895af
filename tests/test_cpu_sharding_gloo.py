"""The N > 1 path on CPU: two processes over gloo run the same broadcast / shard / reduce code that
bench.py runs over RCCL, with the oracle standing in for the device library (no GPU here).
Checks (1) every rank receives rank 0's model bytes, (2) sharded streams == the same streams run in one
process (multi-GPU invariance), (3) the MAX-over-ranks reduction."""
import hashlib
import importlib.util
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL_STREAMS, HOPS = 5, 6


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _run_streams(bv, model_dir, lo, hi):
    oracle = bv.Abi(os.path.join(REPO, "oracle", "libbeatrice_oracle.so"))
    m = bv.Models(oracle, model_dir)
    out = {}
    for s in range(lo, hi):
        st = bv.Stream1(m, speaker=s % 2, vq_k=s % 3)
        x = bv.synth_audio(160 * HOPS, seed=300 + s)
        out[s] = np.stack([st.hop(x[i * 160:(i + 1) * 160]) for i in range(HOPS)])
        st.close()
    m.close()
    return out


def _worker(rank, world, port, src_dir, tmp_root, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bv = _load("beatrice_vst_amd", os.path.join(REPO, "beatrice-vst_amd", "__init__.py"))
    shard = _load("beatrice_shard", os.path.join(REPO, "beatrice-vst_amd", "shard.py"))
    blobs = shard.broadcast_model(src_dir if rank == 0 else "/nonexistent", rank, world, dist, torch, "cpu")
    my_dir = os.path.join(tmp_root, "rank%d" % rank)
    os.makedirs(my_dir, exist_ok=True)
    for f, data in blobs.items():
        with open(os.path.join(my_dir, f), "wb") as fh:
            fh.write(data)
    lo, hi = shard.stream_range(rank, world, TOTAL_STREAMS)
    outs = _run_streams(bv, my_dir, lo, hi)
    # the in-place broadcast bench.py uses for the packed device blobs (here: host memory standing in for the blob)
    blob = np.arange(4096, dtype=np.uint8) if rank == 0 else np.zeros(4096, np.uint8)
    shard.broadcast_inplace(torch.from_numpy(blob), world, dist)
    assert np.array_equal(blob, np.arange(4096, dtype=np.uint8)), "rank %d: in-place broadcast did not land" % rank
    slowest = shard.max_over_ranks(float(rank + 1), world, dist, torch, "cpu")
    digest = {f: hashlib.sha256(d).hexdigest() for f, d in blobs.items()}
    q.put((rank, lo, hi, {s: o.tobytes() for s, o in outs.items()}, digest, slowest))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(bv, built, model_dir, tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, model_dir, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _run_streams(bv, model_dir, 0, TOTAL_STREAMS)
    want = {f: hashlib.sha256(open(os.path.join(model_dir, f), "rb").read()).hexdigest()
            for f in ("phone_extractor.bin", "speaker_embeddings.bin")}
    covered = set()
    for rank, lo, hi, outs, digest, slowest in results:
        assert slowest == 2.0
        for f, h in want.items():
            assert digest[f] == h, "rank %d received different bytes for %s" % (rank, f)
        for s, raw in outs.items():
            got = np.frombuffer(raw, np.float32).reshape(HOPS, 240)
            assert np.array_equal(got, ref[s]), "stream %d differs when sharded" % s
            covered.add(s)
    assert covered == set(range(TOTAL_STREAMS))


def _agree_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = _load("beatrice_shard", os.path.join(REPO, "beatrice-vst_amd", "shard.py"))
    # only rank 1 fails: every rank must learn it, so that all of them take the fallback branch together
    q.put((rank, shard.all_ranks_ok(rank != 1, world, dist, torch, "cpu"), shard.all_ranks_ok(True, world, dist, torch, "cpu")))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_the_load_path():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 27500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, False, True), (1, False, True)]


def test_stream_range_partitions():
    shard = _load("beatrice_shard", os.path.join(REPO, "beatrice-vst_amd", "shard.py"))
    for world in (1, 2, 3, 8):
        for total in (0, 1, 7, 256, 2048, 2049):
            spans = [shard.stream_range(r, world, total) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (checked over gloo: no GPU here);
    a launcher whose world size disagrees with --gpus is refused."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_PORT=str(29300 + os.getpid() % 1000))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                         capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                         capture_output=True, text=True, timeout=600, cwd=REPO, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_speaker_affine_placement():
    shard = _load("beatrice_shard", os.path.join(REPO, "beatrice-vst_amd", "shard.py"))
    for world in (1, 2, 4, 8):
        seen = set()
        for rank in range(world):
            mine = {shard.affine_speaker(rank, world, s, 64) for s in range(256)}
            assert all(spk % world == rank for spk in mine)     # gpu = speaker mod world
            assert len(mine) == 64 // world
            seen |= mine
        assert seen == set(range(64))
    assert {shard.affine_speaker(3, 8, s, 2) for s in range(16)} == {0, 1}   # fewer speakers than ranks
    # a table size the world does not divide: the ranks' sets still partition it, every speaker congruent to its rank
    for world, n in ((3, 64), (5, 64), (7, 10), (8, 9)):
        sets = [{shard.affine_speaker(rank, world, s, n) for s in range(4 * n)} for rank in range(world)]
        assert sorted(x for st in sets for x in st) == list(range(n)), (world, n)
        assert all(x % world == rank for rank, st in enumerate(sets) for x in st)


def test_world_8_partitions_at_the_baseline_config_sizes():
    """BASELINE.json configs[3] (2048 streams, 64 rotating speakers) and configs[4] (512 stereo streams) over the 8 GPUs of a
    node: the shards the 8-GPU run will use, checked here so that run needs no code -- 256 / 64 streams per rank, the shards
    tile the job, and under speaker-affine placement every rank holds 8 speakers x 32 streams, i.e. every key/value slot of a
    rank fills whole 16-row attention tiles (no padded tile, no quad: the tick launch's fast path)."""
    shard = _load("beatrice_shard", os.path.join(REPO, "beatrice-vst_amd", "shard.py"))
    world = 8
    for total, per_rank in ((2048, 256), (512, 64)):
        spans = [shard.stream_range(r, world, total) for r in range(world)]
        assert [h - l for l, h in spans] == [per_rank] * world and spans[0][0] == 0 and spans[-1][1] == total
    all_speakers = []
    for rank in range(world):
        spk = [shard.affine_speaker(rank, world, s, 64) for s in range(256)]
        counts = {v: spk.count(v) for v in set(spk)}
        assert len(counts) == 8 and set(counts.values()) == {32} and all(v % world == rank for v in counts)
        assert all(c % 16 == 0 for c in counts.values())
        all_speakers += list(counts)
        # a stream that rotates to the next speaker stays on its rank's set: the rotation step is `world`
        nxt = [shard.affine_speaker(rank, world, s + 1, 64) for s in range(256)]
        assert set(nxt) == set(spk)
    assert sorted(all_speakers) == list(range(64))
