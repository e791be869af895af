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


def test_latency_loop_program_equals_its_oracle_build(model_dir, built):
    """examples/latency_b1 (the C++ loop bench.py's `latency_b1` runs: the reference's three per-hop calls, clock_gettime) against the
    same source linked to the oracle: the running checksum of one output sample per hop over 600 hops must be EQUAL (bit-identical
    PCM), the report must parse, and no hop of a healthy run takes longer than the 10 ms budget."""
    import json
    exe, ora = os.path.join(REPO, "examples", "latency_b1"), os.path.join(REPO, "oracle", "latency_b1_on_oracle")
    assert os.path.exists(exe), "examples/latency_b1 was not built (make -C beatrice-vst_amd)"
    assert os.path.exists(ora), "oracle/latency_b1_on_oracle was not built (make -C oracle)"
    lines = []
    for prog in (exe, ora):
        r = subprocess.run([prog, model_dir, "600", "50", "2", "--histogram"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines.append(json.loads(r.stdout.strip().splitlines()[-1]))
    hip, cpu = lines
    print("latency loop, 600 hops: HIP p50 %.1f us p99 %.1f max %.1f | oracle p50 %.1f us" % (hip["p50_us"], hip["p99_us"], hip["max_us"], cpu["p50_us"]))
    assert hip["last_hop_peak"] > 1e-3
    assert hip["checksum"] == cpu["checksum"] and hip["last_hop_peak"] == cpu["last_hop_peak"]
    assert hip["hops_over_10ms"] == 0


@pytest.mark.parametrize("speaker,k,placement", [(2, 3, "range"), (-1, 2, "range"), (-1, 0, "speaker")])
def test_node_host_program_matches_single_batch(bv, product, model_dir, tmp_path, speaker, k, placement):
    """examples/node_convert: the C++ host of one node (one thread + one batch per GPU, model read once on GPU 0, parameter
    blobs and speaker tables by ncclBroadcast on librccl directly, frame counters all-reduced).  The test box has one GPU, so
    N = 1 here: the whole flow runs -- communicator, in-place broadcasts, device-scoped objects, the all-reduce -- and its
    output must equal the Python-driven in-order chain's bit for bit, with a fixed target speaker and with per-stream
    speakers (speaker = stream mod n_speakers: the speaker-affine placement's input)."""
    import json
    exe = os.path.join(REPO, "examples", "node_convert")
    assert os.path.exists(exe), "examples/node_convert was not built (make -C beatrice-vst_amd)"
    B, hops = 24, 40
    audio = np.stack([bv.synth_audio(160 * hops, seed=6400 + s) for s in range(B)]).reshape(B, hops, 160)
    x = np.ascontiguousarray(audio.transpose(1, 0, 2))            # [hops][B][160]
    fin, fout = str(tmp_path / "in.f32"), str(tmp_path / "out.f32")
    x.tofile(fin)
    r = subprocess.run([exe, model_dir, "1", str(B), str(hops), fin, fout, str(speaker), str(k), placement], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["frames"] == B * hops and line["gpus"] == 1 and line["streams_per_gpu"] == [B] and line["broadcast_bytes"] > 20e6
    got = np.fromfile(fout, np.float32).reshape(hops, B, 240)
    m = bv.Models(product, model_dir)
    n = m.tables.n_speakers
    batch = bv.Batch(m, B)
    for s in range(B):
        batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, speaker if speaker >= 0 else s % n)
    batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)
    batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, -1, k)
    want = np.stack([batch.convert(np.ascontiguousarray(x[h])) for h in range(hops)])
    batch.close()
    m.close()
    assert np.abs(want).max() > 0.05
    assert np.array_equal(got, want)
    # more GPUs than the box has: refused with a message, not a crash
    r = subprocess.run([exe, model_dir, "64", str(B), str(hops), fin, fout], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "devices" in r.stderr


def test_objects_carry_their_device(bv, product, model_dir):
    """BeatriceHip_SetDevice: objects are created on the calling thread's target device and every entry point runs there.
    With one GPU: ordinal 0 is accepted and everything works while ANOTHER thread-current HIP device cannot be selected
    (there is none); an ordinal the runtime does not have is refused and leaves the target alone."""
    a = bv.bind_batch(product)
    assert a.BeatriceHip_SetDevice(0) == 0 and a.BeatriceHip_GetDevice() == 0
    assert a.BeatriceHip_SetDevice(10 ** 6) == -1 and a.BeatriceHip_GetDevice() == 0
    m = bv.Models(product, model_dir)
    batch = bv.Batch(m, 3)
    assert a.BeatriceBatch_Device(batch.h) == 0
    x = np.stack([bv.synth_audio(160, seed=s) for s in range(3)])
    y = batch.convert(x)
    assert np.isfinite(y).all()
    batch.close()
    m.close()
    assert a.BeatriceHip_SetDevice(-1) == 0      # back to "the thread's current HIP device"
