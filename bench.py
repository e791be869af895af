#!/usr/bin/env python3
"""bench.py -- headline measurement for the Beatrice 2 per-hop voice-conversion path on MI355X.

One "step" = one 10 ms hop (160 samples @16 kHz in -> 240 samples @24 kHz out) for EVERY stream of
the batch: ExtractPhone + EstimatePitch + pitch transform + GenerateWaveform
(reference src/common/processor_core_2.cc:181-255), through the batched C-ABI
(include/beatrice_batch.h).  Workload = BASELINE.json configs[2] ("batch 256 concurrent streams,
1 speaker, 1xMI355X") per GPU; with --gpus N each rank runs its own 256 streams (weak scaling, no
collective inside a hop; weights are broadcast from rank 0 over RCCL at load).  Inputs are resident
in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      -- dominant kernel of the chain: algorithmic FLOP/s (or B/s) vs gfx950 peak, from
                   HIP-event timing on the library's own stream (BeatriceBatch_ProfileKernels)
  cpu_baseline  -- the CPU oracle (oracle/, a port of the frozen spec -- NOT the proprietary
                   beatricelib, which has no Linux build) timed on this box's host cores.
"""
import os as _os
# A pipeline depth of 4 keeps four HIP streams busy; ROCm maps streams onto 4 hardware queues by default, and a
# stream that shares a queue with another one waits for it.  Must be set before the HIP runtime initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import argparse
import ctypes
import importlib.util
import json
import os
import sys
import tempfile
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, dense f32 MFMA
PEAK_HBM_GBS = 8000.0
# The timed workloads and the -m gpu parity tests that run the same shape against the oracle.  tests/test_cpu_bench_defaults_are_tested.py
# fails when a default below has no parity test parametrised at it, so that the bench cannot outrun its tests.
DEFAULT_HOPS_PER_STEP_TICK = 4   # --hops-per-step with the tick pipeline (round 4: 2; rounds 1-3: 1)
DEFAULT_CONFIG = 2
DEFAULT_STREAMS = {2: 256, 3: 256, 4: 64}   # streams per GPU of --config N
PARITY_TESTS = {   # --config -> (test file, test function, name of its hops-per-step parameter, name of its stream-count parameter or None)
    2: [("test_gpu_throughput_vs_oracle.py", "test_bench_workload_in_tick_mode_matches_oracle", "H", None),
        ("test_gpu_throughput_vs_oracle.py", "test_tick_soak_vs_oracle", "H", None)],
    3: [("test_gpu_config_shapes.py", "test_config3_shape_through_the_tick_pipeline", "H", None)],
    4: [("test_gpu_wrapper48k.py", "test_48k_wrapper_around_the_tick_pipeline_two_blocks_per_step", "H", "B")],
}


def load_pkg():
    name = "beatrice_vst_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "beatrice-vst_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_shard():
    spec = importlib.util.spec_from_file_location("beatrice_shard", os.path.join(REPO, "beatrice-vst_amd", "shard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# launch name (BeatriceBatch_ProfileKernels) -> substring of the kernel symbol in the rocprofv3 PMC summary
PMC_SYMBOL = {
    "tick": "table_kernel_w",
    "wave.blk.c2o": "conv_gemm_kernel<Layer<256, 256, 1, 1, 1, 1, 0, 0, 0, true, false>",
    "wave.blk.q": "conv_gemm_kernel<Layer<256, 256, 1, 1, 1, 1, 0, 0, 0, false, false>",
    "wave.blk.attn_qk": "conv_gemm_kernel<Layer<256, 384, 1, 1, 1, 1, 0, 0, 1, false, true>",
    "wave.blk.attn_pv": "attn_pv_kernel",
    "wave.tail": "wave_tail_kernel",
    "phone.rb": "conv_gemm_kernel<Layer<256, 256, 5, 1, 1, 1, 0, 1, 0, true, false>",
    "wave.up1": "conv_gemm_kernel<Layer<256, 640, 2, 1, 1, 1, 1, 0, 0, false, false>",
}


# the sources that define the tick launch: its bodies, the table kernel and the table builder (everything they include)
TICK_LAUNCH_SOURCES = ("tick.hip.h", "fuse.hip.h", "rowchain.hip.h", "tail_stages.hip.h", "wave_tail.hip.h", "chain_layers.hip.h", "conv_gemm.hip.h",
                       "fused_small.hip.h", "kernels_misc.hip.h", "gemv.hip.h", "spec_math.hip.h", "ring.h", "engine.h", "batch_tick.hip.h")


# the PMC passes (tools/profile_round.sh, tools/pmc_restamp.sh) run this bench with its defaults: four hops per step
PMC_PASS_HOPS_PER_STEP = 4


def csrc_sha1():
    """Fingerprint of the sources of the tick launch (beatrice-vst_amd/csrc: TICK_LAUNCH_SOURCES): the PMC summaries under
    profiles/ carry the one they were measured at (tools/pmc_summary.py), and a summary taken at other sources is not quoted."""
    import hashlib
    h = hashlib.sha1()
    for name in TICK_LAUNCH_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(REPO, "beatrice-vst_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def pmc_summary_file():
    """The newest committed PMC summary IF it was measured at the present kernel sources, else None."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    rec = json.load(open(files[-1]))
    if rec.get("csrc_sha1") != csrc_sha1():
        return None   # stale: the tick table (or a kernel) changed since the PMC passes
    return rec


def pmc_traffic(kernel_name, streams):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/r*_pmc_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this bench at B = 256,
    FETCH_SIZE doubled per the gfx950 correction).  None when no measurement of the present sources exists."""
    if streams != 256 or kernel_name not in PMC_SYMBOL:
        return None
    rec = pmc_summary_file()
    if rec is None:
        return None
    kernels = rec["kernels"]
    for sym, rec in kernels.items():
        if PMC_SYMBOL[kernel_name] in sym:
            return rec["hbm_bytes_per_launch"]
    return None


def pmc_mfma_busy(kernel_name, streams):
    """SQ_VALU_MFMA_BUSY_CYCLES per launch of the dominant kernel (same committed PMC summary), or None."""
    if streams != 256 or kernel_name not in PMC_SYMBOL:
        return None
    summary = pmc_summary_file()
    if summary is None:
        return None
    for sym, rec in summary["kernels"].items():
        if PMC_SYMBOL[kernel_name] in sym:
            return rec.get("mfma_busy_cycles_per_launch")
    return None


def cpu_baseline(bv, model_dir, seconds):
    """The oracle (a port of the frozen spec; the proprietary beatricelib has no Linux build) on this box's host cores,
    driven by a C loop (oracle/bench_driver.c: the reference's per-hop call sequence, one pthread per stream): 1 stream on
    1 core -- how the plugin runs -- then one stream per core."""
    lib = ctypes.CDLL(os.path.join(REPO, "oracle", "liboracle_bench.so"))
    lib.oracle_bench_threads.restype = ctypes.c_double
    lib.oracle_bench_threads.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    probe = lib.oracle_bench_threads(model_dir.encode(), 1, 200)
    if probe <= 0:
        raise SystemExit("oracle bench driver failed: %g" % probe)
    hops = max(500, int(seconds * probe))
    single = lib.oracle_bench_threads(model_dir.encode(), 1, hops)
    cores = os.cpu_count() or 1
    hops_mt = max(200, int(0.35 * seconds * probe))
    # one thread per hardware thread, and one per two (a core's two hardware threads share its FMA pipes and caches; the
    # loop streams 22 MB of weights per hop and stream, so more threads are not always more frames): the better one is quoted
    tried = {n: lib.oracle_bench_threads(model_dir.encode(), n, hops_mt) for n in sorted({cores, max(1, cores // 2)})}
    best = max(tried, key=tried.get)
    return {"value": round(single, 1), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d hops of 1 synthetic stream through the 1-stream C-ABI of oracle/libbeatrice_oracle.so (gcc -O3 -mavx2 -mfma), "
                      "C driver loop; the proprietary reference beatricelib has no Linux build" % hops,
            "all_cores": {"value": round(tried[best], 1), "cores": best, "hardware_threads": cores,
                          "tried": {str(n): round(v, 1) for n, v in tried.items()},
                          "sample": "%d hops x %d streams, one pthread each (oracle/bench_driver.c)" % (hops_mt, best)}}


def measured_peaks():
    """What this box reaches on the two rooflines (tools/microbench/peaks.hip): float4 copy, dependency-free FP32 MFMA."""
    path = os.path.join(REPO, "tools", "microbench", "libpeaks.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.peaks_hbm_copy_gbs.restype = ctypes.c_double
    lib.peaks_hbm_copy_gbs.argtypes = [ctypes.c_size_t, ctypes.c_int]
    lib.peaks_mfma_f32_tflops.restype = ctypes.c_double
    lib.peaks_mfma_f32_tflops.argtypes = [ctypes.c_int, ctypes.c_int]
    return {"hbm_copy_GBs": round(lib.peaks_hbm_copy_gbs(1 << 30, 5), 1), "mfma_f32_TFLOPs": round(lib.peaks_mfma_f32_tflops(20000, 3), 1),
            "spec": {"hbm_GBs": PEAK_HBM_GBS, "mfma_f32_TFLOPs": PEAK_FP32_MFMA_TFLOPS},
            "how": "1 GiB non-temporal float4 device-to-device copy (read + write); v_mfma_f32_16x16x4_f32 with 2 independent accumulators per wavefront, 16 wavefronts per CU"}


def saturation(bv, models, product, streams=8192, steps=30):
    """Same chain, same kernels, enough streams to leave the launch-latency regime (not the headline:
    BASELINE.json's throughput config is 256 streams per GPU).  Reports end-to-end FLOP/s and the
    roofline fraction of the dominant kernel at that batch size."""
    batch = bv.Batch(models, streams)
    product.BeatriceBatch_FlushSpeaker(batch.h, -1)
    batch.time_steps(5)
    ms = batch.time_steps(steps)
    rows = batch.profile_kernels(repeats=3)
    batch.close()
    total_flops = sum(r["flops"] * r["launches"] for r in rows)
    dom = max(rows, key=lambda r: r["mean_us"] * r["launches"])
    step_s = ms * 1e-3 / steps
    return {"streams": streams, "frames_per_s": round(streams / step_s, 1), "ms_per_step": round(step_s * 1e3, 4),
            "tflops_end_to_end": round(total_flops / step_s / 1e12, 2),
            "mfma_frac_end_to_end": round(total_flops / step_s / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
            "dominant_kernel": dom["name"], "dominant_share": round(dom["mean_us"] * dom["launches"] / sum(r["mean_us"] * r["launches"] for r in rows), 3),
            "dominant_tflops": round(dom["flops"] / (dom["mean_us"] * 1e-6) / 1e12, 2),
            "dominant_frac_of_mfma_peak": round(dom["flops"] / (dom["mean_us"] * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
            "best_gemm_tflops": round(max(r["flops"] / (r["mean_us"] * 1e-6) / 1e12 for r in rows), 2)}


def tick_rate(bv, models, product, torch, streams, flops_per_stream_hop, steps=150, hops=1):
    """Tick pipelining at another batch size / number of hops per step (resident noise input, 64 slots, fill and drain inside the
    timed region)."""
    import time
    n_cycle = 64
    batch = bv.Batch(models, streams, hops_per_step=hops)
    product.BeatriceBatch_FlushSpeaker(batch.h, -1)
    d_in = torch.randn((n_cycle, streams, hops * 160), dtype=torch.float32, device="cuda") * 0.1
    d_out = torch.zeros((n_cycle, streams, hops * 240), dtype=torch.float32, device="cuda")
    out = None
    if product.BeatriceBatch_BindResidentIO(batch.h, d_in.data_ptr(), d_out.data_ptr(), n_cycle) == 0 and \
            product.BeatriceBatch_EnableTickPipeline(batch.h, 1) == 0:
        for n in (40, steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                product.BeatriceBatch_ConvertFramesDevice(batch.h, None, None)
            product.BeatriceBatch_Synchronize(batch.h)
            dt = time.perf_counter() - t0
        fps = streams * hops * steps / dt
        out = {"streams": streams, "hops_per_step": hops, "steps": steps, "frames_per_s": round(fps, 1), "ms_per_step": round(dt / steps * 1e3, 4),
               "tflops_end_to_end": round(fps * flops_per_stream_hop / 1e12, 2),
               "mfma_frac_end_to_end": round(fps * flops_per_stream_hop / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
        product.BeatriceBatch_EnableTickPipeline(batch.h, 0)
    batch.close()
    return out


def block_mode(bv, models, product, streams, steps=200):
    """Bulk / utterance conversion (not the headline, which is one 10 ms hop per step): the same chain with
    H = 2, 4 and 8 consecutive hops per step (BeatriceBatch_CreateBlock), bit-identical results, launch cost
    shared by H hops.  Same resident-input timing as the headline; per-kernel roofline at the largest H."""
    out = {"streams": streams}
    for H in (2, 4, 8):
        batch = bv.Batch(models, streams, hops_per_step=H)
        product.BeatriceBatch_FlushSpeaker(batch.h, -1)
        batch.time_steps(20)
        ms = batch.time_steps(steps)
        rec = {"frames_per_s": round(streams * H * steps / (ms * 1e-3), 1), "ms_per_step": round(ms / steps, 4)}
        if product.BeatriceBatch_EnablePipelining(batch.h, 4) == 0:  # the same steps, four in flight (see the headline's config)
            batch.time_steps(20)
            msp = batch.time_steps(steps)
            rec["pipelined_depth4_frames_per_s"] = round(streams * H * steps / (msp * 1e-3), 1)
            product.BeatriceBatch_EnablePipelining(batch.h, 0)
        if H == 8:
            rows = batch.profile_kernels(repeats=5)
            flops = sum(r["flops"] * r["launches"] for r in rows)
            dom = max(rows, key=lambda r: r["mean_us"] * r["launches"])
            rec.update({"tflops_end_to_end": round(flops / (ms * 1e-3 / steps) / 1e12, 2),
                        "mfma_frac_end_to_end": round(flops / (ms * 1e-3 / steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                        "dominant_kernel": dom["name"], "dominant_launches": dom["launches"],
                        "dominant_us_per_launch": round(dom["mean_us"], 2),
                        "dominant_frac_of_mfma_peak": round(dom["flops"] / (dom["mean_us"] * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)})
        batch.close()
        out["H%d" % H] = rec
    return out


def morph_timing(bv, product, n_real=8):
    """Speaker morphing: one BeatriceBatch_MorphSpeaker call = 385 spherical
    means on the device + re-projection of the entry, against the same 385 solves by the host solver
    (beatrice-vst_amd/host, the bit-exact counterpart of the reference's SphericalAverage) on one core."""
    import ctypes as C
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_model
    tmp = tempfile.TemporaryDirectory()
    make_model.make_model(tmp.name, n_speakers=n_real)
    models = bv.Models(product, tmp.name)
    batch = bv.Batch(models, 4, max_speakers=n_real + 2)
    w = np.zeros(n_real, np.float32)
    w[:min(n_real, 4)] = (0.4, 0.3, 0.2, 0.1)[:min(n_real, 4)]
    product.BeatriceBatch_MorphSpeaker(batch.h, n_real, bv.fptr(w), n_real, 1)
    t0 = time.perf_counter()
    reps = 20
    for i in range(reps):
        product.BeatriceBatch_MorphSpeaker(batch.h, n_real, bv.fptr(w), n_real, 1)
    dev_ms = (time.perf_counter() - t0) * 1e3 / reps
    batch.close()
    out = {"device_ms_per_morph": round(dev_ms, 3), "solves": 385}
    host_path = os.path.join(REPO, "oracle", "libhost_on_oracle.so")
    if os.path.exists(host_path):
        lib = C.CDLL(host_path)
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int)
        lib.BeatriceHost_SphericalMean.argtypes = [C.c_int, C.c_int, f32p, f32p, i32p, C.c_int, C.c_int, f32p]
        t = models.tables
        order = np.argsort(-w, kind="stable").astype(np.int32)
        o256, o128 = np.zeros(256, np.float32), np.zeros(128, np.float32)
        kvp = [np.ascontiguousarray(t.kv[:n_real, tok]) for tok in range(t.kv.shape[1])]
        addp = np.ascontiguousarray(t.additive[:n_real])
        t0 = time.perf_counter()
        lib.BeatriceHost_SphericalMean(256, n_real, addp.ctypes.data_as(f32p), w.ctypes.data_as(f32p), order.ctypes.data_as(i32p), 8, 4,
                                       o256.ctypes.data_as(f32p))
        for pts in kvp:
            lib.BeatriceHost_SphericalMean(128, n_real, pts.ctypes.data_as(f32p), w.ctypes.data_as(f32p), order.ctypes.data_as(i32p), 8, 4,
                                           o128.ctypes.data_as(f32p))
        out["host_ms_per_morph_1core"] = round((time.perf_counter() - t0) * 1e3, 3)
    models.close()
    tmp.cleanup()
    return out


def host_buffer_rate(bv, models, product, streams, steps=200):
    """The boundary's host-buffer variant (BeatriceBatch_ConvertFrames: pageable caller buffers -> pinned staging ->
    PCIe -> chain -> PCIe -> caller buffer, synchronous per step).  PCIe-inclusive; never the headline `value`."""
    batch = bv.Batch(models, streams)
    product.BeatriceBatch_FlushSpeaker(batch.h, -1)
    x = np.stack([bv.synth_audio(160 * 8, seed=s) for s in range(streams)]).reshape(streams, 8, 160)
    xs = [np.ascontiguousarray(x[:, i]) for i in range(8)]
    out = np.zeros((streams, 240), np.float32)
    for i in range(20):
        product.BeatriceBatch_ConvertFrames(batch.h, bv.fptr(xs[i % 8]), bv.fptr(out))
    t0 = time.perf_counter()
    for i in range(steps):
        product.BeatriceBatch_ConvertFrames(batch.h, bv.fptr(xs[i % 8]), bv.fptr(out))
    dt = time.perf_counter() - t0
    res = {"frames_per_s": round(streams * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4),
           "bytes_over_pcie_per_step": streams * (160 + 240) * 4}
    # the same host buffers through the tick pipeline (BeatriceBatch_StreamFrames): uploads, ticks and downloads of
    # neighbouring steps on three HIP streams; fill and flush inside the timed region
    if product.BeatriceBatch_EnableHostStreaming(batch.h, 1) == 0:
        n = 600
        for timed in (False, True):
            t0 = time.perf_counter()
            got = 0
            for i in range(n):
                got += product.BeatriceBatch_StreamFrames(batch.h, bv.fptr(xs[i % 8]), bv.fptr(out))
            while product.BeatriceBatch_StreamFlush(batch.h, bv.fptr(out)) == 1:
                got += 1
            dt = time.perf_counter() - t0
        res["streamed_through_tick_pipeline"] = {"frames_per_s": round(streams * n / dt, 1), "ms_per_step": round(dt / n * 1e3, 4),
                                                 "steps": n, "steps_returned": got,
                                                 "delay_steps": int(product.BeatriceBatch_HostStreamDelay(batch.h))}
        product.BeatriceBatch_EnableHostStreaming(batch.h, 0)
    batch.close()
    # ... and with two hops per step ([B][320] in, [B][480] out per call)
    batch = bv.Batch(models, streams, hops_per_step=2)
    product.BeatriceBatch_FlushSpeaker(batch.h, -1)
    x2 = [np.ascontiguousarray(np.concatenate([xs[(2 * i) % 8], xs[(2 * i + 1) % 8]], axis=1)) for i in range(4)]
    out2 = np.zeros((streams, 480), np.float32)
    if product.BeatriceBatch_EnableHostStreaming(batch.h, 1) == 0:
        n = 300
        for timed in (False, True):
            t0 = time.perf_counter()
            got = 0
            for i in range(n):
                got += product.BeatriceBatch_StreamFrames(batch.h, bv.fptr(x2[i % 4]), bv.fptr(out2))
            while product.BeatriceBatch_StreamFlush(batch.h, bv.fptr(out2)) == 1:
                got += 1
            dt = time.perf_counter() - t0
        res["streamed_through_tick_pipeline_two_hops_per_step"] = {"frames_per_s": round(2 * streams * n / dt, 1), "ms_per_step": round(dt / n * 1e3, 4),
                                                                   "steps": n, "steps_returned": got}
        product.BeatriceBatch_EnableHostStreaming(batch.h, 0)
    batch.close()
    return res


def wrapper_around_ticks(bv, models, product, torch, streams, calls=400):
    """The reference's whole per-instance wrapper (any host rate, gains, FIFO: processor_core_2.cc:24-48) around the tick pipeline with
    resident host-rate blocks: 44.1 kHz callers with 10 ms blocks on a batch of four hops per step (BeatriceBatch_BindResidentBlocks), and
    callers at four different rates with clocks per stream (BeatriceBatch_BindResidentBlocksRagged, one hop per step).  Drain inside
    the timed region; frames = 10 ms model hops."""
    import ctypes as C
    res = {}
    sr, block, H = 44100, 441, 4
    batch = bv.Batch(models, streams, hops_per_step=H)
    product.BeatriceBatch_FlushSpeaker(batch.h, -1)
    if product.BeatriceBatch_ConfigureWrapper(batch.h, float(sr)) == 0:
        delay = int(product.BeatriceBatch_ResidentBlocksDelayFor(batch.h, block))
        slots = delay + 8
        d_in = (0.1 * torch.randn((slots, streams, 1, block), device="cuda")).contiguous()
        d_out = torch.zeros_like(d_in)
        if product.BeatriceBatch_BindResidentBlocks(batch.h, d_in.data_ptr(), d_out.data_ptr(), 1, block, slots) == 0:
            for timed in (False, True):
                t0 = time.perf_counter()
                for _ in range(calls):
                    product.BeatriceBatch_ProcessBlocksDevice(batch.h, None, None, 1, block)
                product.BeatriceBatch_Synchronize(batch.h)
                dt = time.perf_counter() - t0
            res["uniform_44k1_four_hops_per_step"] = {"frames_per_s": round(streams * calls * block * 100.0 / sr / dt, 1), "ms_per_call": round(dt / calls * 1e3, 4),
                                                      "calls": calls, "block": block, "delay_calls": delay}
            product.BeatriceBatch_BindResidentBlocks(batch.h, None, None, 0, 0, 0)
        del d_in, d_out
    batch.close()
    rates = [(44100.0, 441), (48000.0, 480), (96000.0, 960), (32000.0, 320)]
    batch = bv.Batch(models, streams)
    product.BeatriceBatch_FlushSpeaker(batch.h, -1)
    rs = (C.c_double * streams)(*[rates[s % 4][0] for s in range(streams)])
    ns = (C.c_int * streams)(*[rates[s % 4][1] for s in range(streams)])
    if product.BeatriceBatch_ConfigureWrapperRates(batch.h, rs) == 0:
        slots, cap = int(product.BeatriceBatch_TickStages(batch.h)) + 8, 960
        d_in = (0.1 * torch.randn((slots, streams, cap), device="cuda")).contiguous()
        d_out = torch.zeros_like(d_in)
        if product.BeatriceBatch_BindResidentBlocksRagged(batch.h, d_in.data_ptr(), d_out.data_ptr(), 1, cap, slots) == 0:
            for timed in (False, True):
                t0 = time.perf_counter()
                for _ in range(calls):
                    product.BeatriceBatch_ProcessBlocksRaggedDevice(batch.h, ns)
                product.BeatriceBatch_Synchronize(batch.h)
                dt = time.perf_counter() - t0
            res["clocks_per_stream_four_rates"] = {"frames_per_s": round(streams * calls / dt, 1), "ms_per_call": round(dt / calls * 1e3, 4), "calls": calls,
                                                   "rates": [r for r, _ in rates]}
            product.BeatriceBatch_BindResidentBlocksRagged(batch.h, None, None, 0, 0, 0)
        del d_in, d_out
    batch.close()
    return res


def hop_synchronous(bv, models, product, streams, steps=300):
    """The same chain with pipelining off: every step complete before the next starts (what a real-time server that
    receives one hop per stream every 10 ms runs); its step time is the latency of a 256-stream hop."""
    batch = bv.Batch(models, streams)
    product.BeatriceBatch_FlushSpeaker(batch.h, -1)
    batch.time_steps(30)
    ms = batch.time_steps(steps)
    batch.close()
    return {"frames_per_s": round(streams * steps / (ms * 1e-3), 1), "ms_per_step": round(ms / steps, 4),
            "x_realtime_per_stream": round(10.0 / (ms / steps), 1)}


def latency_b1(model_dir, hops=100000, warm=2000, paced_hops=1500):
    """BASELINE.json configs[1]: 1 stream, 1 speaker, hop-synchronous 1-stream C-ABI -- timed by examples/latency_b1 (a C++ loop
    around the reference's three per-hop calls with clock_gettime; no interpreter between the clock and the calls), in a process
    of its own.  100 000 hops back to back: p50 / p99 / p99.9 / max, the counts of hops over 1 ms and over the 10 ms budget and how
    many of those coincide with the OS preempting the thread; then `paced_hops` hops arriving every 10 ms on a SCHED_FIFO thread --
    what a DAW's audio callback sees, the GPU idle between hops -- as `paced_10ms`.  Round 6: the pitch estimator's hop runs beside the phone call
    (pitch_hops_claimed; include/beatrice_batch.h BeatriceHip_PitchSpeculation); `calls_one_by_one` is the same loop with that switched off."""
    import subprocess
    exe = os.path.join(REPO, "examples", "latency_b1")
    if not os.path.exists(exe):
        raise RuntimeError("examples/latency_b1 was not built (make -C beatrice-vst_amd)")

    def run(*args, env=None):
        r = subprocess.run([exe, model_dir] + [str(a) for a in args], capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
        if r.returncode != 0:
            raise RuntimeError("examples/latency_b1 failed (%d): %s" % (r.returncode, r.stderr[-500:]))
        return json.loads(r.stdout.strip().splitlines()[-1])

    out = run(hops, warm, 0, "--histogram")
    # the same loop with the pitch estimator's hop NOT started inside the phone call (csrc/abi.hip "pre-execution"; rounds 1-5's figure)
    plain = run(min(hops, 20000), warm, 0, "--histogram", env={"BEATRICE_HIP_NO_SPECULATION": "1"})
    out["calls_one_by_one"] = {k: plain[k] for k in ("hops", "p50_us", "p99_us", "p999_us", "max_us", "per_call_p50_us", "pitch_hops_claimed", "checksum")}
    out["calls_one_by_one"]["note"] = "BEATRICE_HIP_NO_SPECULATION=1: EstimatePitch1 runs its hop itself, after ExtractPhone1 has returned"
    if paced_hops > 0:
        paced = run(paced_hops, 100, 0, "--histogram", "--period-us", 10000, "--rt")
        out["paced_10ms"] = {k: paced[k] for k in ("workload", "period_us", "realtime_thread", "hops", "p50_us", "p99_us", "p999_us", "max_us", "hops_over_1ms",
                                                    "hops_over_10ms", "involuntary_context_switches", "per_call_p50_us", "pitch_hops_claimed")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--streams", type=int, default=DEFAULT_STREAMS[DEFAULT_CONFIG], help="streams per GPU")
    ap.add_argument("--speakers", type=int, default=None)
    ap.add_argument("--hops-per-step", type=int, default=None, choices=(1, 2, 4),
                    help="10 ms hops of every stream per step (tick pipeline: hops per stage per launch); default 4 with the tick "
                         "pipeline (round 4: 2; rounds 1-3: 1 -- the same run at those definitions is in the line), 1 otherwise")
    ap.add_argument("--config", type=int, default=DEFAULT_CONFIG, choices=(2, 3, 4),
                    help="BASELINE.json configs index: 2 = 256 streams/GPU, 1 speaker (default, the headline); "
                         "3 = 256 streams/GPU, 64 rotating speakers, VQ k=4; 4 = 64 streams/GPU, 48 kHz stereo, wrapper on the device")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--pipeline-depth", type=int, default=4, choices=(0, 2, 3, 4),
                    help="BeatriceBatch_EnablePipelining: stages of the per-hop chain that overlap across consecutive steps "
                         "while steps are enqueued ahead (0 = off: one HIP stream, in order)")
    ap.add_argument("--pipeline", choices=("tick", "stages", "off"), default="tick",
                    help="how steps enqueued ahead of their completion overlap: tick = every layer its own pipeline stage, one "
                         "launch per tick on one stream (BeatriceBatch_EnableTickPipeline); stages = --pipeline-depth stages on "
                         "as many HIP streams; off = in order")
    ap.add_argument("--copy-io", action="store_true",
                    help="device-to-device copy of each hop into / out of the library's own buffers instead of "
                         "binding the resident audio buffers (BeatriceBatch_BindResidentIO)")
    ap.add_argument("--device-warm-ms", type=float, default=1000.0,
                    help="milliseconds of the workload's own steps (untimed, drained) before the warm-up steps, so that a run of a few milliseconds "
                         "executes at the clocks of a device that has been converting, not at an idle one's (+5 %% at 20 steps); 0 = off; stated in config")
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline / B=1 latency / kernel profile")
    ap.add_argument("--total-streams", type=int, default=None,
                    help="strong scaling: this many streams in total, split evenly over the GPUs (default: --streams per GPU, weak scaling)")
    ap.add_argument("--placement", choices=("round-robin", "speaker-affine"), default=None,
                    help="configs[3]: which speakers a rank's streams start on (speaker-affine: speaker mod world == rank; the default "
                         "when there are several ranks and at least as many speakers, SURVEY.md 8e; round-robin otherwise)")
    ap.add_argument("--collectives", choices=("nccl", "gloo"), default="nccl",
                    help="nccl = RCCL over xGMI, one GPU per rank (the product path); gloo = test hook: the same flow with several ranks "
                         "sharing the GPUs of a smaller box (tests/test_gpu_two_ranks_load_path.py)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, run one all-reduce over gloo and exit (CPU check that --gpus N launches N ranks)")
    a = ap.parse_args()

    # --gpus N without a launcher: start N ranks of this script under torch.distributed.run (one process per GPU)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        port = os.environ.get("MASTER_PORT") or str(29400 + os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus, or without a launcher)"
                         % (a.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    import torch
    import torch.distributed as dist
    if a.rendezvous_only:
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"rendezvous": "gloo", "n_gpus": a.gpus, "ranks_seen": int(t.item())}))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if a.collectives == "gloo":  # test hook: several ranks on however many GPUs the box has (RCCL wants one device per rank)
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if a.collectives == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    bv = load_pkg()
    product = bv.bind_batch(bv.load_product())
    if hasattr(product, "BeatriceHip_SetDevice"):   # every object of this rank on this rank's GPU, whatever the thread's current device
        product.BeatriceHip_SetDevice(local_rank)
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_model

    if a.config == 4 and a.streams == DEFAULT_STREAMS[DEFAULT_CONFIG]:
        a.streams = DEFAULT_STREAMS[4]   # batch 512 over 8 GPUs
    if a.speakers is None:
        a.speakers = 64 if a.config == 3 else 1
    scaling = "weak"
    if a.total_streams:
        if a.total_streams % world:
            raise SystemExit("--total-streams must be a multiple of --gpus")
        a.streams = a.total_streams // world
        scaling = "strong"
    B = a.streams
    if a.hops_per_step is None:
        a.hops_per_step = DEFAULT_HOPS_PER_STEP_TICK if (a.pipeline == "tick" and not a.copy_io) else 1
    H = a.hops_per_step
    if H > 1 and (a.pipeline != "tick" or a.copy_io):
        raise SystemExit("--hops-per-step 2 / 4 is the tick pipeline's form (resident I/O)")
    tmp = tempfile.TemporaryDirectory()
    model_dir = tmp.name
    if rank == 0:
        make_model.make_model(model_dir, n_speakers=a.speakers)
    shard = load_shard()

    # One file read, on rank 0; the packed parameter blobs and the raw speaker tables reach the other GPUs device to
    # device (RCCL broadcast over xGMI), straight into the memory the kernels read.
    class Loaded:
        pass
    m = Loaded()
    m.abi = product
    m.phone, m.pitch = product.CreatePhoneExtractor(), product.CreatePitchEstimator()
    m.wave, m.embed = product.CreateWaveformGenerator(), product.CreateEmbeddingSetter()
    bcast_bytes, load_path = shard.load_models_from_rank0(product, {"phone": m.phone, "pitch": m.pitch, "wave": m.wave, "embed": m.embed},
                                                           model_dir, rank, world, dist, torch)
    if rank == 0:
        m.tables = bv.SpeakerTables(product, model_dir)
    batch = bv.Batch(m, B, max_speakers=a.speakers + 1, hops_per_step=H, upload_tables=(rank == 0))
    table_bytes, table_path = shard.share_speaker_tables(product, batch.h, a.speakers + 1, rank, world, dist, torch,
                                                         host_tables=m.tables if rank == 0 else None)
    bcast_bytes += table_bytes
    if rank != 0:
        batch.apply_defaults()
    if a.no_graph:
        product.BeatriceBatch_EnableGraph(batch.h, 0)
    if a.placement is None:
        a.placement = "speaker-affine" if (world > 1 and a.speakers >= world) else "round-robin"
    if a.placement == "speaker-affine":
        current_speaker = [shard.affine_speaker(rank, world, s, a.speakers) for s in range(B)]
    else:  # global stream index modulo the table size
        current_speaker = [(rank * B + s) % a.speakers for s in range(B)]
    for s in range(B):
        product.BeatriceBatch_SetTargetSpeaker(batch.h, s, current_speaker[s])
    product.BeatriceBatch_FlushSpeaker(batch.h, -1)
    if a.config == 3:
        product.BeatriceBatch_SetVQNumNeighbors(batch.h, -1, 4)

    # synthetic audio, resident on the device: 64 steps (of H hops) x B streams, cycled
    n_cycle = 64
    audio = np.stack([bv.synth_audio(160 * H * n_cycle, seed=rank * 100000 + s, silence_gap=(s % 10 == 3)) for s in range(B)])  # (every tenth stream: 0.5 s of digital silence)
    audio = np.ascontiguousarray(audio.reshape(B, n_cycle, H * 160).transpose(1, 0, 2))
    d_audio = torch.from_numpy(audio).cuda()
    resident = a.config != 4 and not a.copy_io
    d_out = torch.zeros((n_cycle if resident else 1, B, H * 240), dtype=torch.float32, device="cuda")
    base, hop_bytes = d_audio.data_ptr(), B * H * 160 * 4
    a.pipeline_request = a.pipeline
    tick48 = False
    if a.config == 4 or a.pipeline == "off":
        a.pipeline, a.pipeline_depth = "off", 0
    pipelined = a.pipeline_depth if a.pipeline == "stages" else 0
    if resident:  # the 64 resident hops are the slots: every step reads one and writes one, no copy
        if product.BeatriceBatch_BindResidentIO(batch.h, d_audio.data_ptr(), d_out.data_ptr(), n_cycle):
            print("bench: BindResidentIO refused, copying each hop in and out", file=sys.stderr)
            resident = False
            d_out = torch.zeros((1, B, 240), dtype=torch.float32, device="cuda")
    tick = a.pipeline == "tick"
    if tick and (not resident or product.BeatriceBatch_EnableTickPipeline(batch.h, 1)):
        print("bench: tick pipelining refused (needs resident I/O, one hop per step), using stage pipelining", file=sys.stderr)
        tick, pipelined = False, a.pipeline_depth
    if pipelined and product.BeatriceBatch_EnablePipelining(batch.h, pipelined):
        print("bench: EnablePipelining(%d) refused, running in order" % pipelined, file=sys.stderr)
        pipelined = 0

    if a.config == 4:  # 48 kHz stereo blocks, resident: [n_cycle][B][2][480]
        a48 = np.stack([np.stack([bv.synth_audio(480 * H * n_cycle, seed=rank * 100000 + 2 * s + c, sr=48000) for c in range(2)])
                        for s in range(B)])
        a48 = np.ascontiguousarray(a48.reshape(B, 2, n_cycle, H, 480).transpose(2, 0, 3, 1, 4))   # [n_cycle][B][H][2][480]
        d_audio48 = torch.from_numpy(a48).cuda()
        d_out48 = torch.zeros((n_cycle, B, H, 2, 480), dtype=torch.float32, device="cuda")
        base48, blk_bytes = d_audio48.data_ptr(), B * H * 2 * 480 * 4
        # throughput form: the 64 resident 48 kHz blocks are the slots, the tick pipeline runs between the two resamplers
        tick48 = a.pipeline_request == "tick" and product.BeatriceBatch_BindResidentIO48k(batch.h, d_audio48.data_ptr(), d_out48.data_ptr(), 2, n_cycle) == 0
        if H > 1 and not tick48:
            raise SystemExit("configs[4] with two blocks per step needs the 48 kHz wrapper around the tick pipeline")

    if product.BeatriceBatch_Prepare(batch.h):  # graph capture now, not inside the first (possibly timed) steps
        raise SystemExit("Prepare failed")

    # configs[3]'s speaker switches: stream s moves to the next speaker every 200 hops, staggered by stream index (it switches at hops
    # congruent to -(s * 200 // B) modulo 200).  The whole run's schedule is worked out BEFORE the timed loop as one (streams, speakers) array
    # pair per step -- what a server's control plane hands over --, so that a step's host work is one BeatriceBatch_SetTargetSpeakers call
    # (settings travel with the step: the switches of its hops, before it)
    switchers = [[s for s in range(B) if (r + s * 200 // B) % 200 == 0] for r in range(200)]
    schedule = []

    def schedule_until(n_steps):
        while len(schedule) < n_steps:
            i = len(schedule)
            moved = []
            if a.config == 3 and i > 0:
                for hop in range(i * H, (i + 1) * H):
                    for s in switchers[hop % 200]:
                        current_speaker[s] = (current_speaker[s] + 1) % a.speakers
                        moved.append(s)
            n_mv = len(moved)
            schedule.append((n_mv, (ctypes.c_int * n_mv)(*moved), (ctypes.c_int * n_mv)(*[current_speaker[s] for s in moved])) if n_mv else None)

    if a.config == 3:
        schedule_until(a.warmup + a.steps + 700)

    fed = [0]   # steps fed so far (the resident ring's slot and configs[3]'s switch schedule follow it)

    def step(_unused=None):
        i = fed[0]
        fed[0] += 1
        if a.config == 3:
            if i >= len(schedule):
                schedule_until(i + 1)
            if schedule[i] is not None:   # one call for all of the step's switches
                product.BeatriceBatch_SetTargetSpeakers(batch.h, *schedule[i])
        if a.config == 4 and tick48:
            rc = product.BeatriceBatch_ConvertBlocks48kDevice(batch.h, None, None, 2)
        elif a.config == 4:
            rc = product.BeatriceBatch_ConvertBlocks48kDevice(batch.h, base48 + (i % n_cycle) * blk_bytes, d_out48.data_ptr(), 2)
        elif resident:
            rc = product.BeatriceBatch_ConvertFramesDevice(batch.h, None, None)
        else:
            rc = product.BeatriceBatch_ConvertFramesDevice(batch.h, base + (i % n_cycle) * hop_bytes, d_out.data_ptr())
        if rc:
            raise SystemExit("Convert* failed: %d" % rc)

    if a.device_warm_ms > 0:
        # Bring the GPU to the clocks of a server that has been converting for a while: the workload's OWN steps, untimed, drained, before the W warm-up
        # steps (stated in `config`).  A run of a few milliseconds otherwise executes at whatever clocks the device was left at: 20 steps + drain
        # measured 3.59 M frames/s after 50 ms of idling, 3.96-4.09 M after a second of torch matmuls with synchronisation gaps (rounds 3-5's
        # warm-up), 4.15-4.19 M after 300 of its own steps (tools/debug/clock_state_probe.py, profiles/r06_notes.md section 9).
        t_end = time.perf_counter() + a.device_warm_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(64):
                step()
            product.BeatriceBatch_Synchronize(batch.h)
        torch.cuda.synchronize()
        if a.config == 3:
            schedule_until(fed[0] + a.warmup + a.steps + 700)
    for i in range(a.warmup):
        step(i)
    product.BeatriceBatch_Synchronize(batch.h)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    enqueue_s = time.perf_counter() - t0   # host time to enqueue all steps (the GPU may lag behind)
    if product.BeatriceBatch_Synchronize(batch.h):  # both of the batch's streams (torch only knows its own)
        raise SystemExit("Synchronize failed")
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    if world > 1:
        dist.barrier()
        elapsed = shard.max_over_ranks(elapsed, world, dist, torch, "cuda")
    per_rank_elapsed = shard.gather_over_ranks(own_elapsed, world, dist, torch, "cuda") if world > 1 else [own_elapsed]
    # the host's own work per step, measured where the device cannot push back: 12 steps (fewer than the 16 settings snapshots the
    # library lets the host run ahead) into a drained pipeline, enqueue time only
    t_probe = time.perf_counter()
    for i in range(12):
        step(a.warmup + a.steps + i)
    host_work_s = (time.perf_counter() - t_probe) / 12
    product.BeatriceBatch_Synchronize(batch.h)
    out_rms = float((d_out48 if a.config == 4 else d_out).float().pow(2).mean().sqrt().item())

    if rank == 0:
        frames = world * B * a.steps * H
        res = {
            "metric": "audio frames/sec (24 kHz out, 10 ms hop)", "value": round(frames / elapsed, 1), "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {2: "BASELINE.json configs[2]: %d concurrent streams per GPU, %d speaker(s), 10 ms hop "
                                       "(160 in @16 kHz -> 240 out @24 kHz), synthetic weights of MODEL_SPEC v1; a step = %d consecutive hop(s) of every stream" % (B, a.speakers, H),
                                    3: "BASELINE.json configs[3] per-GPU share: %d streams, %d speakers, every stream switches "
                                       "speaker every 200 hops (K/V blocks one per hop), VQ k=4" % (B, a.speakers),
                                    4: "BASELINE.json configs[4] per-GPU share: %d streams of 48 kHz stereo, downmix + resample "
                                       "wrapper on the device, 480-sample blocks, %d block(s) per stream and step" % (B, H)}[a.config],
                       "streams_per_gpu": B, "speakers": a.speakers, "hops_per_step": H, "frames_per_step": world * B * H, "hipgraph": not a.no_graph, "device_warm_ms": a.device_warm_ms, "device_warm": "the workload's own steps for device_warm_ms, untimed and drained, before the warm-up steps",
                       "pipelining": ("tick: every layer of the chain its own pipeline stage (%d stages), one launch per tick on one HIP "
                                      "stream, stage s works on the step fed s ticks earlier (%d hop(s) of every stream per stage per launch); "
                                      "steps enqueued without waiting; the timed region includes the %d ticks that drain the pipeline"
                                      % (product.BeatriceBatch_TickStages(batch.h), H, product.BeatriceBatch_TickStages(batch.h) - 1)) if (tick or tick48)
                                     else ("%d stages of the chain on %d HIP streams; stage s of step t+1 overlaps stage s+1 of step t, "
                                      "steps enqueued without waiting; GPU_MAX_HW_QUEUES=%s" % (pipelined, pipelined, os.environ.get("GPU_MAX_HW_QUEUES"))) if pipelined
                                     else "off: one stream, in order",
                       "io": "resident 48 kHz stereo blocks, 64 per stream cycled, bound as I/O slots; resamplers and FIFO on the device either side of the tick pipeline" if tick48
                             else "resident device buffers, 64 steps per stream cycled, bound as I/O slots (no per-step copy)" if resident
                             else "resident device buffers, one device-to-device copy in and out per step",
                       "parallelism": "streams sharded over %d GPU(s), no per-hop collective; load: one file read on rank 0, "
                                      "%d bytes of parameters (%s) and speaker tables (%s) broadcast over RCCL" % (world, bcast_bytes, load_path, table_path),
                       "placement": a.placement if a.config == 3 else "n/a"},
            "ms_per_hop": round(1e3 * elapsed / (a.steps * H), 4),   # (of all streams: ms_per_step / hops_per_step)
            "x_realtime_per_stream": round(a.steps * H / elapsed / 100.0, 2), "output_rms": round(out_rms, 4),
            # what "N concurrent streams" means in this mode: a throughput (offline / faster-than-real-time) pipeline -- a step's samples
            # appear n_stages - 1 launches after its input, with n_stages steps of H hops of every stream in flight
            "in_flight": ({"steps": int(product.BeatriceBatch_TickStages(batch.h)), "audio_ms_per_stream": int(product.BeatriceBatch_TickStages(batch.h)) * H * 10,
                           "wall_ms": round(int(product.BeatriceBatch_TickStages(batch.h)) * 1e3 * elapsed / a.steps, 3)} if (tick or tick48) else None),
            "host_enqueue_ms_per_step": round(1e3 * enqueue_s / a.steps, 4),   # (includes waiting for the device once the host is 16 settings snapshots ahead)
            "host_work_ms_per_step": round(1e3 * host_work_s, 4),               # (12 steps into a drained pipeline: the host's own work)
            "per_rank_frames_per_s": [round(B * a.steps * H / e, 1) for e in per_rank_elapsed],   # (a straggler shows; `value` uses the MAX-reduced time)
        }
        tick_roof = None
        if tick48:  # configs[4]: the wrapper kernels run beside each tick launch; per-kernel figures below are of the chain in order
            product.BeatriceBatch_BindResidentIO48k(batch.h, None, None, 0, 0)
        if tick:
            # the dominant kernel of the headline IS the tick launch: refill the pipeline, then time 64 more launches between
            # one pair of HIP events on the batch's stream (BeatriceBatch_TimeTickLaunch)
            stages = product.BeatriceBatch_TickStages(batch.h)
            for i in range(stages + 2):
                step(a.warmup + a.steps + 12 + i)
            us, fl, by = ctypes.c_float(0), ctypes.c_double(0), ctypes.c_double(0)
            # (clocks and caches settle over a few hundred ticks after the refill above -- 83 -> 81 -> 79 -> 77 us over four calls on
            #  an idle-cooled device, tools/debug/time_tick.py -- so the figure is the mean of the last three of eight calls of 64
            #  ticks each; every call's value is kept in `launch_us_per_call`)
            calls = []
            rc = 0
            for _ in range(8):
                rc = product.BeatriceBatch_TimeTickLaunch(batch.h, 64, ctypes.byref(us), ctypes.byref(fl), ctypes.byref(by))
                calls.append(round(us.value, 2))
            if rc == 0:
                us.value = sum(calls[-3:]) / 3.0
                ach = fl.value / (us.value * 1e-6) / 1e12
                tick_roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                             # (the PMC passes of tools/profile_round.sh run the default workload: quoted for that one only)
                             "traffic": pmc_traffic("tick", B) if (a.config == 2 and a.speakers == 1 and H == PMC_PASS_HOPS_PER_STEP) else None,
                             "traffic_unit": "bytes per launch (PMC, profiles/)",
                             "algorithmic_bytes": int(by.value), "algorithmic_flops": int(fl.value),
                             "kernel": "tick launch (fuse::table_kernel_w: one workgroup-table launch holding every stage of the "
                                       "chain, %d stages each on its own step of %d hop(s) per stream)" % (stages, H),
                             "launches_per_hop": 1.0 / H, "hops_per_launch": H, "mean_us_per_launch": round(us.value, 2), "launch_us_per_call": calls,
                             "share_of_chain": round(us.value * 1e-3 / (1e3 * elapsed / a.steps), 3) if a.steps >= 200 else None}
                summary = pmc_summary_file()
                tick_roof["traffic_measured_at"] = ({"csrc_sha1": summary["csrc_sha1"], "commit": summary.get("commit")} if summary else
                                                    "no PMC pass at the present kernel sources (csrc_sha1 %s): traffic not quoted" % csrc_sha1()[:12])
                busy = pmc_mfma_busy("tick", B)
                if busy is not None:
                    tick_roof["mfma_busy_cycles"] = busy
                    tick_roof["mfma_pipe_busy_frac"] = round(busy / (us.value * 1e-6 * 2.4e9 * 1024), 4)
            product.BeatriceBatch_Synchronize(batch.h)
            if a.steps < 200:  # the timed region of a short run is mostly pipeline fill and drain: steady state beside it
                t1 = time.perf_counter()
                for i in range(600):
                    step(i)
                product.BeatriceBatch_Synchronize(batch.h)
                res["steady_state"] = {"steps": 600, "frames_per_s": round(B * H * 600 / (time.perf_counter() - t1), 1),
                                       "note": "same loop, 600 steps: fill and drain (%d ticks) amortised" % (stages - 1)}
        if not a.no_extras:
            # per-kernel timing with HIP events on the library's own stream (eager, 10 launches per bracket), chain in order
            if tick:
                product.BeatriceBatch_EnableTickPipeline(batch.h, 0)
            if H > 1:   # the per-kernel table below is of the in-order chain at ONE hop per step: a batch of its own
                batch.close()
                batch = bv.Batch(m, B, max_speakers=a.speakers + 1)
                for s_ in range(B):
                    product.BeatriceBatch_SetTargetSpeaker(batch.h, s_, current_speaker[s_])
                product.BeatriceBatch_FlushSpeaker(batch.h, -1)
                if a.config == 3:
                    product.BeatriceBatch_SetVQNumNeighbors(batch.h, -1, 4)
            rows = batch.profile_kernels(repeats=10)
            for r in rows:
                r["total_us"] = r["mean_us"] * r["launches"]
            total_us = sum(r["total_us"] for r in rows)
            dom = max(rows, key=lambda r: r["total_us"])
            t_mfma = dom["flops"] / (PEAK_FP32_MFMA_TFLOPS * 1e12)
            t_hbm = dom["bytes"] / (PEAK_HBM_GBS * 1e9)
            if t_mfma >= t_hbm:
                ach = dom["flops"] / (dom["mean_us"] * 1e-6) / 1e12
                roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4)}
            else:
                ach = dom["bytes"] / (dom["mean_us"] * 1e-6) / 1e9
                roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach / PEAK_HBM_GBS, 4)}
            busy = pmc_mfma_busy(dom["name"], B)
            if busy is not None:  # fraction of the chip's 1024 MFMA pipes kept busy over the launch (2.4 GHz)
                roof["mfma_busy_cycles"] = busy
                roof["mfma_pipe_busy_frac"] = round(busy / (dom["mean_us"] * 1e-6 * 2.4e9 * 1024), 4)
            roof.update({"traffic": pmc_traffic(dom["name"], B), "traffic_unit": "bytes per launch (PMC, profiles/)",
                         "algorithmic_bytes": int(dom["bytes"]), "algorithmic_flops": int(dom["flops"]), "kernel": dom["name"], "launches_per_hop": dom["launches"],
                         "mean_us_per_launch": round(dom["mean_us"], 2),
                         "share_of_chain": round(dom["total_us"] / total_us, 3)})
            if tick_roof is not None:
                res["roofline_in_order_chain"] = roof   # dominant kernel of the in-order chain, for comparison
                roof = tick_roof
            res["roofline"] = roof
            chain_flops = sum(r["flops"] * r["launches"] for r in rows)
            res["chain"] = {"launches_per_hop": sum(r["launches"] for r in rows),
                            "sum_kernel_us": round(total_us, 1), "gflop_per_step": round(chain_flops / 1e9, 3),
                            "tflops_end_to_end": round(chain_flops * H / (elapsed / a.steps) / 1e12, 2),
                            "mfma_frac_end_to_end": round(chain_flops * H / (elapsed / a.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                            "state_bytes_per_stream": int(product.BeatriceBatch_StateBytes(batch.h)) // B}
            res["kernels"] = [{"name": r["name"], "n": r["launches"], "us": round(r["mean_us"], 2)}
                              for r in sorted(rows, key=lambda r: -r["total_us"])[:12]]
            if world == 1:
                batch.close()  # its streams would share the hardware queues with those of the batches measured below
                batch = None
                res["hop_synchronous"] = hop_synchronous(bv, m, product, B)
                res["hop_synchronous_frames_per_s"] = res["hop_synchronous"]["frames_per_s"]
                res["saturation"] = saturation(bv, m, product)
                res["saturation"]["tick_pipelined"] = tick_rate(bv, m, product, torch, 1024, res["chain"]["gflop_per_step"] * 1e9 / B)
                if tick and H > 1 and a.config == 2:   # the same K steps at the headline's earlier definitions: ONE hop per step (rounds 1-3), TWO (round 4); fill and drain inside
                    res["one_hop_per_step"] = tick_rate(bv, m, product, torch, B, res["chain"]["gflop_per_step"] * 1e9 / B, steps=a.steps)
                    # (first-class beside `value`: the same K steps under the headline's round 1-3 definition, a step = ONE hop of every stream)
                    res["frames_per_s_at_one_hop_per_step"] = res["one_hop_per_step"].get("frames_per_s")
                    if H > 2:
                        res["two_hops_per_step"] = tick_rate(bv, m, product, torch, B, res["chain"]["gflop_per_step"] * 1e9 / B, steps=a.steps, hops=2)
                        res["same_hop_count_at_two_hops_per_step"] = tick_rate(bv, m, product, torch, B, res["chain"]["gflop_per_step"] * 1e9 / B, steps=a.steps * H // 2, hops=2)
                res["block_mode"] = block_mode(bv, m, product, B)
                res["morph"] = morph_timing(bv, product)
                res["host_buffer_variant"] = host_buffer_rate(bv, m, product, B)
                res["any_rate_wrapper_around_ticks"] = wrapper_around_ticks(bv, m, product, torch, B)
                res["latency_b1"] = latency_b1(model_dir)
                res["cpu_baseline"] = cpu_baseline(bv, model_dir, a.cpu_seconds)
                peaks = measured_peaks()
                if peaks:
                    res["measured_peaks"] = peaks
                    if res["roofline"]["bound"] == "mfma" and peaks["mfma_f32_TFLOPs"] > 0:
                        res["roofline"]["frac_of_measured_peak"] = round(res["roofline"]["achieved"] / peaks["mfma_f32_TFLOPs"], 4)
        print(json.dumps(res))
    if batch is not None:
        batch.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
