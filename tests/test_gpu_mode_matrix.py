"""The mode lattice of include/beatrice_batch.h, walked cell by cell ("a mode lattice nobody can hold in their head ...
no test walks the forbidden cells for 'refuses cleanly, state untouched'").

The header's table "which entry point in which mode" is restated here as MATRIX: for every mode (a way a batch is set up: in order, stage
pipelining, resident I/O, tick mode, host streaming, the 48 kHz blocks around the ticks, resident blocks around the ticks with one set of
clocks and with clocks per stream, the in-order silent-block rule) the entry points that must REFUSE with -1.  For every mode and every
hops-per-step the mode exists at, two identical batches run the same steps through the mode's own entry point; one of them is asked, between
steps, for every entry point its mode refuses -- each must return -1 -- and must still produce, to the bit, what the batch that never asked
produces (a refusal that drained, re-bound, advanced a clock or cleared a flag would show).  The ALLOWED cells are what the other -m gpu tests
run against the oracle (tests/test_gpu_tick_*.py, test_gpu_wrapper*.py, test_gpu_host_streaming.py, test_gpu_resident_io.py ...); here every
mode's own entry point is seen to return 0 and to produce sound."""
import ctypes as C

import numpy as np
import pytest

import wrapperlib
from tick_driver import Hip

pytestmark = pytest.mark.gpu

B = 3
CH = 1


class Ctx:
    """a batch in one mode + the buffers the probes need (all valid: only the MODE can be the reason for a refusal)"""

    def __init__(self, bv, product, model_dir, H):
        self.bv, self.H, self.hip = bv, H, Hip()
        self.m = bv.Models(product, model_dir)
        self.batch = bv.Batch(self.m, B, hops_per_step=H)
        self.a, self.h = self.batch.a, self.batch.h
        self.stages = self.a.BeatriceBatch_TickStages(self.h)
        self.slots = self.stages + 4
        self.frees = []
        n = self.slots * B * H * 8 * 1024 * 4
        self.d_a, self.d_b = self.dev(n), self.dev(n)        # spare device buffers for the probes
        self.h_in = np.zeros(B * H * 2 * 1024, np.float32)
        self.h_out = np.zeros(B * H * 2 * 1024, np.float32)

    def dev(self, nbytes):
        p = self.hip.malloc(nbytes)
        assert self.hip.lib.hipMemset(p, 0, C.c_size_t(nbytes)) == 0     # (slots no call writes must compare equal between the two batches)
        self.frees.append(p)
        return p

    def close(self):
        self.batch.close()
        self.m.close()
        for p in self.frees:
            self.hip.free(p)


# ---- the entry points, each with arguments that are valid in the mode(s) where it is allowed --------------------------------------------
def _f(c):
    return c.bv.fptr(c.h_in), c.bv.fptr(c.h_out)


PROBES = {
    "ConvertFrames": lambda c: c.a.BeatriceBatch_ConvertFrames(c.h, *_f(c)),
    "ConvertFramesDevice(ptrs)": lambda c: c.a.BeatriceBatch_ConvertFramesDevice(c.h, c.d_a, c.d_b),
    "ConvertFramesDevice(NULL)": lambda c: c.a.BeatriceBatch_ConvertFramesDevice(c.h, None, None),
    "ConvertBlocks48k": lambda c: c.a.BeatriceBatch_ConvertBlocks48k(c.h, *_f(c), CH),
    "ConvertBlocks48kDevice(ptrs)": lambda c: c.a.BeatriceBatch_ConvertBlocks48kDevice(c.h, c.d_a, c.d_b, CH),
    "ConvertBlocks48kDevice(NULL)": lambda c: c.a.BeatriceBatch_ConvertBlocks48kDevice(c.h, None, None, CH),
    "ProcessBlocks": lambda c: c.a.BeatriceBatch_ProcessBlocks(c.h, *_f(c), CH, 441),
    "ProcessBlocksDevice(ptrs)": lambda c: c.a.BeatriceBatch_ProcessBlocksDevice(c.h, c.d_a, c.d_b, CH, 441),
    "ProcessBlocksDevice(NULL)": lambda c: c.a.BeatriceBatch_ProcessBlocksDevice(c.h, None, None, CH, 441),
    "ProcessBlocksRagged": lambda c: c.a.BeatriceBatch_ProcessBlocksRagged(c.h, c.h_in.ctypes.data_as(C.c_void_p), c.h_out.ctypes.data_as(C.c_void_p), CH, (C.c_int * B)(*([441] * B)), 0),
    "ProcessBlocksRaggedDevice": lambda c: c.a.BeatriceBatch_ProcessBlocksRaggedDevice(c.h, (C.c_int * B)(*([441] * B))),
    "StreamFrames": lambda c: c.a.BeatriceBatch_StreamFrames(c.h, *_f(c)),
    "EnableSilentBlockRule(1)": lambda c: c.a.BeatriceBatch_EnableSilentBlockRule(c.h, 1),
    "EnableSilentBlockRule(0)": lambda c: c.a.BeatriceBatch_EnableSilentBlockRule(c.h, 0),
    "SetSilentStreams": lambda c: c.a.BeatriceBatch_SetSilentStreams(c.h, bytes([1] + [0] * (B - 1))),
    "EnablePipelining(2)": lambda c: c.a.BeatriceBatch_EnablePipelining(c.h, 2),
    "EnableTickPipeline(1)": lambda c: c.a.BeatriceBatch_EnableTickPipeline(c.h, 1),
    "EnableTickPipeline(0)": lambda c: c.a.BeatriceBatch_EnableTickPipeline(c.h, 0),
    "EnableHostStreaming(1)": lambda c: c.a.BeatriceBatch_EnableHostStreaming(c.h, 1),
    "BindResidentIO(bind)": lambda c: c.a.BeatriceBatch_BindResidentIO(c.h, c.d_a, c.d_b, c.slots),
    "BindResidentIO(unbind)": lambda c: c.a.BeatriceBatch_BindResidentIO(c.h, None, None, 0),
    "BindResidentIO48k(bind)": lambda c: c.a.BeatriceBatch_BindResidentIO48k(c.h, c.d_a, c.d_b, CH, c.slots),
    "BindResidentBlocks(bind)": lambda c: c.a.BeatriceBatch_BindResidentBlocks(c.h, c.d_a, c.d_b, CH, 441, 8 * c.slots),
    "BindResidentBlocksRagged(bind)": lambda c: c.a.BeatriceBatch_BindResidentBlocksRagged(c.h, c.d_a, c.d_b, CH, 480, c.slots),
    "ConfigureWrapper": lambda c: c.a.BeatriceBatch_ConfigureWrapper(c.h, 44100.0),
    "ConfigureWrapperRates": lambda c: c.a.BeatriceBatch_ConfigureWrapperRates(c.h, (C.c_double * B)(*([44100.0] * B))),
    "ProfileKernels": lambda c: c.a.BeatriceBatch_ProfileKernels(c.h, 1, 4, C.create_string_buffer(4 * 64), c.bv.iptr(np.zeros(4, np.int32)), (C.c_double * 4)(), (C.c_double * 4)(), (C.c_double * 4)()),
    "TimeTickLaunch": lambda c: c.a.BeatriceBatch_TimeTickLaunch(c.h, 1, C.byref(C.c_float(0)), None, None),
    "FlushResidentBlocks": lambda c: c.a.BeatriceBatch_FlushResidentBlocks(c.h),
    "TimeSteps": lambda c: c.a.BeatriceBatch_TimeSteps(c.h, 1, c.bv.fptr(np.zeros(1, np.float32))),
}

# ---- the table of include/beatrice_batch.h: mode -> entry points that refuse with -1 (H: only at that many hops per step) ------------------
_WRAPPERS_IN_ORDER = ["ConvertBlocks48k", "ConvertBlocks48kDevice(ptrs)", "ProcessBlocks", "ProcessBlocksDevice(ptrs)", "ProcessBlocksRagged"]
_NOT_MINE = ["ConvertBlocks48kDevice(NULL)", "ProcessBlocksDevice(NULL)", "ProcessBlocksRaggedDevice", "StreamFrames", "TimeTickLaunch", "FlushResidentBlocks"]
_BINDS = ["BindResidentIO48k(bind)", "BindResidentBlocks(bind)", "BindResidentBlocksRagged(bind)"]
_HOST_AND_PTRS = ["ConvertFrames", "ConvertFramesDevice(ptrs)"]
_OWNS_TICKS = _HOST_AND_PTRS + ["ConvertFramesDevice(NULL)", "TimeSteps", "EnablePipelining(2)", "EnableTickPipeline(1)", "EnableTickPipeline(0)", "BindResidentIO(bind)", "BindResidentIO(unbind)",
                                "ConfigureWrapperRates", "ProfileKernels", "SetSilentStreams"] + _WRAPPERS_IN_ORDER
MATRIX = {
    # nothing bound, in order.  (Allowed here and not probed: ConvertFrames[Device], EnablePipelining, EnableHostStreaming, BindResidentIO[48k], ConfigureWrapper,
    # ProfileKernels, TimeSteps; at one hop per step also the 48 kHz blocks, EnableSilentBlockRule and ConfigureWrapperRates)
    "in_order": {"all": _NOT_MINE + ["ProcessBlocks", "ProcessBlocksDevice(ptrs)", "ProcessBlocksRagged", "SetSilentStreams", "EnableTickPipeline(1)", "BindResidentBlocks(bind)",
                                     "BindResidentBlocksRagged(bind)"],
                 "H>1": ["ConvertBlocks48k", "ConvertBlocks48kDevice(ptrs)", "EnableSilentBlockRule(1)", "ConfigureWrapperRates"]},
    "stage_pipelining": {"all": _NOT_MINE + _WRAPPERS_IN_ORDER + _BINDS + ["SetSilentStreams", "EnableSilentBlockRule(1)", "EnableTickPipeline(1)", "EnableHostStreaming(1)", "ConfigureWrapperRates"]},
    "resident_io": {"all": _NOT_MINE + _WRAPPERS_IN_ORDER + _BINDS + _HOST_AND_PTRS + ["SetSilentStreams", "EnableSilentBlockRule(1)", "EnableHostStreaming(1)", "ConfigureWrapperRates"]},
    "tick": {"all": [p for p in _NOT_MINE if p != "TimeTickLaunch"] + _WRAPPERS_IN_ORDER + _BINDS + _HOST_AND_PTRS + ["SetSilentStreams", "EnablePipelining(2)", "EnableHostStreaming(1)", "BindResidentIO(bind)", "BindResidentIO(unbind)",
                                                                                    "ConfigureWrapperRates", "ProfileKernels"]},
    "host_streaming": {"all": [p for p in _NOT_MINE if p != "StreamFrames"] + _BINDS + _OWNS_TICKS + ["EnableSilentBlockRule(1)"]},
    "blocks48k_around_ticks": {"all": [p for p in _NOT_MINE if p != "ConvertBlocks48kDevice(NULL)"] + ["BindResidentBlocks(bind)", "BindResidentBlocksRagged(bind)", "EnableHostStreaming(1)"] + _OWNS_TICKS},
    "resident_blocks": {"all": [p for p in _NOT_MINE if p not in ("ProcessBlocksDevice(NULL)", "FlushResidentBlocks")] + ["BindResidentIO48k(bind)", "EnableHostStreaming(1)", "EnableSilentBlockRule(1)", "EnableSilentBlockRule(0)",
                                                                                                                      "ConfigureWrapper"] + _OWNS_TICKS},
    "resident_blocks_per_stream_clocks": {"all": [p for p in _NOT_MINE if p not in ("ProcessBlocksRaggedDevice", "FlushResidentBlocks")] + ["BindResidentIO48k(bind)", "EnableHostStreaming(1)", "EnableSilentBlockRule(1)",
                                                                                                                                        "EnableSilentBlockRule(0)", "ConfigureWrapper"] + _OWNS_TICKS},
    # the in-order silent-block rule of the 48 kHz blocks (one hop per step)
    "silent_rule_in_order": {"all": _NOT_MINE + _BINDS + ["EnableHostStreaming(1)", "EnablePipelining(2)", "EnableTickPipeline(1)", "BindResidentIO(bind)", "ProcessBlocks", "ProcessBlocksDevice(ptrs)", "ProcessBlocksRagged"]},
}
MODES_AT = [("in_order", 1), ("in_order", 2), ("in_order", 4), ("stage_pipelining", 1), ("stage_pipelining", 2), ("resident_io", 1), ("resident_io", 2), ("resident_io", 4),
            ("tick", 1), ("tick", 2), ("tick", 4), ("host_streaming", 1), ("host_streaming", 2), ("host_streaming", 4), ("blocks48k_around_ticks", 1),
            ("blocks48k_around_ticks", 2), ("blocks48k_around_ticks", 4), ("resident_blocks", 1), ("resident_blocks", 2), ("resident_blocks", 4),
            ("resident_blocks_per_stream_clocks", 1), ("silent_rule_in_order", 1)]


def refused(mode, H):
    m = MATRIX[mode]
    return list(dict.fromkeys(m["all"] + (m.get("H>1", []) if H > 1 else [])))


# ---- how a batch enters each mode and runs one step in it; returns the step's result (or None) -----------------------------------------------
def enter(c, mode):
    a, h, H, hip = c.a, c.h, c.H, c.hip
    if mode == "stage_pipelining":
        assert a.BeatriceBatch_EnablePipelining(h, 2) == 0
    if mode in ("resident_io", "tick"):
        c.d_in, c.d_out = c.dev(c.slots * B * H * 160 * 4), c.dev(c.slots * B * H * 240 * 4)
        assert a.BeatriceBatch_BindResidentIO(h, c.d_in, c.d_out, c.slots) == 0
        if mode == "tick":
            assert a.BeatriceBatch_EnableTickPipeline(h, 1) == 0
    if mode == "host_streaming":
        assert a.BeatriceBatch_EnableHostStreaming(h, 1) == 0
    if mode == "blocks48k_around_ticks":
        c.d_in, c.d_out = c.dev(c.slots * B * H * CH * 480 * 4), c.dev(c.slots * B * H * CH * 480 * 4)
        assert a.BeatriceBatch_BindResidentIO48k(h, c.d_in, c.d_out, CH, c.slots) == 0
    if mode == "resident_blocks":
        assert a.BeatriceBatch_ConfigureWrapper(h, 44100.0) == 0
        c.rb_slots = a.BeatriceBatch_ResidentBlocksDelayFor(h, 441) + 2 + 6
        c.d_in, c.d_out = c.dev(c.rb_slots * B * CH * 441 * 4), c.dev(c.rb_slots * B * CH * 441 * 4)
        assert a.BeatriceBatch_BindResidentBlocks(h, c.d_in, c.d_out, CH, 441, c.rb_slots) == 0
    if mode == "resident_blocks_per_stream_clocks":
        c.rates = [44100.0, 48000.0, 32000.0][:B]
        c.ns = [441, 480, 320][:B]
        assert a.BeatriceBatch_ConfigureWrapperRates(h, (C.c_double * B)(*c.rates)) == 0
        c.rb_slots = c.stages + 6
        c.d_in, c.d_out = c.dev(c.rb_slots * B * CH * 480 * 4), c.dev(c.rb_slots * B * CH * 480 * 4)
        assert a.BeatriceBatch_BindResidentBlocksRagged(h, c.d_in, c.d_out, CH, 480, c.rb_slots) == 0
    if mode == "silent_rule_in_order":
        assert a.BeatriceBatch_EnableSilentBlockRule(h, 1) == 0


def step(c, mode, k, x16, x48):
    """x16: [B][H*160] model-rate input of step k, x48: [B][H][CH][480] host-rate input of step k"""
    a, h, H, hip = c.a, c.h, c.H, c.hip
    if mode in ("in_order", "stage_pipelining"):
        return c.batch.convert(x16)
    if mode in ("resident_io", "tick"):
        buf = np.zeros((c.slots, B, H * 160), np.float32)
        buf[k % c.slots] = x16
        hip.h2d(c.d_in, buf)
        assert a.BeatriceBatch_ConvertFramesDevice(h, None, None) == 0
        assert a.BeatriceBatch_Synchronize(h) == 0
        out = np.zeros((c.slots, B, H * 240), np.float32)
        hip.d2h(out, c.d_out)
        return out[k % c.slots].copy()
    if mode == "host_streaming":
        out = np.zeros((B, H * 240), np.float32)
        rc = a.BeatriceBatch_StreamFrames(h, c.bv.fptr(np.ascontiguousarray(x16)), c.bv.fptr(out))
        assert rc in (0, 1)
        return out if rc == 1 else None
    if mode == "blocks48k_around_ticks":
        buf = np.zeros((c.slots, B, H, CH, 480), np.float32)
        buf[k % c.slots] = x48
        hip.h2d(c.d_in, buf)
        assert a.BeatriceBatch_ConvertBlocks48kDevice(h, None, None, CH) == 0
        assert a.BeatriceBatch_Synchronize(h) == 0
        out = np.zeros_like(buf)
        hip.d2h(out, c.d_out)
        return out[k % c.slots].copy()
    if mode == "silent_rule_in_order":
        y = np.zeros((B, CH, 480), np.float32)
        assert a.BeatriceBatch_ConvertBlocks48k(h, c.bv.fptr(np.ascontiguousarray(x48[:, 0])), c.bv.fptr(y), CH) == 0
        return y
    if mode == "resident_blocks":   # 441-sample blocks at 44.1 kHz: the first 441 samples of every stream's 48 kHz material will do
        blk = np.ascontiguousarray(x48[:, 0, :, :441])   # (only the call's own slot is written: earlier calls are still in flight)
        hip.h2d(C.c_void_p(c.d_in.value + (k % c.rb_slots) * blk.nbytes), blk)
        assert a.BeatriceBatch_ProcessBlocksDevice(h, None, None, CH, 441) == 0
        return None
    if mode == "resident_blocks_per_stream_clocks":
        blk = np.zeros((B, CH * 480), np.float32)
        for s in range(B):
            blk[s, :CH * c.ns[s]] = x48[s, 0, :, :c.ns[s]].reshape(-1)
        hip.h2d(C.c_void_p(c.d_in.value + (k % c.rb_slots) * blk.nbytes), blk)
        assert a.BeatriceBatch_ProcessBlocksRaggedDevice(h, (C.c_int * B)(*c.ns)) == 0
        return None
    raise AssertionError(mode)


def finish(c, mode):
    """what is still inside the mode when the steps are over"""
    a, h, H, hip = c.a, c.h, c.H, c.hip
    outs = []
    if mode == "host_streaming":
        while True:
            out = np.zeros((B, H * 240), np.float32)
            if a.BeatriceBatch_StreamFlush(h, c.bv.fptr(out)) != 1:
                break
            outs.append(out)
    if mode.startswith("resident_blocks"):
        assert a.BeatriceBatch_Synchronize(h) == 0
        n = c.rb_slots * B * CH * (441 if mode == "resident_blocks" else 480)
        out = np.zeros(n, np.float32)
        hip.d2h(out, c.d_out)
        outs.append(out)
    return outs


@pytest.mark.parametrize("mode,H", MODES_AT)
def test_refused_cells_refuse_and_leave_the_batch_untouched(bv, product, model_dir, mode, H):
    product = bv.bind_batch(product)
    steps = 7 if not mode.startswith("resident_blocks") else (10 if H == 1 else 44)
    x16 = np.stack([bv.synth_audio(160 * H * steps, seed=8800 + s) for s in range(B)]).reshape(B, steps, H * 160)
    x48 = np.stack([wrapperlib.test_signal(480 * H * steps * CH, 48000, seed=8900 + s) for s in range(B)]).astype(np.float32).reshape(B, steps, H, CH, 480)
    names = refused(mode, H)
    assert names and all(n in PROBES for n in names)
    results = []
    for tries in (True, False):
        c = Ctx(bv, product, model_dir, H)
        try:
            enter(c, mode)
            got = []
            for k in range(steps):
                if tries and k in (1, 4):        # (with steps in flight, and again later)
                    wrong = [(n, rc) for n in names for rc in [PROBES[n](c)] if rc != -1]
                    assert not wrong, "mode %s at %d hop(s) per step: expected -1 from %s" % (mode, H, wrong)
                y = step(c, mode, k, x16[:, k], x48[:, k])
                if y is not None:
                    got.append(np.array(y, copy=True))
            got += finish(c, mode)
            results.append(got)
        finally:
            c.close()
    tried, control = results
    assert len(tried) == len(control) and len(tried) > 0
    assert max(float(np.abs(y).max()) for y in control) > 1e-3, "the mode's own entry point produced silence"
    for i, (p, q) in enumerate(zip(tried, control)):
        assert np.array_equal(p, q), "mode %s, H = %d: result %d differs after the refused calls (max-abs %g)" % (mode, H, i, float(np.abs(p - q).max()))
