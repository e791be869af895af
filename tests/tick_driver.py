"""Drives the product's tick pipeline (BeatriceBatch_EnableTickPipeline over BeatriceBatch_BindResidentIO) from host arrays:
shared by the throughput-mode parity tests and __graft_entry__.smoke().  Test plumbing only (ctypes + libamdhip64 copies)."""
import ctypes as C

import numpy as np


class Hip:
    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so")  # the runtime the product library is linked against

    def malloc(self, nbytes):
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
        return p

    def h2d(self, dst, arr):
        assert self.lib.hipMemcpy(dst, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), 1) == 0

    def d2h(self, arr, src):
        assert self.lib.hipMemcpy(arr.ctypes.data_as(C.c_void_p), src, C.c_size_t(arr.nbytes), 2) == 0

    def free(self, p):
        self.lib.hipFree(p)


def run_tick(bv, batch, steps, hop_input, change=None, slots=None, chunk=None, leave=True):
    """Feeds `steps` steps through tick mode and returns their samples [steps][B][H * 240] (H = the batch's hops per step).

    hop_input(k) -> [B][H * 160] is step k's input; change(batch, k) runs before step k is fed (settings travel with the step).
    The resident I/O has `slots` slots (default: stages + 6) used round-robin as the library does (step k <-> slot k mod
    slots); steps are fed `chunk` (<= slots) at a time without waiting, then the pipeline is drained and the chunk read back,
    so the ring wraps many times over a long run."""
    hip = Hip()
    a, h, B, H = batch.a, batch.h, batch.B, batch.H
    stages = a.BeatriceBatch_TickStages(h)
    slots = slots or stages + 6
    chunk = min(chunk or slots, slots)
    d_in, d_out = hip.malloc(slots * B * H * 160 * 4), hip.malloc(slots * B * H * 240 * 4)
    try:
        assert a.BeatriceBatch_BindResidentIO(h, d_in, d_out, slots) == 0
        assert a.BeatriceBatch_EnableTickPipeline(h, 1) == 0
        got = np.zeros((steps, B, H * 240), np.float32)
        buf = np.zeros((slots, B, H * 160), np.float32)
        k0 = 0
        while k0 < steps:
            n = min(chunk, steps - k0)
            for k in range(k0, k0 + n):
                buf[k % slots] = hop_input(k)
            hip.h2d(d_in, buf)
            for k in range(k0, k0 + n):
                if change is not None:
                    change(batch, k)
                assert a.BeatriceBatch_ConvertFramesDevice(h, None, None) == 0
            assert a.BeatriceBatch_Synchronize(h) == 0
            out = np.zeros((slots, B, H * 240), np.float32)
            hip.d2h(out, d_out)
            for k in range(k0, k0 + n):
                got[k] = out[k % slots]
            k0 += n
        if leave:   # (streams that have sat steps out come back to the batch's step counter at every drained point)
            assert a.BeatriceBatch_EnableTickPipeline(h, 0) == 0
            assert a.BeatriceBatch_BindResidentIO(h, None, None, 0) == 0
    finally:
        hip.free(d_in)
        hip.free(d_out)
    return got
