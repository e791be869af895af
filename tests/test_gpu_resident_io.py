"""Resident I/O (BeatriceBatch_BindResidentIO): steps that read and write slots of device buffers without a
per-step copy give the same samples as the host-buffer calls, for single hops and in block mode, across the
wrap-around of the slot index."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("B,H,slots,steps,pipelined", [(7, 1, 5, 13, 0), (4, 4, 3, 7, 0), (37, 1, 6, 18, 2), (5, 2, 4, 11, 3), (19, 1, 7, 23, 3), (40, 1, 9, 30, 4)])
def test_resident_io_matches_host_buffers(bv, product, model_dir, B, H, slots, steps, pipelined):
    """pipelined = 2..4: BeatriceBatch_EnablePipelining with that many stages -- stage s of step t+1 overlaps stage s+1
    of step t on separate streams while steps are enqueued without waiting; settings keep changing in between."""
    hip = Hip()
    audio = np.stack([bv.synth_audio(160 * H * steps, seed=40 + s) for s in range(B)])  # [B][steps*H*160]
    m = bv.Models(product, model_dir)

    def settings(batch):
        for s in range(B):
            batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, s, s % 3)
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, s % 3)
        batch.a.BeatriceBatch_FlushSpeaker(batch.h, -1)

    def change(batch, k):  # settings that change while earlier steps may still be in flight
        if k % 3 == 1:
            s = (5 * k) % B
            batch.a.BeatriceBatch_SetTargetSpeaker(batch.h, s, (k + s) % 3)
            batch.a.BeatriceBatch_SetFormantShift(batch.h, (s + 1) % B, float(k % 5) - 2.0)
            batch.a.BeatriceBatch_SetVQNumNeighbors(batch.h, (s + 2) % B, k % 4)
        if k == 5:
            assert batch.a.BeatriceBatch_ResetStream(batch.h, 1) == 0   # fresh state for one stream, mid-flight

    # reference: the synchronous host-buffer entry point
    ref_batch = bv.Batch(m, B, hops_per_step=H)
    settings(ref_batch)
    ref = []
    for k in range(steps):
        change(ref_batch, k)
        ref.append(ref_batch.convert(audio[:, k * H * 160:(k + 1) * H * 160]))
    ref = np.stack(ref)  # [steps][B][H*240]
    ref_batch.close()

    batch = bv.Batch(m, B, hops_per_step=H)
    settings(batch)
    a, h = batch.a, batch.h
    d_in, d_out = hip.malloc(slots * B * H * 160 * 4), hip.malloc(slots * B * H * 240 * 4)
    assert a.BeatriceBatch_BindResidentIO(h, d_in, d_out, slots) == 0
    assert a.BeatriceBatch_EnablePipelining(h, pipelined) == 0
    x0 = np.zeros((B, H * 160), np.float32)
    assert a.BeatriceBatch_ConvertFrames(h, bv.fptr(x0), bv.fptr(np.zeros((B, H * 240), np.float32))) == -1  # bound
    got = np.zeros_like(ref)
    for first in range(0, steps, slots):      # fill the slots, run them, read them back; then wrap around
        n = min(slots, steps - first)
        buf = np.zeros((slots, B, H * 160), np.float32)
        for k in range(n):
            buf[(first + k) % slots] = audio[:, (first + k) * H * 160:(first + k + 1) * H * 160]
        hip.h2d(d_in, buf)
        for k in range(n):
            change(batch, first + k)
            assert a.BeatriceBatch_ConvertFramesDevice(h, None, None) == 0
        assert a.BeatriceBatch_Synchronize(h) == 0
        out = np.zeros((slots, B, H * 240), np.float32)
        hip.d2h(out, d_out)
        for k in range(n):
            got[first + k] = out[(first + k) % slots]
    # unbinding returns to the copying entry points
    assert a.BeatriceBatch_BindResidentIO(h, None, None, 0) == 0
    y = batch.convert(np.zeros((B, H * 160), np.float32))
    assert y.shape == (B, H * 240)
    batch.close()
    m.close()
    hip.free(d_in); hip.free(d_out)
    print("resident I/O B=%d H=%d slots=%d pipelined=%d: %s" % (B, H, slots, pipelined, "bit-identical" if np.array_equal(ref, got) else
                                                   "max-abs %g" % np.abs(ref - got).max()))
    assert np.abs(got).max() > 0.05
    assert np.array_equal(ref, got)
