"""ctypes view of the two wrapper libraries: the oracle's C restatement (prefix wo_) and the driver
built from the reference's own headers (prefix ref_, oracle/_ref/libref_wrapper.so)."""
import ctypes as C
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_WRAPPER = os.path.join(REPO, "oracle", "libwrapper_oracle.so")
REF_WRAPPER = os.path.join(REPO, "oracle", "_ref", "libref_wrapper.so")
HOP_FN = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p)
_f32p = C.POINTER(C.c_float)


def stub_hop(in160, out240, _user):
    """Deterministic stand-in for the model hop: out[i] = in[(2 i) // 3]."""
    for i in range(240):
        out240[i] = in160[(2 * i) // 3]


class Wrapper:
    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        p = prefix
        self.f_fraction = getattr(self.lib, p + "fraction")
        self.f_fraction.argtypes = [C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self.f_create = getattr(self.lib, p + "create")
        self.f_create.restype, self.f_create.argtypes = C.c_void_p, [C.c_double, HOP_FN, C.c_void_p]
        self.f_destroy = getattr(self.lib, p + "destroy")
        self.f_destroy.argtypes = [C.c_void_p]
        self.f_process = getattr(self.lib, p + "process")
        self.f_process.restype, self.f_process.argtypes = C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int]
        self.f_in_gain = getattr(self.lib, p + "set_input_gain")
        self.f_in_gain.argtypes = [C.c_void_p, C.c_double]
        self.f_out_gain = getattr(self.lib, p + "set_output_gain")
        self.f_out_gain.argtypes = [C.c_void_p, C.c_double]
        self.g_create = getattr(self.lib, p + "gain_create")
        self.g_create.restype, self.g_create.argtypes = C.c_void_p, [C.c_double, C.c_double]
        self.g_set = getattr(self.lib, p + "gain_set_target")
        self.g_set.argtypes = [C.c_void_p, C.c_double]
        self.g_process = getattr(self.lib, p + "gain_process")
        self.g_process.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int]
        self.g_destroy = getattr(self.lib, p + "gain_destroy")
        self.g_destroy.argtypes = [C.c_void_p]

    def fraction(self, ratio):
        n, d = C.c_int(0), C.c_int(0)
        self.f_fraction(ratio, C.byref(n), C.byref(d))
        return n.value, d.value

    def run_chain(self, sample_rate, x, block, hop=stub_hop, in_gain_events=(), out_gain_events=()):
        """Stream x through Process() in blocks of `block` samples; gain events = [(sample_index, dB)]."""
        cb = HOP_FN(hop)
        p = self.f_create(float(sample_rate), cb, None)
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros_like(x)
        pos = 0
        ev_in, ev_out = list(in_gain_events), list(out_gain_events)
        while pos < len(x):
            while ev_in and ev_in[0][0] <= pos:
                self.f_in_gain(p, ev_in.pop(0)[1])
            while ev_out and ev_out[0][0] <= pos:
                self.f_out_gain(p, ev_out.pop(0)[1])
            n = min(block, len(x) - pos)
            rc = self.f_process(p, x[pos:pos + n].ctypes.data_as(_f32p), out[pos:pos + n].ctypes.data_as(_f32p), n)
            assert rc == 0
            pos += n
        self.f_destroy(p)
        return out

    def gain_trace(self, sample_rate, x, events, block=97):
        g = self.g_create(float(sample_rate), 0.0)
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros_like(x)
        pos, ev = 0, list(events)
        while pos < len(x):
            while ev and ev[0][0] <= pos:
                self.g_set(g, ev.pop(0)[1])
            n = min(block, len(x) - pos)
            self.g_process(g, x[pos:pos + n].ctypes.data_as(_f32p), out[pos:pos + n].ctypes.data_as(_f32p), n)
            pos += n
        self.g_destroy(g)
        return out


def oracle_wrapper():
    return Wrapper(ORACLE_WRAPPER, "wo_")


def ref_wrapper():
    return Wrapper(REF_WRAPPER, "ref_") if os.path.exists(REF_WRAPPER) else None


def test_signal(n, sample_rate, seed):
    """Chirp + white noise, amplitude 0.5, from the portable counter-based generator."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from make_model import Stream
    noise = Stream(0xA0D10 + seed).uniform(n).astype(np.float64)
    t = np.arange(n) / float(sample_rate)
    f0, f1 = 60.0, min(9000.0, 0.45 * sample_rate)
    phase = 2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t / max(t[-1], 1e-9))
    return (0.25 * np.sin(phase) + 0.25 * noise).astype(np.float32)
