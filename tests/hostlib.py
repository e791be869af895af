"""ctypes view of the C++ host layer (beatrice-vst_amd/host/processor_core.cc)."""
import ctypes as C
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_PRODUCT = os.path.join(REPO, "beatrice-vst_amd", "host", "libbeatrice_host.so")   # linked against the HIP library
HOST_ON_ORACLE = os.path.join(REPO, "oracle", "libhost_on_oracle.so")                   # same source, oracle core (test only)
_f32p = C.POINTER(C.c_float)


class Host:
    def __init__(self, path, sample_rate, pitch_trace=4096):
        self.lib = C.CDLL(path)
        L = self.lib
        L.BeatriceHost_Create.restype, L.BeatriceHost_Create.argtypes = C.c_void_p, [C.c_double]
        L.BeatriceHost_Destroy.argtypes = [C.c_void_p]
        L.BeatriceHost_LoadModel.argtypes = [C.c_void_p, C.c_char_p]
        L.BeatriceHost_Process.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int]
        L.BeatriceHost_ResetContext.argtypes = [C.c_void_p]
        L.BeatriceHost_NumSpeakers.argtypes = [C.c_void_p]
        L.BeatriceHost_TakePitchTrace.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.BeatriceHost_EnablePitchTrace.restype, L.BeatriceHost_EnablePitchTrace.argtypes = None, [C.c_void_p, C.c_int]
        L.BeatriceHost_ReserveBlocks.argtypes = [C.c_void_p, C.c_int]
        L.BeatriceHost_BufferFingerprint.restype, L.BeatriceHost_BufferFingerprint.argtypes = C.c_ulonglong, [C.c_void_p]
        for name in ("SetSampleRate", "SetFormantShift", "SetPitchShift", "SetInputGain", "SetOutputGain", "SetAverageSourcePitch",
                     "SetIntonationIntensity", "SetPitchCorrection", "SetMinSourcePitch", "SetMaxSourcePitch"):
            getattr(L, "BeatriceHost_" + name).argtypes = [C.c_void_p, C.c_double]
        for name in ("SetTargetSpeaker", "SetPitchCorrectionType", "SetVQNumNeighbors"):
            getattr(L, "BeatriceHost_" + name).argtypes = [C.c_void_p, C.c_int]
        self.h = L.BeatriceHost_Create(float(sample_rate))
        if pitch_trace:   # (the library's default is off: a ring of this capacity, allocated here, off the audio path)
            L.BeatriceHost_EnablePitchTrace(self.h, pitch_trace)

    def call(self, name, *args):
        return getattr(self.lib, "BeatriceHost_" + name)(self.h, *args)

    def load(self, model_dir):
        return self.call("LoadModel", os.path.join(model_dir, "model.toml").encode())

    def process(self, x, block):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros_like(x)
        codes = []
        for pos in range(0, len(x), block):
            n = min(block, len(x) - pos)
            codes.append(self.call("Process", x[pos:pos + n].ctypes.data_as(_f32p), out[pos:pos + n].ctypes.data_as(_f32p), n))
        return out, codes

    def pitch_trace(self):
        buf = (C.c_int * 4096)()
        n = self.call("TakePitchTrace", buf, 4096)
        return list(buf[:min(n, 4096)])

    def close(self):
        if self.h:
            self.lib.BeatriceHost_Destroy(self.h)
            self.h = None
