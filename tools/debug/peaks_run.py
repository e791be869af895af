import ctypes
lib = ctypes.CDLL("tools/microbench/libpeaks.so")
lib.peaks_hbm_copy_gbs.restype = ctypes.c_double; lib.peaks_hbm_copy_gbs.argtypes = [ctypes.c_size_t, ctypes.c_int]
lib.peaks_mfma_f32_tflops.restype = ctypes.c_double; lib.peaks_mfma_f32_tflops.argtypes = [ctypes.c_int, ctypes.c_int]
print("copy GB/s", lib.peaks_hbm_copy_gbs(1 << 30, 5), "mfma TF", lib.peaks_mfma_f32_tflops(20000, 3))
