"""Per-body-type runs of the tick launch (MEASUREMENT BUILD: build_variants/lib_meas.so, BeatriceBatchMeas_TickOnlyTypes).

For every group of body types: the table holds ONLY that group's bodies, the pipeline is refilled, then N_MEAS full ticks run.
Under `rocprofv3 --pmc ... --kernel-trace` the table-kernel dispatches of a group are a fixed-size slice of the dispatch list
(tools/debug/tick_inst_by_body.sh cuts them); without a profiler the script prints the group's own launch time.
  python tools/debug/tick_inst_by_body.py [streams] [hops per step]
"""
import ctypes, importlib, os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
torch.cuda.init()
import make_model
bv = importlib.import_module("beatrice-vst_amd")

# tick::BodyType (csrc/tick.hip.h), in order
TYPES = ("F1 FFT F2 F3 F4 F5 P1 RB P23 POUT HEAD OUT COND INP UP1 RES1A RES1B UP2 QGRU PGRU VQ TAIL TAIL1 TAIL2 TAIL3 "
         "BLKA1 BLKA2 BLKA4 BLKA8 BLKB BLKBQ F4S F5S RBS P1S UP1S TAIL1S TAIL2S QGRU1 PGRU1 QGRUM PGRUM").split()
GROUPS = [("all", None), ("f1", ["F1"]), ("fft", ["FFT"]), ("f2", ["F2"]), ("f3", ["F3"]), ("f4", ["F4"]), ("f5", ["F5"]), ("p1", ["P1"]),
          ("rb x4", ["RB"]), ("p23 x2", ["P23"]), ("pout", ["POUT"]), ("head", ["HEAD"]), ("out", ["OUT"]), ("cond", ["COND"]), ("inp", ["INP"]),
          ("up1", ["UP1"]), ("res1a", ["RES1A"]), ("res1b", ["RES1B"]), ("up2", ["UP2"]), ("qgru", ["QGRU", "QGRU1", "QGRUM"]), ("pgru", ["PGRU", "PGRU1", "PGRUM"]),
          ("tail1", ["TAIL1"]), ("tail2", ["TAIL2"]), ("tail3", ["TAIL3"]), ("blk.a x4", ["BLKA1", "BLKA2", "BLKA4", "BLKA8"]), ("blk.b x4", ["BLKB", "BLKBQ"]),
          ("all again", None)]
N_FILL, N_MEAS = 30, 8


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    product = bv.bind_batch(bv.load_product())
    meas = product.lib.BeatriceBatchMeas_TickOnlyTypes   # AttributeError: not a measurement build
    meas.restype, meas.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_ulonglong]
    tmp = tempfile.TemporaryDirectory(); make_model.make_model(tmp.name, n_speakers=1)
    m = bv.Models(product, tmp.name)
    n = 64
    batch = bv.Batch(m, B, hops_per_step=H)
    d_in = torch.randn((n, B, H * 160), device="cuda") * 0.1
    d_out = torch.zeros((n, B, H * 240), device="cuda")
    assert product.BeatriceBatch_BindResidentIO(batch.h, d_in.data_ptr(), d_out.data_ptr(), n) == 0
    assert product.BeatriceBatch_EnableTickPipeline(batch.h, 1) == 0
    us, fl, by = ctypes.c_float(0), ctypes.c_double(0), ctypes.c_double(0)
    print("groups: %d  fill %d  meas %d  streams %d  hops %d" % (len(GROUPS), N_FILL, N_MEAS, B, H))
    for name, types in GROUPS:
        mask = (1 << 64) - 1 if types is None else sum(1 << TYPES.index(t) for t in types)
        assert meas(batch.h, mask) == 0
        for _ in range(N_FILL):
            product.BeatriceBatch_ConvertFramesDevice(batch.h, None, None)
        assert product.BeatriceBatch_TimeTickLaunch(batch.h, N_MEAS, ctypes.byref(us), ctypes.byref(fl), ctypes.byref(by)) == 0
        print("group %-10s  %8.2f us per launch alone" % (name, us.value), flush=True)
    product.BeatriceBatch_Synchronize(batch.h)


if __name__ == "__main__":
    main()
