"""Summarise a BEATRICE_HIP_TICK_TRACE dump: per body type, workgroup count and duration; makespan; slot utilisation."""
import sys
import numpy as np
NAMES = "f1 fft f2 f3 f4 f5 p1 rb p23 pout head out cond inp up1 res1a res1b up2 qgru pgru vq tail tail1 tail2 tail3 blkA1 blkA2 blkA4 blkA8 blkB blkBq f4s f5s rbs p1s up1s tail1s tail2s qgru1 pgru1 qgrum pgrum f2l f3l p23l poutl outl inpl up1l res1al res1bl up2l".split()
d = np.loadtxt(sys.argv[1], dtype=np.uint64).astype(np.int64)
d = d[d[:, 1] > 0]
d = d[np.abs(d[:, 0] - np.median(d[:, 0])) < 100000]   # (entries of workgroups that left at once keep an older tick's stamps: beyond 1 ms of the median start)
t0 = d[:, 0].min()
start, end, typ = (d[:, 0] - t0) / 100.0, (d[:, 1] - t0) / 100.0, d[:, 2] & 255   # 100 MHz -> us
cyc = d[:, 2] >> 8
long_ = (end - start) > 5.0
if long_.any() and cyc[long_].sum() > 0:
    print("shader clock over the workgroups longer than 5 us: %.0f MHz (shader cycles / wall time)" % (cyc[long_].sum() / (end - start)[long_].sum()))
print("workgroups %d, makespan %.1f us, sum of workgroup time %.0f us -> %.1f workgroups resident on average" %
      (len(d), end.max(), (end - start).sum(), (end - start).sum() / end.max()))
for t in sorted(set(typ.tolist())):
    m = typ == t
    name = NAMES[t] if 0 <= t < len(NAMES) else "idle"
    print("  %-6s n %4d  dur mean %6.1f max %6.1f  first start %6.1f  last end %6.1f" % (name, m.sum(), (end - start)[m].mean(), (end - start)[m].max(), start[m].min(), end[m].max()))
# residency over time: workgroups resident per 5 us bin, split heavy (mean duration >= 25 us) / other
dur = end - start
heavy_types = {t for t in set(typ.tolist()) if dur[typ == t].mean() >= 25.0}
edges = np.arange(0, end.max() + 5, 5.0)
print("  t(us)   resident  heavy  other")
for lo in edges[:-1]:
    mid = lo + 2.5
    live = (start <= mid) & (end > mid)
    h = sum(int((live & (typ == t)).sum()) for t in heavy_types)
    print("  %5.1f   %6d   %5d  %5d" % (mid, live.sum(), h, live.sum() - h))
