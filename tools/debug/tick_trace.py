"""Summarise a BEATRICE_HIP_TICK_TRACE dump: per body type, workgroup count and duration; makespan; slot utilisation."""
import sys
import numpy as np
NAMES = "f1 fft f2 f3 f4 f5 p1 rb p23 pout head out cond inp up1 res1a res1b up2 qgru pgru vq tail blkA1 blkA2 blkA4 blkA8 blkB".split()
d = np.loadtxt(sys.argv[1], dtype=np.int64)
d = d[d[:, 1] > 0]
t0 = d[:, 0].min()
start, end, typ = (d[:, 0] - t0) / 100.0, (d[:, 1] - t0) / 100.0, d[:, 2]   # 100 MHz -> us
print("workgroups %d, makespan %.1f us, sum of workgroup time %.0f us -> %.1f workgroups resident on average" %
      (len(d), end.max(), (end - start).sum(), (end - start).sum() / end.max()))
for t in sorted(set(typ.tolist())):
    m = typ == t
    name = NAMES[t] if 0 <= t < len(NAMES) else "idle"
    print("  %-6s n %4d  dur mean %6.1f max %6.1f  first start %6.1f  last end %6.1f" % (name, m.sum(), (end - start)[m].mean(), (end - start)[m].max(), start[m].min(), end[m].max()))
