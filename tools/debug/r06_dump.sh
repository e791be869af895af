M=$(python - <<'PY'
import sys, tempfile, os
sys.path.insert(0, "tools")
import make_model
d = tempfile.mkdtemp(); make_model.make_model(d, n_speakers=2); print(d)
PY
)
examples/latency_b1 $M 20000 2000 --histogram --dump /tmp/dump.txt > /dev/null
python - <<'PY'
import numpy as np
a = np.loadtxt("/tmp/dump.txt")
hop = a[:, 0]
print("p50 %.1f p90 %.1f p99 %.1f" % tuple(np.percentile(hop, [50, 90, 99])))
slow = np.where(hop > np.percentile(hop, 50) + 20)[0]
print("hops more than 20 us over the median: %d of %d" % (len(slow), len(hop)))
print("gaps between them:", np.bincount(np.diff(slow))[:80].nonzero()[0][:40], "counts", np.bincount(np.diff(slow))[np.bincount(np.diff(slow)).nonzero()[0][:40]])
for name, col in (("phone", 1), ("pitch", 2), ("wave", 3)):
    med = np.median(a[:, col])
    print(name, "median %.1f; in the slow hops: median %.1f, mean excess %.1f" % (med, np.median(a[slow, col]), (a[slow, col] - med).mean()))
print("first slow hops:", slow[:30])
for i in slow[:8]:
    print(i, a[i])
PY
