#!/bin/bash
# The CPU-side native code (oracle, C++ host layer: processor core, legacy core, proxy, TOML reader, state blob) under the
# sanitizers, SURVEY.md section 5:
#   tools/sanitize_host.sh asan   -> -fsanitize=address,undefined  over the host / oracle / wrapper CPU tests
#   tools/sanitize_host.sh tsan   -> -fsanitize=thread             over the proxy / state tests + a two-thread stress of the proxy
# Builds in a scratch copy of the repo (the in-tree .so files are not touched); python runs with the sanitizer runtime preloaded.
set -e
MODE=${1:-asan}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WORK=$(mktemp -d /tmp/beatrice_san.XXXX)
trap 'rm -rf "$WORK"' EXIT
mkdir -p "$WORK/repo"
( cd "$ROOT" && tar --exclude=.git --exclude=gpurun_out --exclude=build_variants -cf - . ) | tar -xf - -C "$WORK/repo"
cd "$WORK/repo"
rm -f oracle/*.so   # (the product library stays as built: the `built` fixture must find it; the CPU tests never call into it)
GCCLIB=$(dirname "$(gcc -print-file-name=libasan.so)")
if [ "$MODE" = asan ]; then
  SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
  PRELOAD="$GCCLIB/libasan.so:$GCCLIB/libubsan.so"
  export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1   # (CPython itself leaks by design; leaks of the host objects are covered by the lifecycle tests)
  export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
  TESTS="tests/test_host_layer.py tests/test_host_proxy.py tests/test_host_legacy.py tests/test_morph.py tests/test_wrapper_oracle.py tests/test_cpu_oracle_and_abi.py tests/test_cpu_legacy_generations.py"
else
  SAN="-fsanitize=thread -fno-omit-frame-pointer -g"
  PRELOAD="$GCCLIB/libtsan.so"
  export TSAN_OPTIONS=halt_on_error=1:report_signal_unsafe=0
  TESTS="tests/test_host_proxy.py"   # (tests that call back from C into Python through ctypes do not finish under TSAN)
fi
make -s -C oracle CC="gcc $SAN" CXX="g++ $SAN" libbeatrice_oracle.so libhost_on_oracle.so libwrapper_oracle.so liboracle_bench.so
# (the product library cannot be built here for the sanitizers' sake: tests that need it are the -m gpu ones)
echo "== $MODE: pytest $TESTS"
LD_PRELOAD=$PRELOAD timeout 600 python -m pytest $TESTS -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
if [ "$MODE" = tsan ]; then
  echo "== tsan: two threads driving two proxies (and one shared read-only model directory) at once"
  LD_PRELOAD=$PRELOAD timeout 300 python - <<'PY'
import ctypes as C, os, sys, threading
sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import numpy as np
import hostlib, make_model
from test_host_proxy import Proxy
d = "/tmp/beatrice_san_model_%d" % os.getpid()
os.makedirs(d, exist_ok=True)
make_model.make_model(d, n_speakers=2)
def work(seed):
    p = Proxy(48000.0)
    assert p.call("SetString", 1, (d + "/model.toml").encode()) == 0   # kModel: the path is part of the state blob
    x = np.random.default_rng(seed).standard_normal(480 * 6).astype(np.float32) * 0.1
    for rep in range(3):
        p.call("SetInt", 2, rep % 2)
        out, codes = p.process(x)
        assert set(codes) == {0}
        blob = p.state()
        assert p.call("ReadState", blob, len(blob)) == 0
    p.close()
ts = [threading.Thread(target=work, args=(s,)) for s in (1, 2)]
[t.start() for t in ts]; [t.join() for t in ts]
print("two-thread proxy stress: ok")
PY
fi
