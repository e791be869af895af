"""bench.py's timed workloads must be the shapes the -m gpu parity tests run against the oracle.

Round 5's bench moved to four hops per step while the tests that said "what the bench times" still ran two.
bench.py names its defaults and the parity tests of every --config (bench.PARITY_TESTS); this test reads the parametrisation of
those tests without running them and fails when a default has no parity case."""
import importlib.util
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _cases(fn):
    """[{parameter: value}] over the test's (stacked) parametrize marks"""
    out = [{}]
    for mark in getattr(fn, "pytestmark", []):
        if mark.name != "parametrize":
            continue
        names = [n.strip() for n in mark.args[0].split(",")] if isinstance(mark.args[0], str) else list(mark.args[0])
        rows = []
        for v in mark.args[1]:
            v = getattr(v, "values", v)          # pytest.param(...)
            rows.append(dict(zip(names, v if len(names) > 1 else [v])))
        out = [dict(a, **b) for a in out for b in rows]
    return out


def test_every_bench_default_has_a_gpu_parity_case():
    sys.path.insert(0, os.path.join(REPO, "tests"))
    bench = _load(os.path.join(REPO, "bench.py"), "bench_for_defaults")
    H = bench.DEFAULT_HOPS_PER_STEP_TICK
    assert H in (1, 2, 4)
    assert set(bench.PARITY_TESTS) == set(bench.DEFAULT_STREAMS) == {2, 3, 4}
    for config, tests in bench.PARITY_TESTS.items():
        assert tests, "config %d has no parity test" % config
        for fname, func, h_name, b_name in tests:
            mod = _load(os.path.join(REPO, "tests", fname), "defaults_probe_" + fname[:-3])
            fn = getattr(mod, func)
            marks = [m.name for m in getattr(mod, "pytestmark", [])] if isinstance(getattr(mod, "pytestmark", None), list) else [getattr(getattr(mod, "pytestmark", None), "name", None)]
            assert "gpu" in marks or any(m.name == "gpu" for m in getattr(fn, "pytestmark", [])), "%s::%s is not a -m gpu test" % (fname, func)
            cases = _cases(fn)
            at_default = [c for c in cases if c.get(h_name) == H]
            assert at_default, "%s::%s has no case at the bench's default of %d hops per step (has %s)" % (
                fname, func, H, sorted({c.get(h_name) for c in cases}))
            if b_name is not None:
                want = bench.DEFAULT_STREAMS[config]
                assert any(c.get(b_name) == want for c in at_default), "%s::%s has no case at %d streams and %d hops per step" % (fname, func, want, H)


def test_argument_defaults_are_the_named_constants():
    """the argparse defaults are the constants the test above reads (a literal edited in main() would slip past it)"""
    src = open(os.path.join(REPO, "bench.py")).read()
    assert "a.hops_per_step = DEFAULT_HOPS_PER_STEP_TICK if" in src
    assert 'ap.add_argument("--config", type=int, default=DEFAULT_CONFIG' in src
    assert 'ap.add_argument("--streams", type=int, default=DEFAULT_STREAMS[DEFAULT_CONFIG]' in src
