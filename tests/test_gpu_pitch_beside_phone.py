"""The pitch call beside the phone call (csrc/abi.hip "pre-execution", include/beatrice_batch.h BeatriceHip_PitchSpeculation).

The reference's hop calls ExtractPhone1 and EstimatePitch1 on the same 160 samples, one after the other (src/common/processor_core_2.cc:184,188).
The library learns that pattern per (phone context, pitch context) pair and from then on starts the pitch context's hop inside the phone call;
EstimatePitch1 claims it when it is called with the same samples / bin range / estimator / parameters and drops it otherwise.  Whatever the host
does -- other samples, a bin range moved between the two calls, a phone call without a pitch call, two pitch calls in a row, the partner destroyed,
the estimator reloaded, two plugin instances on one thread --, every call's results must equal the oracle's for the same call sequence, bit for bit."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Pair:
    """one plugin instance's phone + pitch contexts through `abi` (the product or the oracle)"""

    def __init__(self, bv, abi, models, speaker=1, k=2):
        self.bv, self.abi, self.m = bv, abi, models
        self.pc, self.tc = abi.CreatePhoneContext1(), abi.CreatePitchContext1()
        abi.SetCodebook(self.pc, bv.fptr(models.tables.codebooks[speaker]))
        abi.SetVQNumNeighbors(self.pc, k)
        self.range(1, 383)

    def range(self, lo, hi):
        self.abi.SetMinQuantizedPitch(self.tc, lo)
        self.abi.SetMaxQuantizedPitch(self.tc, hi)

    def phone(self, x):
        v = np.zeros(self.bv.PHONE_CH, np.float32)
        self.abi.ExtractPhone1(self.m.phone, self.bv.fptr(np.ascontiguousarray(x)), self.bv.fptr(v), self.pc)
        return v

    def pitch(self, x):
        q, f = np.zeros(1, np.int32), np.zeros(4, np.float32)
        self.abi.EstimatePitch1(self.m.pitch, self.bv.fptr(np.ascontiguousarray(x)), self.bv.iptr(q), self.bv.fptr(f), self.tc)
        return np.concatenate([q.astype(np.float32), f])

    def new_pitch_context(self):
        self.abi.DestroyPitchContext1(self.tc)
        self.tc = self.abi.CreatePitchContext1()
        self.range(1, 383)

    def close(self):
        self.abi.DestroyPhoneContext1(self.pc)
        self.abi.DestroyPitchContext1(self.tc)


def stats(product, tc):
    hit, miss = C.c_longlong(0), C.c_longlong(0)
    on = product.BeatriceHip_PitchSpeculation(tc, C.byref(hit), C.byref(miss))
    return on, hit.value, miss.value


def both(bv, product, oracle, model_dir, script):
    """runs script(pair, is_product) against the product and the oracle; returns (product's list of arrays, oracle's, whatever the script returned on the product)"""
    out, extra = [], None
    for abi in (product, oracle):
        m = bv.Models(abi, model_dir)
        p = Pair(bv, abi, m)
        got = []
        r = script(p, got, abi is product)
        if abi is product:
            extra = r
        p.close()
        m.close()
        out.append(got)
    return out[0], out[1], extra


def same(got, want):
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), "call %d: max-abs %g" % (i, np.abs(g - w).max())


def test_the_reference_call_pattern_is_claimed_and_equals_the_oracle(bv, oracle, product, model_dir):
    bv.bind_batch(product)
    hops = 120
    x = bv.synth_audio(160 * hops, seed=301)

    def script(p, got, is_product):
        for i in range(hops):
            h = x[i * 160:(i + 1) * 160]
            got.append(p.phone(h))
            got.append(p.pitch(h))
        return stats(product, p.tc) if is_product else None

    got, want, (on, hit, miss) = both(bv, product, oracle, model_dir, script)
    same(got, want)
    assert on == 1 and miss == 0 and hit == hops - 1, (on, hit, miss)   # (the first hop teaches the pair)


@pytest.mark.parametrize("every", [29, 5])
def test_whatever_the_host_does_instead_equals_the_oracle(bv, oracle, product, model_dir, every):
    """other samples for the pitch call every `every` hops, the bin range moved between the two calls, a hop without its pitch call, a pitch call
    without a phone call before it.  every = 29: a few dropped hops, the mechanism stays on; every = 5: more than it tolerates, it turns itself off."""
    bv.bind_batch(product)
    hops = 110
    x = bv.synth_audio(160 * hops, seed=302)
    y = bv.synth_audio(160 * hops, seed=303)

    def script(p, got, is_product):
        for i in range(hops):
            h = x[i * 160:(i + 1) * 160]
            got.append(p.phone(h))
            if i == 40:
                continue                                   # no pitch call for this hop
            if i % every == 3:
                got.append(p.pitch(y[i * 160:(i + 1) * 160]))   # other samples than the phone call's
                continue
            if i == 20:
                p.range(120, 200)                          # between the two calls
            if i == 62:
                p.range(1, 383)
            got.append(p.pitch(h))
            if i == 50:
                got.append(p.pitch(y[:160]))               # two pitch calls in a row
        return stats(product, p.tc) if is_product else None

    got, want, (on, hit, miss) = both(bv, product, oracle, model_dir, script)
    same(got, want)
    if every == 29:
        assert on == 1 and 6 <= miss <= 8 and hit > 90, (on, hit, miss)
    else:
        assert on == 0 and miss >= 8 and hit > 0, (on, hit, miss)


def test_partner_destroyed_and_estimator_reloaded(bv, oracle, product, model_dir):
    bv.bind_batch(product)
    hops = 60
    x = bv.synth_audio(160 * hops, seed=304)
    path = os.path.join(model_dir, "pitch_estimator.bin").encode()

    def script(p, got, is_product):
        seen = []
        for i in range(hops):
            h = x[i * 160:(i + 1) * 160]
            got.append(p.phone(h))
            if i == 15:
                p.new_pitch_context()                      # with a pre-executed hop of the old context under way
            if i == 35:
                assert p.abi.ReadPitchEstimatorParameters(p.m.pitch, path) == 0   # the same file again: new device blob, new generation
            got.append(p.pitch(h))
            if is_product and i in (14, 16, 17, 34, 35, 36, 59):
                seen.append((i,) + stats(product, p.tc))
        return seen

    got, want, seen = both(bv, product, oracle, model_dir, script)
    same(got, want)
    seen = {s[0]: s[1:] for s in seen}
    assert seen[14][0] == 1 and seen[14][2] == 0
    assert seen[16] == (1, 1, 0)            # the new context: learnt at its first call (hop 15), hop 16 claimed ...
    assert seen[17] == (1, 2, 0)
    assert seen[35][2] == seen[34][2] + 1   # the hop pre-executed with the old parameters is dropped
    assert seen[59][2] == seen[35][2] and seen[59][1] > seen[36][1]


def test_two_instances_on_one_thread(bv, oracle, product, model_dir):
    """A host runs many plugin instances (src/vst/factory.cc:21), possibly on one audio thread: phone A, pitch A, phone B, pitch B ..."""
    bv.bind_batch(product)
    hops = 50
    xa = bv.synth_audio(160 * hops, seed=305)
    xb = bv.synth_audio(160 * hops, seed=306)
    res = {}
    for abi in (product, oracle):
        m = bv.Models(abi, model_dir)
        a, b = Pair(bv, abi, m, speaker=0), Pair(bv, abi, m, speaker=1, k=4)
        got = []
        for i in range(hops):
            got.append(a.phone(xa[i * 160:(i + 1) * 160]))
            got.append(a.pitch(xa[i * 160:(i + 1) * 160]))
            got.append(b.phone(xb[i * 160:(i + 1) * 160]))
            got.append(b.pitch(xb[i * 160:(i + 1) * 160]))
        if abi is product:
            sa, sb = stats(product, a.tc), stats(product, b.tc)
        a.close(); b.close(); m.close()
        res[abi is product] = got
    same(res[True], res[False])
    assert sa == (1, hops - 1, 0) and sb == (1, hops - 1, 0), (sa, sb)


def test_switched_off_by_the_environment(model_dir):
    """BEATRICE_HIP_NO_SPECULATION=1 (read once per process: hence subprocesses): the same samples out of the whole chain, nothing claimed."""
    tool = os.path.join(REPO, "examples", "latency_b1")
    if not os.path.exists(tool):
        pytest.skip("examples/latency_b1 not built")
    outs = []
    for env in ({}, {"BEATRICE_HIP_NO_SPECULATION": "1"}):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([tool, model_dir, "300", "20"], capture_output=True, text=True, env=e, timeout=300)
        assert r.returncode == 0, r.stderr[-400:]
        import json
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["checksum"] == outs[1]["checksum"] and outs[0]["last_hop_peak"] == outs[1]["last_hop_peak"]
    assert outs[0]["pitch_hops_claimed"] >= 300 and outs[1]["pitch_hops_claimed"] == 0


def test_instances_on_threads_of_their_own(bv, oracle, product, model_dir):
    """four plugin instances, each on its own thread (ctypes releases the GIL inside the calls): every instance equals its oracle twin"""
    import threading
    bv.bind_batch(product)
    hops, n = 80, 4
    xs = [bv.synth_audio(160 * hops, seed=310 + t) for t in range(n)]
    res = {}
    for abi in (product, oracle):
        m = bv.Models(abi, model_dir)
        pairs = [Pair(bv, abi, m, speaker=t % 2, k=2 * (t % 3)) for t in range(n)]
        outs = [[] for _ in range(n)]

        def work(t):
            for i in range(hops):
                h = xs[t][i * 160:(i + 1) * 160]
                outs[t].append(pairs[t].phone(h))
                outs[t].append(pairs[t].pitch(h))

        if abi is product:
            th = [threading.Thread(target=work, args=(t,)) for t in range(n)]
            [t.start() for t in th]
            [t.join() for t in th]
            st = [stats(product, p.tc) for p in pairs]
        else:
            for t in range(n):
                work(t)
        for p in pairs:
            p.close()
        m.close()
        res[abi is product] = outs
    for t in range(n):
        same(res[True][t], res[False][t])
    # (the registry lock is only ever TRIED on a per-hop call: while another thread pairs its contexts, a hop goes without its pre-execution)
    assert all(s[0] == 1 and s[2] == 0 and hops - 10 <= s[1] <= hops - 1 for s in st), st


def test_phone_and_pitch_calls_of_one_instance_on_two_threads(bv, oracle, product, model_dir):
    """A host may run the two independent calls of a hop on two threads.  The pair is learnt on one thread first; then thread A makes the phone calls
    and thread B the pitch calls, unsynchronised: whatever interleaving happens, each context's results equal the oracle's for its own call sequence."""
    import threading
    bv.bind_batch(product)
    first, hops = 10, 70
    x = bv.synth_audio(160 * hops, seed=320)
    res = {}
    for abi in (product, oracle):
        m = bv.Models(abi, model_dir)
        p = Pair(bv, abi, m)
        ph, pi = [], []
        for i in range(first):
            ph.append(p.phone(x[i * 160:(i + 1) * 160]))
            pi.append(p.pitch(x[i * 160:(i + 1) * 160]))

        def phones():
            for i in range(first, hops):
                ph.append(p.phone(x[i * 160:(i + 1) * 160]))

        def pitches():
            for i in range(first, hops):
                pi.append(p.pitch(x[i * 160:(i + 1) * 160]))

        if abi is product:
            assert stats(product, p.tc)[0] == 1
            th = [threading.Thread(target=phones), threading.Thread(target=pitches)]
            [t.start() for t in th]
            [t.join() for t in th]
        else:
            phones()
            pitches()
        p.close()
        m.close()
        res[abi is product] = (ph, pi)
    same(res[True][0], res[False][0])
    same(res[True][1], res[False][1])
