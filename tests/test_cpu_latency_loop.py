"""examples/latency_b1.cc -- configs[1]'s per-hop timing loop on the reference ABI (processor_core_2.cc:184,188,253) -- built against
the ORACLE library (oracle/latency_b1_on_oracle): the loop's call protocol produces sound and its one-line report parses, on a box
without a GPU.  The product build of the same source is run by tests/test_gpu_cpp_example.py and by bench.py."""
import json
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_latency_loop_on_the_oracle(built, model_dir):
    exe = os.path.join(REPO, "oracle", "latency_b1_on_oracle")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "latency_b1_on_oracle"])
    r = subprocess.run([exe, model_dir, "40", "4", "1", "--histogram"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["hops"] == 40 and len(line["slowest"]) == 10
    assert line["p50_us"] <= line["p99_us"] <= line["p999_us"] <= line["max_us"] == line["slowest"][0]["us"]
    assert line["hops_over_10ms"] <= line["hops_over_1ms"] <= 40
    assert line["hops_over_1ms_with_an_involuntary_switch_nearby"] + line["hops_over_1ms_without"] == line["hops_over_1ms"]
    assert line["involuntary_context_switches"] >= 0
    assert line["last_hop_peak"] > 1e-3                      # the stream produces sound
    assert set(line["per_call_p50_us"]) == {"ExtractPhone1", "EstimatePitch1", "GenerateWaveform1"}
    # bad arguments are refused, not crashed on
    assert subprocess.run([exe, model_dir, "5", "0", "99"], capture_output=True, timeout=60).returncode == 2
    assert subprocess.run([exe, os.path.join(model_dir, "nowhere")], capture_output=True, timeout=60).returncode == 1
