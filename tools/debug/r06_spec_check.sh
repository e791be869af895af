timeout 900 python -m pytest tests/test_gpu_pitch_beside_phone.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity_1stream.py tests/test_gpu_realtime_contract.py tests/test_gpu_cpp_example.py tests/test_gpu_host_layer.py -x -q 2>&1 | tail -5
M=$(python - <<'PY'
import sys, tempfile, os
sys.path.insert(0, "tools")
import make_model
d = tempfile.mkdtemp(); make_model.make_model(d, n_speakers=2); print(d)
PY
)
for i in 1 2; do
  examples/latency_b1 $M 30000 2000 --histogram | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('beside   ', {k: d[k] for k in ('p50_us','p90_us','p99_us','p999_us','max_us','pitch_hops_claimed','pitch_hops_dropped','per_call_p50_us','checksum')})"
  BEATRICE_HIP_NO_SPECULATION=1 examples/latency_b1 $M 30000 2000 --histogram | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one by one', {k: d[k] for k in ('p50_us','p90_us','p99_us','p999_us','max_us','pitch_hops_claimed','pitch_hops_dropped','per_call_p50_us','checksum')})"
done
examples/latency_b1 $M 1500 200 --period-us 10000 --rt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('paced 10 ms', {k: d[k] for k in ('p50_us','p90_us','p99_us','max_us','pitch_hops_claimed')})"
