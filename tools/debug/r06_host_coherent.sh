M=$(python - <<'PY'
import sys, tempfile
sys.path.insert(0, "tools")
import make_model
d = tempfile.mkdtemp(); make_model.make_model(d, n_speakers=2); print(d)
PY
)
for v in 1 0; do
  HIP_HOST_COHERENT=$v timeout 120 examples/latency_b1 $M 5000 500 --histogram | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HIP_HOST_COHERENT=$v', d['p50_us'], d['p99_us'], d['max_us'], d['per_call_p50_us'], d['checksum'], d['pitch_hops_claimed'])"
done
