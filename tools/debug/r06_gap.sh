M=$(python - <<'PY'
import sys, tempfile, os
sys.path.insert(0, "tools")
import make_model
d = tempfile.mkdtemp(); make_model.make_model(d, n_speakers=2); print(d)
PY
)
for g in 0 10 20 40; do
  examples/latency_b1 $M 20000 2000 --histogram --gap-us $g | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gap $g', d['p50_us'], d['per_call_p50_us'])"
done
