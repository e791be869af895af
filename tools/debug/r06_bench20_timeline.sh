mkdir -p gpurun_out/r06g
for i in 1 2 3; do timeout 200 python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('enqueue_ms'), d.get('host_work_ms_per_step'))"; done
R=$PWD; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pb
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pb -o k -- python $R/bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 | cut -c1-200
python $R/tools/debug/launch_timeline.py $(find /tmp/pb -name k_kernel_trace.csv | head -1) 47 --skip=32
