cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests/test_gpu_tick_pipeline.py tests/test_gpu_tick_hops2.py tests/test_gpu_tick_ragged.py -x -q -m gpu > gpurun_out/r06a/pytest_tick.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06a/pytest_tick.log
tail -5 gpurun_out/r06a/pytest_tick.log
bash tools/debug/r06_split_ab.sh 2 > gpurun_out/r06a/split_ab.txt 2>&1
cat gpurun_out/r06a/split_ab.txt
export BEATRICE_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/lib_meas.so
BEATRICE_HIP_TICK_TRACE=/tmp/trace_split.txt python tools/debug/time_tick.py 256 - 4 > /dev/null 2>&1; python tools/debug/tick_trace.py /tmp/trace_split.txt > gpurun_out/r06a/trace_split.txt 2>&1
BEATRICE_HIP_TICK_ONE_KERNEL=1 BEATRICE_HIP_TICK_TRACE=/tmp/trace_one.txt python tools/debug/time_tick.py 256 - 4 > /dev/null 2>&1; python tools/debug/tick_trace.py /tmp/trace_one.txt > gpurun_out/r06a/trace_one.txt 2>&1
unset BEATRICE_HIP_LIB
python tools/make_model.py /tmp/m1 1 > /dev/null 2>&1 || python -c "
import sys; sys.path.insert(0,'tools'); import make_model, os; os.makedirs('/tmp/m1',exist_ok=True); make_model.make_model('/tmp/m1', n_speakers=1)"
./examples/latency_b1 /tmp/m1 100000 2000 0 --histogram > gpurun_out/r06a/latency_b1.json 2>&1
cat gpurun_out/r06a/latency_b1.json
