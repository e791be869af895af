cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
bash tools/debug/r06_serial_ab.sh 2 > gpurun_out/r06b/serial_ab.txt 2>&1
cat gpurun_out/r06b/serial_ab.txt
export BEATRICE_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/lib_meas.so
for mode in 1 2; do
  BEATRICE_HIP_TICK_LIGHT_SERIAL=$mode BEATRICE_HIP_TICK_TRACE=/tmp/trace_serial$mode.txt python tools/debug/time_tick.py 256 - 4 > /dev/null 2>&1
  python tools/debug/tick_trace.py /tmp/trace_serial$mode.txt > gpurun_out/r06b/trace_serial$mode.txt 2>&1
done
BEATRICE_HIP_TICK_TRACE=/tmp/trace_split.txt python tools/debug/time_tick.py 256 - 4 > /dev/null 2>&1; python tools/debug/tick_trace.py /tmp/trace_split.txt > gpurun_out/r06b/trace_split.txt 2>&1
unset BEATRICE_HIP_LIB
python -c "
import sys; sys.path.insert(0,'tools'); import make_model, os; os.makedirs('/tmp/m1',exist_ok=True); make_model.make_model('/tmp/m1', n_speakers=1)"
for i in 1 2 3; do ./examples/latency_b1 /tmp/m1 100000 2000 0 --histogram > gpurun_out/r06b/latency_b1_run$i.json 2>&1; cat gpurun_out/r06b/latency_b1_run$i.json; done
