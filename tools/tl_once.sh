export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_gate -o t -- python3 bench.py --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0 --steps 30 --warmup 5 --kt-every 1000 > /dev/null 2>&1
python3 tools/step_timeline.py $(find gpurun_out/tl_gate -name "*kernel_trace.csv") 6
rm -rf gpurun_out/tl_gate
