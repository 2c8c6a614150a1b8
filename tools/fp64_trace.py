"""The fp64_n200 leg of bench.py alone (BASELINE configs[1]: N=200, fp64, predict + update of all measured landmarks): for rocprofv3 --kernel-trace."""
import importlib, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pre3 = importlib.import_module("3pre_amd"); synth = importlib.import_module("3pre_amd.synth")
print(json.dumps(bench.fp64_n200_leg(pre3, synth, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 40)), flush=True)
