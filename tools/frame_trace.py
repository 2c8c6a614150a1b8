"""The frame leg of bench.py alone (N=500, K2=600): for rocprofv3 --kernel-trace (what one frame's launches are)."""
import importlib, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pre3 = importlib.import_module("3pre_amd"); synth = importlib.import_module("3pre_amd.synth")
print(json.dumps(bench.frame_leg(pre3, synth, frames=int(sys.argv[1]) if len(sys.argv) > 1 else 12)), flush=True)
