"""One secondary record of bench.py alone: python tools/sec_one.py <function name of tools/bench_secondary.py> [reps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_secondary
fn = getattr(bench_secondary, sys.argv[1])
print(json.dumps(fn("cuda:0")), flush=True)
