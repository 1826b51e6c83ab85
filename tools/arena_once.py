"""One compare_networks at the reference's arena parameters (bench.py arena_block), without the headline around it: A/B aid."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import azhip
r = bench.arena_block(azhip, 0)
r2 = bench.arena_block(azhip, 0)
print(json.dumps({"first": r["seconds"], "second": r2["seconds"], "avgr": r2["avgr"]}))
