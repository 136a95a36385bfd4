"""The drop-in boundary end to end (getProbableTransformsSuper4PCS per object through shim/test_shim):
file hand-off vs in-memory overload, 6 calls each.  usage: python tools/dropin_time.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
print(json.dumps(bench.drop_in_row(), indent=1))
