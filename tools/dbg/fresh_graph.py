import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
import multiprocessing as mp
from test_gpu_model import _fresh_graph_fit_worker
class Q:
    def put(self, out):
        for k, (n, hist) in out.items():
            print("graphs" if k else "eager", n)
            for h in hist: print("   ", h)
_fresh_graph_fit_worker(Q())
