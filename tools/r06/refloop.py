"""Where does the reference loop's time go?  bench.reference_loop alone, then again after a GraphedEpoch exists / after gc.freeze()."""
import gc, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B
sys.argv = ["bench.py"] + sys.argv[1:]
args = B.parse()
B.cap_cpu_threads(1)
device = torch.device("cuda", 0)
hp = dict(B.HP)
import efficient_gnns_amd.data as D
import efficient_gnns_amd.models as PM
B.seed_all(0)
data = D.arxiv_like(1.0, seed=0)
d = B.to_device(data, device)
def show(tag):
    r = B.reference_loop(args, d, device, hp, epochs=30)
    print(tag, {k: v for k, v in r.items() if k not in ("what",)}, flush=True)
import importlib.util
show("fresh")
show("again")
# A/B of the deferred inference conv (accel.DEFER_CONV) inside the lazy leg, same process: bench.reference_loop loads its own accel module
# from the file, so the switch is flipped in the source of truth -- an environment variable read at import
