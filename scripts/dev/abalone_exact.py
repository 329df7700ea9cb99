"""abalone's 100-lambda path in exact mode, alone (for rocprofv3 --stats: kernel time against wall time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sgdnet_amd as sa
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
ab = np.load(os.path.join(GOLD, "abalone.npz"))
sa.sgdnet(ab["x"], ab["y"], nlambda=3, mode="exact", family="gaussian")
t = time.time(); fit = sa.sgdnet(ab["x"], ab["y"], mode="exact", seed=1, family="gaussian"); dt = time.time() - t
print(f"abalone exact: {dt:.3f} s, npasses={fit.npasses:.0f}, {dt / fit.npasses / ab['x'].shape[0] * 1e6:.3f} us per iteration", flush=True)
