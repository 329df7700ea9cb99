import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sgdnet_amd as sa
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
ir = np.load(os.path.join(GOLD, "iris.npz"))
sa.sgdnet(ir["x"], ir["y"], family="multinomial", alpha=0.8, nlambda=3)
for rep in range(2):
    t = time.time()
    fit = sa.sgdnet(ir["x"], ir["y"], family="multinomial", alpha=0.8, seed=1)
    print("iris exact", time.time() - t, fit.npasses, flush=True)
