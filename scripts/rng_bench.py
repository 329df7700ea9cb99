"""Times the device-side R-compatible Mersenne-Twister (sgdnet_solver_generate_stream)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import sgdnet_amd as sa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
x = sp.identity(8, format="csc")[:, np.zeros(n, dtype=int)] if False else sp.csc_matrix((np.ones(n), np.zeros(n, dtype=np.int32), np.arange(n + 1)), shape=(4, n))
S = sa.SagaSolver(x, np.zeros((1, n)), family="gaussian", n_classes=1)
rng = sa.RRng(1)
S.generate_stream(rng, n)
t = time.time()
for _ in range(5):
    S.generate_stream(rng, n)
dt = (time.time() - t) / 5
print(f"device MT19937: {n} draws in {dt*1e3:.2f} ms ({n/dt/1e9:.2f} G draws/s)")
ref = sa.RRng(1).stream(n, 1000)
S2 = sa.RRng(1)
S.generate_stream(S2, n)
assert (S.get_stream(0, 1000) == ref).all()
print("first 1000 draws equal the host generator")
