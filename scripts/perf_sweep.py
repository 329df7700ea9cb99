"""Perf iteration helper: one workload, several batch sizes; prints ms/epoch and kernel times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sgdnet_amd as sa
from sgdnet_amd import data as D
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
batches = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [20000, 65536]
n, p, dens, seed, family, K = {"C4": (10_000_000, 10_000, 0.001, 4, "binomial", 1),
                               "C3": (1_000_000, 1_000, 0.01, 3, "binomial", 1),
                               "C4s": (2_000_000, 10_000, 0.001, 4, "binomial", 1),
                               "C4z6": (10_000_000, 10_000, 0.0006, 4, "binomial", 1),
                               "C5s": (2_000_000, 100_000, 0.0001, 5, "multinomial", 10),
                               "C5": (50_000_000, 100_000, 0.0001, 5, "multinomial", 10)}[wl]
t = time.time(); pr = D.make_sparse_glm(n, p, dens, family=family, n_classes=K, seed=seed); X = D.as_scipy(pr)
print(f"{wl}: gen {time.time()-t:.1f}s nnz={X.nnz}", flush=True)
epochs = 2 + 3 * len(batches) + 1
stream = sa.RRng(seed).stream(n, n * 4)
row_sq = np.add.reduceat(pr["val"] ** 2, pr["ptr"][:-1])
gamma = D.step_size(row_sq.max(), 0.5 / n, True, family, n)
S = sa.SagaSolver(X, pr["y"], family=family, n_classes=K)
S.set_penalty("elasticnet", gamma, 0.5 / n, 0.5 / n)
S.upload_stream(stream)
ab = D.algorithmic_bytes(S.row_nnz, stream[:n], K)
for batch in batches:
    S.enqueue_epochs(1, batch=batch); S.sync()
    t = time.time(); S.enqueue_epochs(3, batch=batch, stream_offset=n); S.sync(); dt = (time.time() - t) / 3
    pe = S.profile_epoch(batch=batch)
    print(f"batch={batch:6d}: {dt*1e3:7.2f} ms/epoch {1/dt:7.1f} epochs/s {ab/dt/1e9:7.1f} GB/s | gather {pe['gather_ms']/pe['gather_launches']*1e3:6.1f} us x{pe['gather_launches']} "
          f"({ab/pe['gather_ms']/1e6:6.1f} GB/s) sweep {pe['sweep_ms']/pe['sweep_launches']*1e3:5.1f} us", flush=True)
