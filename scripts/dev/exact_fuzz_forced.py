"""The random-fit sweep of tests/test_gpu_fuzz.py with the multi-wavefront exact kernels forced wherever they are legal
(option exact_row_registers = 3): usage exact_fuzz_forced.py [first_seed] [count]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import sgdnet_amd as sa
import test_gpu_fuzz as F
from oracle import pyoracle as po
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sa.set_option("exact_row_registers", 3)
bad = 0
fn = getattr(F.test_random_fit_matches_the_oracle_fit, "__wrapped__", F.test_random_fit_matches_the_oracle_fit)
for seed in range(first, first + count):
    try:
        fn(sa, po, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:200], flush=True)
print(f"{count} fits, {bad} failures", flush=True)
