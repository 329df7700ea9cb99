"""Which fused-epoch test leaves torch.cuda.is_available() false in the same process?  (development probe)"""
import subprocess, sys
for k in ("equals_the_separate", "odd_number", "generators_inside", "linked_solvers", "sharded_over_two", "two_processes"):
    code = (f"import pytest, sys; rc = pytest.main(['tests/test_gpu_fused.py', '-q', '-m', 'gpu', '-k', '{k}', '-p', 'no:cacheprovider']);"
            "import torch; print('AFTER', '" + k + "', 'rc', rc, 'cuda', torch.cuda.is_available(), flush=True)")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    print([l for l in r.stdout.splitlines() if l.startswith("AFTER")], r.stderr[-300:] if "AFTER" not in r.stdout else "", flush=True)
