"""bench.py as the driver runs it: one JSON line with the contract's keys, at N = 1 and -- rehearsed on the
one GPU a test box has, over gloo -- through its own `--gpus N` launch path (torch.distributed.run child
processes, one rank per process).  The measured multi-GPU numbers come from the driver's 8-GPU node; this
test keeps the control flow of that run from rotting."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _run_bench(args, extra_env=None, timeout=600):
    env = dict(os.environ)
    env.update(extra_env or {})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def have_gpu():
    # asked of the library, not of torch: torch.cuda.is_available() is false in a process where this library opened the
    # device first; bench.py itself runs in child processes
    import torch  # noqa: F401  (before the library: sgdnet_amd/_lib.py, loaded_before_torch -- later tests use parallel.py)
    from sgdnet_amd import _lib
    if _lib.load().sgdnet_device_count() < 1:
        pytest.fail("bench.py needs a HIP device: the backend has no CPU fallback")


@pytest.mark.timeout(900)
def test_bench_single_gpu_line_keeps_the_contract(have_gpu):
    out = _run_bench(["--workload", "tiny", "--steps", "3", "--warmup", "1", "--cpu-epochs", "1"])
    for key in CONTRACT:
        assert key in out, key
    assert out["metric"] == "saga_epochs_per_sec" and out["unit"] == "epochs/s" and out["higher_is_better"] is True
    assert out["n_gpus"] == 1 and out["n_ranks_seen"] == 1 and out["steps"] == 3 and out["warmup"] == 1
    assert out["value"] > 0 and abs(out["value"] * out["ms_per_step"] - 1e3) < 1e-6 * 1e3
    assert out["dtype"] == "f64" and out["vs_baseline"] is None and "workload" in out["config"]
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0.0 < roof["frac"] < 1.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["value"] > 0


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks(have_gpu):
    # `python bench.py --gpus 2`, not inside torch.distributed.run: bench.py starts the two ranks itself;
    # both share the test box's one GPU and reduce over gloo (RCCL wants one device per rank)
    out = _run_bench(["--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                     {"SGDNET_BENCH_BACKEND": "gloo", "SGDNET_BENCH_ONE_GPU": "1"})
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2       # an all-reduce really summed over two ranks
    assert out["steps"] == 3 and out["value"] > 0
    # the default exchange: the ranks' replicas averaged inside their epoch kernels through hipIpc-mapped buffers
    assert out["config"]["samples_per_gpu"] == 50_000 and out["config"]["merge"].startswith("peers")
    assert out["roofline"]["kernel"] == "saga_vs_epoch_kernel"
    assert out["scaling"] == "strong"
    conv = out.get("convergence")
    assert conv is None or conv["epochs"] > 0


@pytest.mark.timeout(900)
def test_bench_rccl_scheme_still_runs(have_gpu):
    # --merge avg: local runs between merges + an all-reduce of the weighted state deltas (gloo here)
    out = _run_bench(["--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                      "--no-convergence", "--merge", "avg"],
                     {"SGDNET_BENCH_BACKEND": "gloo", "SGDNET_BENCH_ONE_GPU": "1"})
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["config"]["merge"].startswith("avg")


@pytest.mark.timeout(900)
def test_bench_falls_back_when_the_linked_epochs_fail(have_gpu):
    # rank 1 never launches: rank 0's linked epoch kernel waits for it at the first merge, gives up after its bounded spin
    # and raises; both ranks then start over with the all-reduce scheme and the line says why
    out = _run_bench(["--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                      "--no-convergence"],
                     {"SGDNET_BENCH_BACKEND": "gloo", "SGDNET_BENCH_ONE_GPU": "1", "SGDNET_BENCH_TEST_PEERS_FAILURE": "1"})
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["value"] > 0
    merge = out["config"]["merge"]
    assert merge.startswith("(peers scheme not used: epochs failed") and "avg:" in merge
