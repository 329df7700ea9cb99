"""The C-ABI library loads on a CPU-only host, exports every symbol include/*.h declares,
and refuses to compute without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sgdnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgdnet_[a-z_0-9]+)\s*\(", text)) - {"sgdnet_unif_fn"})


def test_every_declared_symbol_is_exported():
    from sgdnet_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for sym in declared:
        assert hasattr(L, sym), f"{sym} declared in include/sgdnet_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared


def test_abi_version_and_rng_match_r(oracle):
    import sgdnet_amd as sa
    import re
    from sgdnet_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sgdnet_hip.h")).read()
    declared = int(re.search(r"#define SGDNET_ABI_VERSION (\d+)", hdr).group(1))
    assert sa.load().sgdnet_abi_version() == declared == _lib.ABI_VERSION     # library, header and binding agree
    # R: set.seed(1); runif(3)
    np.testing.assert_allclose(sa.RRng(1).unif(3), [0.2655087, 0.3721239, 0.5728534], atol=5e-8)
    # product RNG == oracle RNG, draw for draw
    for seed, n in ((4, 10_000_000), (123, 150)):
        a = sa.RRng(seed).stream(n, 5000)
        b = oracle.Rng(seed).stream(n, 5000)
        assert np.array_equal(a, b)


def test_no_device_fails_loudly():
    import sgdnet_amd as sa
    from sgdnet_amd import SgdnetError
    if sa.load().sgdnet_device_count() > 0:
        pytest.skip("a HIP device is present")
    x = np.random.default_rng(0).normal(size=(20, 3))
    y = x[:, 0] + 1.0
    with pytest.raises(SgdnetError) as e:
        sa.sgdnet(x, y, family="gaussian", nlambda=3)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    import scipy.sparse as sp
    with pytest.raises(SgdnetError) as e:
        sa.SagaSolver(sp.csc_matrix(x.T), y.reshape(1, -1), family="gaussian", n_classes=1)
    assert e.value.code == -2


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sgdnet_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "sgdnet_oracle" not in text, f


def test_auto_batch_rule():
    import sgdnet_amd as sa
    # 2 * L_max / diag(X'X/n), clamped to [64, 131072]
    assert sa.auto_batch(50.0, 0.001) == 100000
    assert sa.auto_batch(50.0, 0.01) == 10000
    assert sa.auto_batch(50.0, 1e-9) == 131072
    assert sa.auto_batch(1.0, 1.0) == 64
    assert sa.auto_batch(0.0, 0.0) == 64


def test_shard_window_saves_a_short_last_round():
    import sgdnet_amd as sa
    assert sa.shard_window(131072, 1250000) == 138889        # config 4, 8 shards: 9 rounds of 138 889 instead of 9 + a tail
    assert -(-1250000 // 138889) == 9
    assert sa.shard_window(131072, 262144) == 131072         # no tail to save
    assert sa.shard_window(131072, 150000) == 131072         # one round would need a window > 9/8 of the rule's
    assert sa.shard_window(131072, 125000) == 131072         # a single round already
    assert sa.shard_window(64, 100) == 64
    assert sa.shard_window(1000, 10050) == 1005
    assert sa.shard_window(0, 10) == 0


def test_backend_options_are_the_documented_ones():
    """sgdnet_set_option is the one documented switchboard (include/sgdnet_hip.h); unknown names and values
    outside the documented range are refused, and the shipped library reads no kernel-selection environment
    variable (they exist in -DSGDNET_EXPERIMENTS builds only)."""
    import sgdnet_amd as sa
    defaults = {"virtual_shards": -1, "rng_generators": 0, "window_eigenvalue": 1, "host_setup": 0,
                "exact_epoch_blocks": 1, "exact_row_registers": 1}
    for name, dflt in defaults.items():
        assert sa.get_option(name) == dflt
        assert name in open(os.path.join(ROOT, "include", "sgdnet_hip.h")).read()
    with sa.option("virtual_shards", 4):
        assert sa.get_option("virtual_shards") == 4
    assert sa.get_option("virtual_shards") == -1
    for name, bad in (("virtual_shards", 9), ("rng_generators", -1), ("host_setup", 2), ("exact_row_registers", 5),
                      ("no_such_option", 0)):
        with pytest.raises(sa.SgdnetError):
            sa.set_option(name, bad)
    src = os.path.join(ROOT, "sgdnet_amd", "csrc")
    for f in os.listdir(src):
        text = open(os.path.join(src, f)).read()
        for m in re.finditer(r'getenv\("(SGDNET_[A-Z0-9_]+)"\)', text):
            # progress printing and builds that are never shipped
            assert m.group(1) in ("SGDNET_TRACE", "SGDNET_ABLATE", "SGDNET_PHASE_DUMP"), (f, m.group(1))


def test_graft_entry_library_check_passes():
    """__graft_entry__.build() ends with this check; it once compared the ABI version with a literal."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    ge = importlib.import_module("__graft_entry__")
    ge.check_library()
    assert "== 1" not in open(os.path.join(ROOT, "__graft_entry__.py")).read()
