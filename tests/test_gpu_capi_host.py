"""The C ABI from a non-Python host: tests/capi_host/host_main.cpp (plain C++ + HIP runtime, include/ftr.h, -lftr_hip, no
torch) runs forward / backward / cummin / prune ranges on a golden fixture written out as raw files, and this test
compares what it wrote back with the fixture.  Stand-in for the TF-ROCm op shim (tf_fast_rnnt_op.cc:48-165), which
cannot be built in this image."""
import os
import subprocess

import numpy as np
import pytest

from helpers import max_rel

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
HOST = os.path.join(HERE, "capi_host", "capi_host.bin")


@pytest.mark.parametrize("name,r", [("seed1234_B2_T10_S7_C4", 3), ("seed12345_B2_T200_S50_C50", 5)])
def test_cxx_host_runs_the_abi(dev, tmp_path, name, r):
    assert os.path.exists(HOST), "tests/capi_host/capi_host.bin is missing: run __graft_entry__.build()"
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    px, py, bd = g["simple_px"], g["simple_py"], g["boundary"].astype(np.int32)
    B, S, T1 = px.shape
    T = py.shape[2]
    assert T1 == T + 1
    rng = np.random.default_rng(0)
    cin = rng.integers(-50, 50, (5, 37)).astype(np.int32)
    d = str(tmp_path)
    for fn, a in (("px", px), ("py", py), ("boundary", bd), ("gx", g["smoothed_px_grad"]), ("gy", g["smoothed_py_grad"]),
                  ("cummin_in", cin)):
        np.ascontiguousarray(a).tofile(os.path.join(d, fn + ".bin"))
    out = subprocess.run([HOST, d, str(B), str(S), str(T), "0", str(r), "5", "37"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    rd = lambda fn, dt, shape: np.fromfile(os.path.join(d, fn + ".bin"), dtype=dt).reshape(shape)
    np.testing.assert_allclose(rd("ans", np.float32, (B,)), -g["simple_loss"], rtol=1e-4)
    # against the fixture's float64-recursion occupancies, as in test_against_committed_golden_fixtures
    assert max_rel(rd("px_grad", np.float32, px.shape), g["simple_px_grad_f64"]) <= 2e-5
    assert max_rel(rd("py_grad", np.float32, py.shape), g["simple_py_grad_f64"]) <= 2e-5
    np.testing.assert_allclose(rd("ans_grad", np.float32, (B,)), 1.0, rtol=2e-4)       # the self check returns the seed
    assert np.array_equal(rd("cummin_out", np.int32, cin.shape), np.minimum.accumulate(cin, axis=1))
    want = g[f"ranges_r{r}"]
    assert np.array_equal(rd("ranges", np.int32, want.shape), want)                      # integer output: bit-exact
