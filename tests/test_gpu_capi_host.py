"""The C ABI from a non-Python host: tests/capi_host/host_main.cpp (plain C++ + HIP runtime, include/ftr.h, -lftr_hip, no
torch) runs forward / backward / cummin / prune ranges on a golden fixture written out as raw files, and this test
compares what it wrote back with the fixture.  Stand-in for the TF-ROCm op shim (tf_fast_rnnt_op.cc:48-165), which
cannot be built in this image."""
import os
import subprocess

import numpy as np
import pytest
import torch

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


@pytest.mark.parametrize("name,r", [("c1_B2_T8_S4_C16", 3), ("seed1234_B2_T10_S7_C4", 3), ("seed12345_B2_T200_S50_C50", 5)])
def test_cxx_host_runs_the_python_level_entry_points(ft, dev, tmp_path, name, r):
    """The entry points that replace the reference's Python functions, from the compiled host: the fused px / py builder
    (when C % 4 == 0) and the simple loss's backward through the library-GEMM entry, do_rnnt_pruning in both output modes, and
    the band-native pruned loss with its gradient -- compared with the golden fixture (builder px / py, pruned loss and
    gradient), with the package's autograd node (d am, d lm) and with numpy (the gather, bit-exact)."""
    assert os.path.exists(HOST), "tests/capi_host/capi_host.bin is missing: run __graft_entry__.build()"
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    am, lm, sym, bd = g["am"], g["lm"], g["symbols"].astype(np.int32), g["boundary"].astype(np.int32)
    ranges = g[f"ranges_r{r}"].astype(np.int32)
    B, T, C = am.shape
    S = lm.shape[1] - 1
    blank = int(g["termination_symbol"])
    am_p = np.broadcast_to(am[:, :, None, :], (B, T, r, C))
    lm_p = np.take_along_axis(lm[:, None, :, :], ranges[:, :, :, None].astype(np.int64), axis=2)     # lm[b, ranges[b,t,k], :]
    logits = (1.0 / (1.0 + np.exp(-(am_p + lm_p)))).astype(np.float32)                               # the fixture's joiner stand-in
    d = str(tmp_path)
    for fn, a in (("am", am), ("lm", lm), ("symbols", sym), ("boundary", bd), ("ranges", ranges), ("logits", logits)):
        np.ascontiguousarray(a).tofile(os.path.join(d, fn + ".bin"))
    out = subprocess.run([HOST, "pipeline", d, str(B), str(S), str(T), str(C), str(r), str(blank), "0.2"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    rd = lambda fn, dt, shape: np.fromfile(os.path.join(d, fn + ".bin"), dtype=dt).reshape(shape)
    if C % 4 == 0:
        assert "builder=1" in out.stdout
        px, py = rd("builder_px", np.float32, g["simple_px"].shape), rd("builder_py", np.float32, g["simple_py"].shape)
        assert np.array_equal(np.isneginf(px), np.isneginf(g["simple_px"]))
        assert max_rel(px, g["simple_px"]) <= 1e-5 and max_rel(py, g["simple_py"]) <= 1e-5
        # the simple loss's gradient w.r.t. am / lm driven from C++ (W -> ftr_normalizer_gemm_f32 x2 -> epilogues) against the
        # package's own autograd node on the same inputs
        lm_t = torch.from_numpy(g["lm"]).to(dev).requires_grad_(True); am_t = torch.from_numpy(g["am"]).to(dev).requires_grad_(True)
        ft.rnnt_loss_simple(lm_t, am_t, torch.from_numpy(g["symbols"]).to(dev), blank, boundary=torch.from_numpy(g["boundary"]).to(dev),
                            reduction="sum").backward()
        assert max_rel(rd("simple_d_am", np.float32, g["am"].shape), am_t.grad.cpu().numpy()) <= 1e-5
        assert max_rel(rd("simple_d_lm", np.float32, g["lm"].shape), lm_t.grad.cpu().numpy()) <= 1e-5
    else:
        assert "builder=0" in out.stdout
    assert np.array_equal(rd("am_pruned", np.float32, (B, T, r, C)), am_p)
    assert np.array_equal(rd("lm_pruned", np.float32, (B, T, r, C)), lm_p)
    assert np.array_equal(rd("lm_pruned_only", np.float32, (B, T, r, C)), lm_p)
    np.testing.assert_allclose(-rd("pruned_ans", np.float32, (B,)).mean(), float(g[f"pruned_loss_r{r}"]), rtol=1e-4)
    assert max_rel(rd("logits_grad", np.float32, (B, T, r, C)), g[f"pruned_logits_grad_r{r}_f64"]) <= 2e-5
