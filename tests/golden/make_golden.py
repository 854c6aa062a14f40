"""Generates the committed golden fixtures (tests/golden/*.npz) from the CPU oracle.

The reference cannot be imported in the build container (ModuleNotFoundError: tensorflow; it would also
need its compiled CUDA op library), and its own tests hold no expected values (they print), so these
vectors are captured from oracle/ -- NOT from the reference: float parity is "unpinned by reference data"
(DESIGN.md).  Inputs follow the reference test's recipe (simple_rnnt_loss_test.py:260-289) and BASELINE
config c1.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def compute(inp, blank, s_ranges):
    import rnnt_oracle as O
    am, lm, sym, bd = inp["am"], inp["lm"], inp["symbols"], inp["boundary"]
    out = {}
    px, py = O.get_rnnt_logprobs(lm, am, sym, blank, "regular", bd)
    out["simple_px"], out["simple_py"] = px, py
    loss, (gx, gy) = O.rnnt_loss_simple(lm, am, sym, blank, bd, reduction="none", calc_gradients=True)
    out["simple_loss"], out["simple_px_grad"], out["simple_py_grad"] = loss, gx, gy
    # the same float32 px / py through the recursion in float64 (the *_f64 keys): the comparison point that carries no
    # float32 log-domain noise of its own (at T = 200 the float32 recursion is 3e-4 away from it)
    _, (gx64, gy64) = O.mutual_information_recursion(px, py, bd, True, np.float64)
    out["simple_px_grad_f64"], out["simple_py_grad_f64"] = gx64.astype(np.float32), gy64.astype(np.float32)
    sl, (sgx, sgy) = O.rnnt_loss_smoothed(lm, am, sym, blank, lm_only_scale=0.1, am_only_scale=0.2, boundary=bd,
                                          reduction="none", delay_penalty=0.2, calc_gradients=True)
    out["smoothed_loss"], out["smoothed_px_grad"], out["smoothed_py_grad"] = sl, sgx, sgy
    spx, spy = O.get_rnnt_logprobs_smoothed(lm, am, sym, blank, 0.1, 0.2, bd, "regular")
    spx = O._delay_penalty(spx, bd, "regular", 0.2)
    _, (sgx64, sgy64) = O.mutual_information_recursion(spx, spy, bd, True, np.float64)
    out["smoothed_px_grad_f64"], out["smoothed_py_grad_f64"] = sgx64.astype(np.float32), sgy64.astype(np.float32)
    for r in s_ranges:
        ranges = O.get_rnnt_prune_ranges(sgx, sgy, bd, r)
        out[f"ranges_r{r}"] = ranges
        am_p, lm_p = O.do_rnnt_pruning(am, lm, ranges)
        logits = (1.0 / (1.0 + np.exp(-(am_p + lm_p)))).astype(np.float32)
        pl, g = O.rnnt_loss_pruned_grad(logits, sym, ranges, blank, bd, delay_penalty=0.2, reduction="mean")
        out[f"pruned_loss_r{r}"] = np.asarray(pl, dtype=np.float32)
        out[f"pruned_logits_grad_r{r}"] = g
        _, g64 = O.rnnt_loss_pruned_grad(logits, sym, ranges, blank, bd, delay_penalty=0.2, reduction="mean", dtype=np.float64)
        out[f"pruned_logits_grad_r{r}_f64"] = g64
    return out


def main():
    from helpers import reference_test_recipe, synthetic
    cases = {
        "c1_B2_T8_S4_C16": (synthetic(2024, 2, 8, 4, 16), [2, 3, 5]),
        "seed1234_B2_T10_S7_C4": (reference_test_recipe(1234, 2, 10, 7, 4), [2, 3]),
        "seed12345_B2_T200_S50_C50": (reference_test_recipe(12345, 2, 200, 50, 50), [5]),
    }
    for name, (d, s_ranges) in cases.items():
        inp = {k: d[k] for k in ("am", "lm", "symbols", "boundary")}
        out = compute(inp, d["termination_symbol"], s_ranges)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), termination_symbol=np.int32(d["termination_symbol"]),
                            s_ranges=np.asarray(s_ranges, dtype=np.int32), **inp, **out)
        print(name, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
