"""The loss steps inside a hipGraph (torch.cuda.CUDAGraph): nothing in the package synchronises with the host, picks a
kernel from device data on the host, or leaves state behind in a workspace that a replay would trip over -- so a step
captured once replays with new input VALUES in the same buffers and gives what the eager step gives.  (bench.py reports the
replayed step time next to the eager one; launch-bound steps such as BASELINE config c2 gain a third.)"""
import numpy as np
import pytest
import torch

import bench

pytestmark = pytest.mark.gpu


def _capture(step):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):          # warm-up outside the capture: allocator pools, workspaces, the library-GEMM selection
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    torch.cuda.synchronize()
    return g, out


def _same(a, b, name):
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    assert a.shape == b.shape, name
    if a.dtype.kind in "iu":
        assert np.array_equal(a, b), name
    else:
        assert np.array_equal(np.isfinite(a), np.isfinite(b)), name
        fin = np.isfinite(b)
        if fin.any():   # the same kernels on the same values: bit for bit, up to the order of autograd's gradient accumulation
            assert np.abs(a[fin] - b[fin]).max() <= 1e-6 * max(1.0, np.abs(b[fin]).max()), name


@pytest.mark.parametrize("shape", [(3, 72, 20, 24, 5), (2, 900, 250, 16, 5)])   # the second: bands of the recursion chained over
@pytest.mark.parametrize("kind", ["simple", "pruned", "pruned_smoothed"])       # four workgroups, the band route in segments,
def test_step_replays_from_a_graph_with_new_values(ft, dev, kind, shape):       # the split column walk of prune_ranges
    B, T, S, C, r = shape
    inp = bench.make_inputs(B, T, S, C, 5, dev, ragged=True)
    if kind == "simple":
        step = lambda: bench.simple_step(inp, keep=True)
    else:
        step = lambda: bench.pruned_step(inp, r, keep=True, first_pass="smoothed" if kind == "pruned_smoothed" else "simple")
    g, out = _capture(step)
    for seed in (11, 12):
        fresh = bench.make_inputs(B, T, S, C, seed, dev, ragged=True)
        for k in ("am", "lm", "symbols", "boundary"):
            inp[k].copy_(fresh[k])                      # new values, same buffers
        g.replay()
        torch.cuda.synchronize()
        got = {k: v.clone() for k, v in out.items() if v is not None}
        ref = step()                                    # eager, same buffers
        torch.cuda.synchronize()
        for k, v in got.items():
            _same(v, ref[k], f"{kind} seed {seed}: {k}")


@pytest.mark.parametrize("kind", ["simple", "pruned", "pruned_smoothed"])
def test_step_is_bit_reproducible(ft, dev, kind):
    """The same step on the same inputs twice: every output bit for bit (no atomics in any reduction of the path, a fixed
    order in every scatter; the library GEMMs keep the kernel they have chosen by then)."""
    B, T, S, C, r = 3, 136, 40, 36, 5
    inp = bench.make_inputs(B, T, S, C, 9, dev, ragged=True)
    if kind == "simple":
        step = lambda: bench.simple_step(inp, keep=True)
    else:
        step = lambda: bench.pruned_step(inp, r, keep=True, first_pass="smoothed" if kind == "pruned_smoothed" else "simple")
    for _ in range(3):
        step()
    a = {k: v.detach().cpu().numpy().copy() for k, v in step().items() if v is not None}
    b = {k: v.detach().cpu().numpy().copy() for k, v in step().items() if v is not None}
    for k in a:
        assert a[k].tobytes() == b[k].tobytes(), f"{kind}: {k} differs between two runs"
