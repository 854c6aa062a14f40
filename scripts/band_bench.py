"""The band recursion alone (ftr_mutual_information_band_ws_f32) on synthetic band arrays: chain kernels (mi_band.hip) against
the segmented route (mi_band_seg.hip), per-kernel times with rocprofv3 if run under it.  python scripts/band_bench.py [c3|c4|c5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import torch
import bench
import tf_fast_rnnt as ft
from tf_fast_rnnt import _lib
from tf_fast_rnnt.mutual_information import _ptr

dev = torch.device("cuda:0")
SHAPES = dict(bench.CONFIGS, m1=(32, 1500, 300, 500, 5), m2=(32, 700, 150, 500, 5), m3=(32, 400, 80, 500, 5))
for cfg in (sys.argv[1:] or ["c3", "c4", "c5"]):
    B, T, S, C, r = SHAPES[cfg]
    inp = bench.make_inputs(B, T, S, C, 1234, dev)
    _, (gx0, gy0) = ft.rnnt_loss_simple(lm=inp["lm"], am=inp["am"], symbols=inp["symbols"], termination_symbol=inp["blank"],
                                        boundary=inp["boundary"], reduction="sum", calc_gradients=True)
    ranges = ft.get_rnnt_prune_ranges(gx0, gy0, inp["boundary"], r)
    g = torch.Generator(device="cpu").manual_seed(1)
    pxb = (torch.randn((B, T, r), generator=g) - 3.0).to(dev); pyb = (torch.randn((B, T, r), generator=g) - 1.0).to(dev)
    bd = inp["boundary"]
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    impls = ("chain", "segments")
    for impl in impls:
        os.environ["FTR_BAND_IMPL"] = impl
        nws = int(_lib.lib().ftr_mutual_information_band_workspace_floats(B, T, S, r))
        ws = torch.empty(max(nws, 4), device=dev)
        ans = torch.empty(B, device=dev); gx = torch.empty((B, T, r), device=dev); gy = torch.empty((B, T, r), device=dev)
        run = lambda: _lib.call("ftr_mutual_information_band_ws_f32", _ptr(pxb), _ptr(pyb), _ptr(ranges), _ptr(bd), _ptr(ws), nws,
                                _ptr(ans), _ptr(gx), _ptr(gy), B, T, S, r, 0, st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res[impl] = (e0.elapsed_time(e1) * 50, ans.clone(), gx.clone(), gy.clone(), nws)
    (tc, ac, gxc, gyc, wc), (ts, as_, gxs, gys, wss) = res["chain"], res["segments"]
    print(f"{cfg} (B={B} T={T} S={S} r={r}): chain {tc:.1f} us (workspace {wc * 4 / 1e6:.1f} MB)   segments {ts:.1f} us ({wss * 4 / 1e6:.1f} MB)   "
          f"|d ans| {float((ac - as_).abs().max()):.2e} of {float(ac.abs().max()):.1f}   max |d gx| {float((gxc - gxs).abs().max()):.2e}  |d gy| {float((gyc - gys).abs().max()):.2e}")
os.environ.pop("FTR_BAND_IMPL", None)
