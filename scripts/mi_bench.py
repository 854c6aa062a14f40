"""Times the native mutual-information forward / backward alone (HIP events on the launch stream) over a few
lattice shapes.  Diagnostic helper for kernel tuning: python scripts/mi_bench.py [B S T ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-fast-rnnt_amd"))
import torch
import tf_fast_rnnt as ft
from tf_fast_rnnt import _lib
from tf_fast_rnnt.mutual_information import _ptr

def run(B, S, T, iters=20, cold=False, warm=3):
    dev = torch.device("cuda:0")
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(0)
    px = (torch.randn((B, S, T + 1), generator=g) - 6.0).to(dev)
    py = (torch.randn((B, S + 1, T), generator=g) - 6.0).to(dev)
    bd = torch.zeros((B, 4), dtype=torch.int32); bd[:, 2] = S; bd[:, 3] = T
    px[:, :, T] = float("-inf")
    bd = bd.to(dev)
    nws = L.ftr_mutual_information_workspace_floats(B, S, T)
    ws = torch.empty(nws, dtype=torch.float32, device=dev)
    _lib.call("ftr_mutual_information_workspace_init", _ptr(ws), nws, B, S, T, torch.cuda.current_stream().cuda_stream)
    product = not os.environ.get("FTR_BENCH_LEGACY_CALLS")   # what the package does: clean cached workspace, seed = ones
    pg = None
    ans = torch.empty(B, device=dev); ag = torch.ones(B, device=dev)
    gx = torch.empty_like(px); gy = torch.empty_like(py)
    st = torch.cuda.current_stream().cuda_stream
    flush = torch.empty(512 * 1024 * 1024 // 4, device=dev) if cold else None
    S1w = min(S, 130)
    px1 = torch.zeros((1, S1w, 65), device=dev) - 3.0; py1 = torch.zeros((1, S1w + 1, 64), device=dev) - 3.0
    bd1 = torch.tensor([[0, 0, S1w, 64]], dtype=torch.int32, device=dev); ans1 = torch.empty(1, device=dev)
    gx1 = torch.empty_like(px1); gy1 = torch.empty_like(py1)
    nws1 = L.ftr_mutual_information_workspace_floats(1, S1w, 64); ws1 = torch.empty(nws1, dtype=torch.float32, device=dev)
    dirty = torch.empty(int(os.environ.get("FTR_BENCH_DIRTY", "0")) * 1024 * 1024 // 4 + 1, device=dev)
    flags = int(os.environ.get("FTR_BENCH_FLAGS", "1"))   # 1 = FTR_MI_WS_CLEAN (what the package passes), 0 = memset per launch
    tf = tb = 0.0
    for i in range(iters + warm):
        if cold: flush.fill_(1.0)
        pre = os.environ.get("FTR_BENCH_PREWARM", "")
        if cold and "code" in pre:      # experiment: which cold resource costs the time?  pull the kernels' code back in with a tiny launch
            _lib.call("ftr_mutual_information_fwd_ws_f32", _ptr(px1), _ptr(py1), _ptr(bd1), _ptr(ws1), nws1, 0, _ptr(ans1), 1, S1w, 64, 0, st)
            _lib.call("ftr_mutual_information_bwd_ws_f32", _ptr(px1), _ptr(py1), _ptr(bd1), _ptr(ws1), nws1, 0, None, _ptr(gx1), _ptr(gy1), None, 0, 1, S1w, 64, 0, st)
        if cold and "ws" in pre: ws.copy_(ws)          # touch the whole workspace (read + write back)
        if cold and "in" in pre: px.copy_(px); py.copy_(py)
        if os.environ.get("FTR_BENCH_FRESH_OUT"):      # outputs from the allocator every iteration, as inside a training step
            gx = torch.empty_like(px); gy = torch.empty_like(py)
        if os.environ.get("FTR_BENCH_DIRTY"):          # a writer of DIRTY MB in front of the pair (what the builder kernel leaves behind)
            dirty.fill_(0.5)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        if product:
            _lib.call("ftr_mutual_information_fwd_ws_f32", _ptr(px), _ptr(py), _ptr(bd), _ptr(ws), nws, flags, _ptr(ans), B, S, T, 0, st)
        else:
            _lib.call("ftr_mutual_information_fwd_f32", _ptr(px), _ptr(py), _ptr(bd), _ptr(ws), _ptr(ans), B, S, T, 0, st)
        e[1].record()
        if product:
            _lib.call("ftr_mutual_information_bwd_ws_f32", _ptr(px), _ptr(py), _ptr(bd), _ptr(ws), nws, flags, None, _ptr(gx), _ptr(gy), None, 0, B, S, T, 0, st)
        else:
            _lib.call("ftr_mutual_information_bwd_f32", _ptr(px), _ptr(py), _ptr(bd), _ptr(ws), _ptr(pg), _ptr(gx), _ptr(gy), _ptr(ag), 1, B, S, T, 0, st)
        e[2].record()
        torch.cuda.synchronize()
        if i >= warm:
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    cells = B * (S + 1) * (T + 1)
    return 1e3 * tf / iters, 1e3 * tb / iters, cells

if __name__ == "__main__":
    shapes = [(32, 200, 1000), (32, 63, 1000), (32, 127, 1000), (8, 200, 1000), (64, 200, 1000), (256, 200, 1000), (32, 100, 512), (32, 300, 2000), (8, 1000, 8000)]
    if len(sys.argv) > 3:
        shapes = [tuple(int(v) for v in sys.argv[1:4])]
    for cold in (False, True):
        for (B, S, T) in shapes:
            f, b, cells = run(B, S, T, cold=cold)
            nslots = ((T + 1 + 63 + 15) // 16) + 5 * ((S + 1 + 63) // 64 - 1)
            print(f"B={B:4d} S={S:5d} T={T:5d} {'cold' if cold else 'warm'}: fwd {f:8.1f} us ({12*cells/f/1e3:7.1f} GB/s alg, {f/nslots*1e3:6.0f} ns/slot)   bwd {b:8.1f} us ({20*cells/b/1e3:7.1f} GB/s alg, {b/nslots*1e3:6.0f} ns/slot)", flush=True)
    import ctypes
    buf = (ctypes.c_ulonglong * 16)()
    if hasattr(_lib.lib(), "ftr_debug_stamps"):      # a diag-flavoured build (FTR_LIB_PATH=.../libftr_<variant>.so)
        _lib.lib().ftr_debug_stamps(buf)
    v = list(buf)
    if any(v):
        # duo kernels: [0..2] fwd compute wave (compute, barrier wait, slots), [3..7] fwd IO wave (park, drain, loads, barrier, slots)
        #              [8..10] bwd compute wave, [11..15] bwd IO wave
        for name, o in (("fwd", 0), ("bwd", 8)):
            n = max(v[o + 2], 1); m = max(v[o + 7], 1)
            print(f"{name} compute wave per slot: compute {v[o]/n:.0f}  barrier-wait {v[o+1]/n:.0f} cycles ({n} slots);  "
                  f"IO wave per slot: park {v[o+3]/m:.0f}  drain {v[o+4]/m:.0f}  loads {v[o+5]/m:.0f}  barrier-wait {v[o+6]/m:.0f} ({m} slots)")
