"""Times the three batched f32 GEMMs of the simple-loss builder (c3 shapes) under the BLAS back ends torch offers."""
import torch, time
dev = torch.device("cuda:0")
B, S1, T, C = 32, 201, 1000, 500
g = torch.Generator(device="cpu").manual_seed(0)
lm = torch.rand((B, S1, C), generator=g).to(dev); am = torch.rand((B, T, C), generator=g).to(dev)
W = torch.rand((B, S1, T), generator=g).to(dev)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for lib in ("default", "hipblaslt", "hipblas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "unavailable", e); continue
    t1 = timeit(lambda: torch.bmm(lm, am.transpose(1, 2)))
    t2 = timeit(lambda: torch.bmm(W, am))
    t3 = timeit(lambda: torch.bmm(W.transpose(1, 2), lm))
    # alternative formulations: one big GEMM is impossible (per-utterance operands); try am @ lm^T then transpose
    t4 = timeit(lambda: torch.bmm(am, lm.transpose(1, 2)))
    print(f"{lib:10s} prod {t1:7.1f} us  dlmp {t2:7.1f} us  damp {t3:7.1f} us  (am@lm^T {t4:7.1f} us)   [6.43 GFLOP each]")
