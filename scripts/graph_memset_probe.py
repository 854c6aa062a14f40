"""Does a hipMemsetAsync node captured into a hipGraph (torch.cuda.CUDAGraph) run on EVERY replay?  graph: memset(buf) ->
y = copy of buf -> buf := 7.  y must be all zeros after every replay."""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")
for n in (3, 1024, 100000, 1 << 22):
    buf = torch.full((n,), 5.0, device=dev)
    y = torch.empty_like(buf)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y.copy_(buf); buf.fill_(7.0)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, buf.numel() * 4, torch.cuda.current_stream().cuda_stream)
        y.copy_(buf)
        buf.fill_(7.0)
    torch.cuda.synchronize()
    res = []
    for i in range(3):
        g.replay(); torch.cuda.synchronize()
        res.append((float(y.abs().max()), float(buf.min())))
    print(f"n={n}: hipMemsetAsync rc {rc}; after each replay (max |y|, min buf): {res}")
