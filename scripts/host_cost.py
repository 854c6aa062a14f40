"""Host-side cost of one step: enqueue time (no sync inside the loop) against GPU time, and a cProfile of the enqueue.
Usage: python scripts/host_cost.py [c2|c3] [steps]"""
import cProfile, pstats, sys, time, io, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
shapes = {"c2": (32, 512, 100, 500, 0), "c3": (32, 1000, 200, 500, 5)}
B, T, S, C, r = shapes[cfg]
dev = torch.device("cuda:0")
inp = bench.make_inputs(B, T, S, C, 0, dev)
step = (lambda: bench.simple_step(inp)) if r == 0 else (lambda: bench.pruned_step(inp, r))
for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{cfg}: enqueue {1e6 * (t1 - t0) / steps:.1f} us/step, with the final sync {1e6 * (t2 - t0) / steps:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:4000])
