"""Randomised bit-exact comparison of ftr_prune_ranges_i32 with the CPU oracle (oracle/rnnt_oracle.get_rnnt_prune_ranges):
random occupancy-like inputs, ragged boundaries, every window length incl. the > 16 generic kernel, ties.
Test infrastructure (uses the oracle).  python scripts/prune_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tf-fast-rnnt_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import tf_fast_rnnt as ft
import rnnt_oracle as O


def main(n=100, seed=0):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    for it in range(n):
        B = int(rng.integers(1, 4)); S = int(rng.choice([1, 2, 3, 7, 16, 17, 33, 70, 130, 200, 331])); T = int(rng.choice([1, 2, 5, 63, 64, 65, 130, 200]))
        mod = bool(rng.integers(0, 2))
        if mod and S > T: S = T
        T1 = T if mod else T + 1
        r = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 24, S, S + 1, S + 3]))
        gx = rng.random((B, S, T1)).astype(np.float32); gy = rng.random((B, S + 1, T)).astype(np.float32)
        if rng.random() < 0.3:          # ties: few distinct values
            gx = np.round(gx * 3).astype(np.float32) / 3; gy = np.round(gy * 3).astype(np.float32) / 3
        bd = np.zeros((B, 4), np.int32); bd[:, 2] = S; bd[:, 3] = T
        for b in range(B):
            if rng.random() < 0.5:
                bd[b, 3] = int(rng.integers(max(1, T // 2), T + 1)); bd[b, 2] = int(rng.integers(max(1, min(S, bd[b, 3]) // 2), min(S, bd[b, 3]) + 1)) if mod else int(rng.integers(max(1, S // 2), S + 1))
        t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        got = ft.get_rnnt_prune_ranges(t_(gx), t_(gy), t_(bd), r).cpu().numpy()
        want = O.get_rnnt_prune_ranges(gx, gy, bd, r)
        assert got.shape == want.shape and np.array_equal(got, want), (it, B, S, T, mod, r, bd.tolist())
    print(f"{n} random cases: prune ranges bit-exact against the oracle")


if __name__ == "__main__":
    O.build()
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
