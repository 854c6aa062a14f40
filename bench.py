#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native pruned RNN-T loss.

One "step" = one pass of the whole hot path over one batch of synthetic utterances (BASELINE.json
config "rnnt_loss_pruned s_range=5 B=32 T=1000 S=200 C=500", SURVEY.md 8d "c3"), inputs resident in HBM:

  rnnt_loss_simple(calc_gradients=True)          px/py builder + recursion forward + backward (occupancies)
  get_rnnt_prune_ranges(s_range=5)               prune ranges (int32)
  do_rnnt_pruning                                prune gather -> am_pruned, lm_pruned [B,T,5,C]
  logits = sigmoid(am_pruned + lm_pruned)        joiner stand-in of the reference's own test
                                                 (tf_fast_rnnt/python/tests/simple_rnnt_loss_test.py:345-348)
  rnnt_loss_pruned                               logsumexp + band->lattice + recursion forward + backward
  (0.5 * simple + pruned).backward()             d/d logits (native), then torch autograd back to am, lm

With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank processes its own batch of
B utterances (the batch dimension shards with no data-path collective; weak scaling) and the scalar loss
is all-reduced once per step over RCCL.  Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "tf-fast-rnnt_amd"),):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

CONFIGS = {
    # name: (B, T, S, C, s_range)
    "c2": (32, 512, 100, 500, 5),
    "c3": (32, 1000, 200, 500, 5),
    "c4": (32, 2000, 300, 1024, 5),     # per-GPU share of B=256 over 8 GPUs
    "c5": (8, 8000, 1000, 512, 10),
}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def make_inputs(B, T, S, C, seed, device, ragged=False):
    """SURVEY.md 8d synthetic inputs: am, lm ~ N(0,1), symbols ~ U{0..C-2}, blank = C-1, full boundary."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    am = torch.randn((B, T, C), generator=g, dtype=torch.float32).to(device)
    lm = torch.randn((B, S + 1, C), generator=g, dtype=torch.float32).to(device)
    symbols = torch.randint(0, C - 1, (B, S), generator=g, dtype=torch.int32).to(device)
    boundary = torch.zeros((B, 4), dtype=torch.int32)
    boundary[:, 2] = S
    boundary[:, 3] = T
    if ragged:
        t_end = torch.randint((T + 1) // 2, T + 1, (B,), generator=g)
        t_end[0] = T
        s_end = torch.minimum(torch.randint((S + 1) // 2, S + 1, (B,), generator=g), t_end)
        s_end[0] = S
        boundary[:, 2] = s_end.to(torch.int32)
        boundary[:, 3] = t_end.to(torch.int32)
    return dict(am=am, lm=lm, symbols=symbols, boundary=boundary.to(device), blank=C - 1, B=B, T=T, S=S, C=C)


def pruned_step(inp, s_range, keep=False, simple_loss_scale=0.5, first_pass="simple", process_group=None, timer=None,
                dense_am_pruned=False):
    """One full step of the hot path (see module docstring).  Returns the scalar loss (and internals).
    first_pass = "smoothed" uses rnnt_loss_smoothed (lm_only_scale 0.1, am_only_scale 0.2 as in
    simple_rnnt_loss_test.py:291-336) for the occupancy pass (BASELINE.json configs[3], "c4")."""
    import tf_fast_rnnt as ft
    am = inp["am"].detach().requires_grad_(True)
    lm = inp["lm"].detach().requires_grad_(True)
    sym, bd, blank = inp["symbols"], inp["boundary"], inp["blank"]
    if first_pass == "smoothed":
        simple_loss, (px_grad, py_grad) = ft.rnnt_loss_smoothed(lm=lm, am=am, symbols=sym, termination_symbol=blank,
                                                               lm_only_scale=0.1, am_only_scale=0.2, boundary=bd,
                                                               reduction="sum", calc_gradients=True,
                                                               process_group=process_group)
    else:
        simple_loss, (px_grad, py_grad) = ft.rnnt_loss_simple(lm=lm, am=am, symbols=sym, termination_symbol=blank,
                                                             boundary=bd, reduction="sum", calc_gradients=True)
    ranges = ft.get_rnnt_prune_ranges(px_grad=px_grad, py_grad=py_grad, boundary=bd, s_range=s_range)
    am_p, lm_p = ft.do_rnnt_pruning(am=am, lm=lm, ranges=ranges, dense=dense_am_pruned)
    if timer is not None and timer.enabled:       # the joiner stand-in is user code, timed apart from the loss
        with timer("joiner_standin_fwd"):
            logits = torch.sigmoid(am_p + lm_p)
        ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
        logits.register_hook(lambda g: (ev[0].record(torch.cuda.current_stream()), None)[1])
        am_p.register_hook(lambda g: (ev[1].record(torch.cuda.current_stream()), None)[1])
        timer.records.setdefault("joiner_standin_bwd", []).append(tuple(ev))
    else:
        logits = torch.sigmoid(am_p + lm_p)
    if keep:
        logits.retain_grad()
    pruned_loss = ft.rnnt_loss_pruned(logits=logits, symbols=sym, ranges=ranges, termination_symbol=blank,
                                      boundary=bd, reduction="sum")
    loss = simple_loss_scale * simple_loss + pruned_loss
    loss.backward()
    if keep:
        return dict(loss=loss.detach(), simple_loss=simple_loss.detach(), pruned_loss=pruned_loss.detach(),
                    ranges=ranges, px_grad=px_grad, py_grad=py_grad, logits_grad=logits.grad,
                    am_grad=am.grad, lm_grad=lm.grad)
    return loss.detach()


def simple_step(inp, keep=False, timer=None):
    """BASELINE.json configs[1] ("c2"): rnnt_loss_simple forward + backward only -- px/py builder, recursion forward and
    backward (occupancies), and the builder's backward to d am, d lm."""
    import tf_fast_rnnt as ft
    am = inp["am"].detach().requires_grad_(True)
    lm = inp["lm"].detach().requires_grad_(True)
    loss, (px_grad, py_grad) = ft.rnnt_loss_simple(lm=lm, am=am, symbols=inp["symbols"], termination_symbol=inp["blank"],
                                                   boundary=inp["boundary"], reduction="sum", calc_gradients=True)
    loss.backward()
    if keep:
        return dict(loss=loss.detach(), px_grad=px_grad, py_grad=py_grad, am_grad=am.grad, lm_grad=lm.grad)
    return loss.detach()


# ------------------------------------------------------------------------------------------------ profiling
class CallTimer:
    """Brackets every native C-ABI call with two HIP events recorded on the stream the kernels are
    launched on (torch's current stream: the package passes exactly that stream to the C ABI)."""

    def __init__(self):
        self.records = {}
        self.enabled = False

    @contextlib.contextmanager
    def __call__(self, name):
        if not self.enabled:
            yield
            return
        st = torch.cuda.current_stream()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        try:
            yield
        finally:
            e1.record(st)
            self.records.setdefault(name, []).append((e0, e1))

    def summary(self):
        out = {}
        for name, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = dict(calls=len(ms), avg_us=1e3 * float(np.mean(ms)), total_ms=float(np.sum(ms)))
        return out


# native call -> the kernels it launches (names as rocprofv3 prints them, namespace stripped), for the PMC traffic
CALL_KERNELS = {
    "ftr_mutual_information_fwd_ws_f32": ["mi_bidir_fwd_kernel<false>"],
    "ftr_mutual_information_bwd_ws_f32": ["mi_bidir_flow_kernel<false>"],
    "ftr_mutual_information_bwd_loss_ws_f32": ["mi_bidir_flow_kernel<false>"],
    "ftr_prune_ranges_i32": ["prune_argmax_once_kernel<5>", "prune_adjust_kernel"],
    "ftr_do_pruning_f32": ["do_pruning_kernel<true>"],
    "ftr_do_pruning_bwd_f32": ["do_pruning_bwd_am_kernel<true>", "do_pruning_bwd_lm_kernel<true>"],
    "ftr_do_pruning_bwd_ws_f32": ["do_pruning_bwd_seg_kernel<5, true>", "do_pruning_bwd_reduce_kernel"],
    "ftr_pruned_logprobs_fwd_f32": ["lse_rows_reg_kernel<2>", "band_to_lattice_kernel<false>"],
    "ftr_pruned_band_fwd_f32": ["lse_rows_reg_kernel<2>", "band_gather_kernel<false>"],
    "ftr_mutual_information_band_f32": ["mi_band_kernel<false, 8>"],
    "ftr_mutual_information_band_ws_f32": ["band_seg_init_kernel<false, 8>", "band_seg_scatter_kernel<false, 8>", "band_seg_transfer_kernel<false, 8>",
                                           "band_seg_prefix_kernel<false, 8>", "band_seg_final_kernel<false, 8>", "band_seg_occupancy_kernel<false, 8>"],
    "ftr_pruned_band_bwd_scaled_f32": ["band_grad_banded_kernel<true>"],
    "ftr_pruned_logprobs_bwd_f32": ["band_grad_kernel<false, true>"],
    "ftr_pruned_logprobs_bwd_scaled_f32": ["band_grad_kernel<false, true>"],
    "ftr_simple_logprobs_bwd_w_scaled_f32": ["simple_bwd_w_kernel<false>"],
    "ftr_simple_logprobs_bwd_am_scaled_f32": ["simple_bwd_am_kernel<false, 16>"],
    "ftr_rowmax_exp_f32": ["rowmax_exp_kernel<true>"],
    "ftr_rowmax_exp_pair_f32": ["rowmax_exp_pair_kernel<true>"],
    "ftr_simple_logprobs_fwd_f32": ["simple_fwd_kernel<false, 16>"],
    "ftr_simple_logprobs_fused_fwd_f32": ["simple_fused_fwd_kernel<false, false, 13>"],
    "ftr_simple_logprobs_fused_bwd_am_f32": ["simple_fused_bwd_am_kernel<false>"],
    "ftr_simple_logprobs_bwd_w_f32": ["simple_bwd_w_kernel<false>"],
    "ftr_simple_logprobs_bwd_am_f32": ["simple_bwd_am_kernel<false>"],
    "ftr_simple_logprobs_bwd_lm_f32": ["simple_bwd_lm_kernel"],
}


def pmc_traffic(config):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/r*_pmc_traffic.json,
    made by scripts/summarize_pmc.py: FETCH_SIZE and WRITE_SIZE in separate runs, KiB units, FETCH_SIZE doubled as the
    MI355X guide prescribes for gfx950).  Counters cannot be read from inside the timed process, so the bench
    line carries the last committed measurement and names its file; None if there is none for this config."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{config}.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        per_kernel = json.load(f)
    out = {}
    for call, kernels in CALL_KERNELS.items():
        if all(k in per_kernel for k in kernels):
            out[call] = sum(per_kernel[k]["hbm_bytes"] for k in kernels)
    return out, os.path.relpath(files[-1], ROOT)


def algorithmic_bytes(B, T, S, C, r):
    """SURVEY.md 8(d) algorithmic bytes per native call (f32 = 4 B).  L = lattice cells."""
    L = B * (S + 1) * (T + 1)
    npx, npy = B * S * (T + 1), B * (S + 1) * T
    N = 4 * B * T * r * C
    nam, nlm = B * T * C, B * (S + 1) * C
    return {
        # simple/smoothed builder ("report bytes and flops separately; it is not the headline")
        "ftr_rowmax_exp_f32": 4 * (nam + nlm),                              # mean of the am call and the lm call: read + write
        "ftr_rowmax_exp_pair_f32": 8 * (nam + nlm),                         # am and lm in one launch: read + write
        "ftr_rowmax_exp_sum_f32": 4 * 2 * nlm,
        "ftr_rowmax_exp_dot_f32": 4 * 2 * nam,                              # am read, am_probs written (+ the dot for free)
        "ftr_colsum_weighted_f32": 4 * (nam + nlm) // 2,                    # mean of the lm_probs call (fwd) and the am_probs call (bwd)
        "ftr_rowdot_f32": 4 * nlm,
        "ftr_simple_logprobs_fwd_f32": 4 * (nam + nlm + npy + npx + npy),   # read am, lm, prod; write px, py
        "ftr_smoothed_logprobs_fwd_f32": 4 * (nam + nlm + npy + npx + npy),
        # fused builder: am_probs, lm_probs (+ the am / lm gathers) in, px, py and the product for the backward out;
        # 2 * B * (S+1) * T * C flops on the f32 MFMA pipe (reported apart)
        "ftr_simple_logprobs_fused_fwd_f32": 4 * (2 * nam + 2 * nlm + npx + 2 * npy),
        "ftr_smoothed_logprobs_fused_fwd_f32": 4 * (2 * nam + 2 * nlm + npx + 2 * npy),
        "ftr_simple_logprobs_bwd_w_f32": 4 * (npx + 3 * npy),               # read gpx, gpy, prod; write W
        "ftr_simple_logprobs_bwd_w_scaled_f32": 4 * (npx + 3 * npy),
        "ftr_simple_logprobs_bwd_am_scaled_f32": 4 * (npx + npy + 3 * nam),
        "ftr_pruned_logprobs_bwd_scaled_f32": 2 * N + 4 * (npx + npy),
        "ftr_smoothed_logprobs_bwd_w_f32": 4 * (npx + 3 * npy),
        "ftr_simple_logprobs_bwd_am_f32": 4 * (npx + npy + 3 * nam),        # read gpx, gpy, damp, am_probs; write d am
        "ftr_smoothed_logprobs_bwd_am_f32": 4 * (npx + npy + 3 * nam),
        "ftr_simple_logprobs_bwd_lm_f32": 4 * 3 * nlm,
        # fused d am: g_px, g_py, prod, lm_probs, am_probs in, d am out (g_px read again by the scatter pass)
        "ftr_simple_logprobs_fused_bwd_am_f32": 4 * (2 * npx + 2 * npy + nlm + 2 * nam),
        "ftr_smoothed_logprobs_fused_bwd_am_f32": 4 * (2 * npx + 2 * npy + nlm + 2 * nam),
        "ftr_smoothed_logprobs_bwd_lm_f32": 4 * 3 * nlm,
        "ftr_do_pruning_bwd_f32": 2 * N + 4 * (nam + nlm + B * T * r),      # read both pruned gradients, write d am, d lm
        "ftr_do_pruning_bwd_ws_f32": N + 4 * (nam + nlm + B * T * r),       # the joiner's gradient is ONE tensor: read once
        # fwd reads px,py and writes p; bwd (reference algorithm) reads px,py,p and writes both grads: 32 L total
        "ftr_mutual_information_fwd_ws_f32": 4 * (npx + npy + L),
        "ftr_mutual_information_bwd_ws_f32": 4 * (npx + npy + L + npx + npy),
        "ftr_mutual_information_bwd_loss_ws_f32": 4 * (npx + npy + L + npx + npy),   # the same launch, the loss tail rides along
        "ftr_prune_ranges_i32": 4 * (npx + npy + B * T * r),
        "ftr_do_pruning_f32": 4 * (B * (S + 1) * C + B * T * r) + N,     # the gather; am_pruned stays a broadcast view of am
        "ftr_pruned_logprobs_fwd_f32": N + 4 * (npx + npy),              # stream logits once, write px,py
        # the band path of rnnt_loss_pruned (3 N + O(B T r) for the pruned loss with a banded DP)
        "ftr_pruned_band_fwd_f32": N + 4 * 3 * B * T * r,                # stream logits once; lse, px_band, py_band
        "ftr_mutual_information_band_f32": 4 * 5 * B * T * r,            # band in, occupancies out (lives in LDS in between)
        "ftr_mutual_information_band_ws_f32": 4 * 5 * B * T * r,
        "ftr_pruned_band_bwd_scaled_f32": 2 * N + 4 * 3 * B * T * r,     # re-read logits, write the gradient
        "ftr_pruned_logprobs_bwd_f32": 2 * N + 4 * (npx + npy),          # re-read logits, write the gradient
    }


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_pipeline_worker(job):
    """One worker of the CPU baseline: the oracle's whole loss pipeline on batches of `sample_B` utterances of the
    workload, single-threaded, repeated until `seconds` have passed.  Returns (utterances done, seconds)."""
    B, T, S, C, r, sample_B, seed, seconds, simple_only = job
    os.environ["OMP_NUM_THREADS"] = "1"
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rnnt_oracle as O
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:
        limiter = contextlib.nullcontext()
    O.build()
    rng = np.random.default_rng(seed)
    am = rng.standard_normal((sample_B, T, C)).astype(np.float32)
    lm = rng.standard_normal((sample_B, S + 1, C)).astype(np.float32)
    sym = rng.integers(0, C - 1, (sample_B, S)).astype(np.int32)
    bd = np.zeros((sample_B, 4), np.int32); bd[:, 2] = S; bd[:, 3] = T
    done = 0
    with limiter:
        t0 = time.perf_counter()
        while True:
            _, (gx, gy) = O.rnnt_loss_simple(lm, am, sym, C - 1, bd, reduction="sum", calc_gradients=True)
            if simple_only:
                done += sample_B
                dt = time.perf_counter() - t0
                if dt >= seconds or done >= 4096:
                    break
                continue
            ranges = O.get_rnnt_prune_ranges(gx, gy, bd, r)
            am_p, lm_p = O.do_rnnt_pruning(am, lm, ranges)
            logits = (1.0 / (1.0 + np.exp(-(am_p + lm_p)))).astype(np.float32)
            O.rnnt_loss_pruned_grad(logits, sym, ranges, C - 1, bd, reduction="sum")
            done += sample_B
            dt = time.perf_counter() - t0
            if dt >= seconds or done >= 4096:
                break
    return done, dt


def _host_description():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return dict(nproc=os.cpu_count(), usable_cores=usable, cpu_model=model)


def cpu_baseline(B, T, S, C, r, sample_B=4, seed=0, seconds=9.0, max_workers=16, simple_only=False):
    """SURVEY.md 8(d) CPU baseline: the oracle (CPU port of the same path: px/py builder + recursion fwd+bwd + ranges +
    gather + sigmoid + pruned log-probs + recursion fwd+bwd + gradient w.r.t. logits; simple_only: the first of these)
    timed on this host, (i) on one thread and (ii) on `cores` single-threaded worker processes, parallel over the batch
    dimension, each on its own utterances.  cores = min(usable cores, 16): 16 is the CPU share of one GPU on the pool's
    boxes (a box reports 256 cores for 8 GPUs), and what `cpu_baseline.cores` states is the number actually used.
    `value` is the multi-worker rate; the single-thread rate is reported beside it."""
    import multiprocessing as mp
    host = _host_description()
    one_done, one_dt = _cpu_pipeline_worker((B, T, S, C, r, sample_B, seed, seconds, simple_only))
    single = one_done / one_dt
    workers = max(1, min(host["usable_cores"], max_workers))
    multi = None
    if workers > 1:
        ctx = mp.get_context("spawn")     # fresh interpreters: the parent holds a HIP context
        with ctx.Pool(workers) as pool:
            t0 = time.perf_counter()
            res = pool.map(_cpu_pipeline_worker, [(B, T, S, C, r, sample_B, seed + 1 + i, seconds, simple_only) for i in range(workers)])
            wall = time.perf_counter() - t0
        # every worker runs for >= `seconds` after its own start-up; rate = sum of the workers' own rates
        multi = sum(d / t for d, t in res)
        multi_done = sum(d for d, _ in res)
    value = multi if multi is not None else single
    return dict(value=round(value, 3), unit="utterances/s", cores=workers, kind="port",
                single_thread_value=round(single, 3), host=host,
                sample=("the oracle's rnnt_loss_simple with occupancies (oracle/: numpy builder + C recursion forward + backward, "
                        "without the builder's backward to am/lm)" if simple_only else
                        "the oracle's whole loss pipeline (oracle/: C recursion + numpy builders; forward + gradients w.r.t. "
                        "px/py and pruned logits, without the autograd tail to am/lm)") + f" on batches of {sample_B} utterances of "
                       f"the same workload: 1 thread {one_done} utterances in {one_dt:.1f} s; "
                       + (f"{workers} single-threaded worker processes, {multi_done} utterances, {wall:.1f} s wall incl. start-up"
                          if multi is not None else "one usable core only"))


# ------------------------------------------------------------------------------------------------ hipGraph replay
def graph_replay(step_fn, steps):
    """The same step captured once into a hipGraph (torch.cuda.CUDAGraph: every native call, memset node and library
    GEMM of the step lands in the graph -- the package never synchronises with the host) and replayed `steps` times.
    Reported next to the eager number; `value` stays the eager one (per-call HIP events cannot live inside a graph)."""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = step_fn()
        torch.cuda.synchronize()
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dict(ms_per_step=round(1e3 * dt / steps, 4), loss=float(out.item()))
    except Exception as e:   # capture not possible in this environment: report why, do not fail the bench line
        return dict(ms_per_step=None, error=f"{type(e).__name__}: {e}"[:300])


# ------------------------------------------------------------------------------------------------ self-launch
def visible_gpu_count() -> int:
    """GPUs this process could use, counted WITHOUT any HIP / HSA call (the launching parent must never initialise the
    GPU): the KFD topology in sysfs lists one node per agent, GPU nodes have simd_count > 0; ROCR_VISIBLE_DEVICES /
    HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES restrict it the way the runtime would."""
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split(None, 1) for line in open(path).read().splitlines() if " " in line)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N rank processes (fresh children, one device each, RCCL),
    rank 0 prints the JSON line.  The parent never touches the GPU (devices are counted from sysfs, visible_gpu_count)
    and never re-execs itself.  Returns the exit code."""
    import socket
    import subprocess
    n = args.gpus
    forced = "FTR_BENCH_FORCE_DEVICE" in os.environ      # rehearsal: every rank on one card (gloo backend)
    have = visible_gpu_count()
    if not forced and have < n:
        print(f"bench.py: --gpus {n} asked for but only {have} HIP device(s) are visible; not measuring fewer GPUs "
              f"under an n_gpus={n} label", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    if rc:
        print(f"bench.py: a rank exited with code {rc}", file=sys.stderr)
    return rc


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="skip the extra hipGraph-replay measurement")
    ap.add_argument("--no-dense", action="store_true",
                    help="skip the secondary dense-am_pruned steps (profiling runs: keeps per-kernel averages to the timed path)")
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="leave rocBLAS' default kernel choice for the normaliser GEMMs (FTR_GEMM_TUNE=off)")
    ap.add_argument("--gemm-choices", default=None, help="file the library-GEMM kernel choices are stored in / reloaded from")
    ap.add_argument("--no-gemm-search", action="store_true", help="only apply the choices already in --gemm-choices (profiling runs)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="record the per-call HIP events on every n-th timed step (1 = every step)")
    ap.add_argument("--pass", dest="which", default=None, choices=["pipeline", "simple"],
                    help="pipeline = the whole pruned step (default); simple = rnnt_loss_simple forward + backward only "
                         "(BASELINE.json configs[1]; the default for --config c2)")
    ap.add_argument("--first-pass", default=None, choices=["simple", "smoothed"],
                    help="occupancy pass; default simple, smoothed for c4 (BASELINE.json configs[3])")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))      # before this process makes any GPU call
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
                         f"refusing to print a line whose n_gpus would not be what was asked for")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (never set by the driver): several ranks on ONE card over gloo exercise the N > 1 control flow
    # where no multi-GPU node is at hand
    if "FTR_BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["FTR_BENCH_FORCE_DEVICE"])
    backend = os.environ.get("FTR_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)

    import tf_fast_rnnt as ft
    B, T, S, C, r = CONFIGS[args.config]
    inp = make_inputs(B, T, S, C, seed=1000 + rank, device=dev, ragged=args.ragged)

    timer = CallTimer()
    first_pass = args.first_pass or ("smoothed" if args.config == "c4" else "simple")
    which = args.which or ("simple" if args.config == "c2" else "pipeline")
    group = dist.group.WORLD if (dist is not None and first_pass == "smoothed") else None

    def step(dense=False):
        if which == "simple":
            loss = simple_step(inp, timer=timer)
        else:
            loss = pruned_step(inp, r, first_pass=first_pass, process_group=group, timer=timer, dense_am_pruned=dense)
        if dist is not None:
            dist.all_reduce(loss)           # the single scalar exchange of the sharded loss
        return loss

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ft._lib.set_profile_hook(timer)
    # library-GEMM kernel selection (package feature, csrc/normalizer_gemm.hip): the library measures rocBLAS' candidates at
    # the second call with a shape, i.e. inside the untimed warm-up (two extra untimed steps if W < 2); --gemm-choices FILE
    # stores what was chosen / reapplies it (with --no-gemm-search nothing is measured: profiling runs)
    Bc, Tc, Sc, Cc = inp["B"], inp["T"], inp["S"], inp["C"]
    if args.no_gemm_tuning:
        os.environ["FTR_GEMM_TUNE"] = "off"
    else:
        if args.gemm_choices and os.path.exists(args.gemm_choices):
            with open(args.gemm_choices) as f:
                for key, rec in json.load(f).items():
                    kind, b_, t_, s_, c_ = (int(v) for v in key.split(","))
                    if (b_, t_, s_, c_) == (Bc, Tc, Sc, Cc):
                        ft.set_normalizer_gemm_choice(kind, b_, t_, s_, c_, int(rec["solution"]))
        os.environ["FTR_GEMM_TUNE"] = "off" if args.no_gemm_search else "second"
        for _ in range(max(0, 2 - args.warmup)):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.reset_peak_memory_stats(dev)
    # per-call HIP events are recorded inside the timed region on every `event_every`-th step (rank 0): ~60 event
    # records per step are not free, and sampling keeps the timed region what a training loop would run
    sampled = [i for i in range(args.steps) if i % max(args.event_every, 1) == 0]
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        timer.enabled = (rank == 0) and (i % max(args.event_every, 1) == 0)
        last = step()
    fence()
    dt = time.perf_counter() - t0
    timer.enabled = False
    ft._lib.set_profile_hook(None)
    peak_mb = torch.cuda.max_memory_allocated(dev) / 2**20
    gemm_choices = {}
    for kind in range(3):
        ch = ft.normalizer_gemm_choice(kind, Bc, Tc, Sc, Cc)
        if ch is not None:
            gemm_choices[f"{kind},{Bc},{Tc},{Sc},{Cc}"] = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ch.items()}
    if rank == 0 and args.gemm_choices and not args.no_gemm_tuning and not args.no_gemm_search:
        with open(args.gemm_choices, "w") as f:
            json.dump(gemm_choices, f, indent=1)
    # secondary: the same step with am_pruned MATERIALISED ([B,T,r,C] written by the gather and read by the joiner), which is
    # what the reference's tf.broadcast_to costs and what a TensorFlow binding of the C ABI pays; untimed for `value`
    dense_ms = None
    if which == "pipeline" and world == 1 and not args.no_dense:
        for _ in range(2):
            step(dense=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(max(args.steps // 2, 1)):
            step(dense=True)
        torch.cuda.synchronize()
        dense_ms = 1e3 * (time.perf_counter() - t1) / max(args.steps // 2, 1)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * dt / args.steps
    value = world * B * args.steps / dt
    calls = timer.summary()
    alg = algorithmic_bytes(B, T, S, C, r)
    kernels = {}
    for name, rec in calls.items():
        per_step_calls = rec["calls"] / len(sampled)
        kernels[name] = dict(avg_us=round(rec["avg_us"], 2), calls_per_step=per_step_calls,
                             algorithmic_MB=round(alg[name] / 1e6, 1) if name in alg else None,
                             GBps=round(alg[name] / (rec["avg_us"] * 1e-6) / 1e9, 1) if name in alg else None)
    # roofline: always the kernel pair north_star names, the mutual-information recursion forward + backward (SURVEY.md
    # 8(d): 32 bytes per lattice cell for the pair), whatever else is slow in the step -- so that the line is comparable
    # from run to run; every other native call is in `kernels`.  avg_launch_us = the HIP-event time of the two calls
    # (each call = its kernels back to back on the stream; the rocprofv3 --kernel-trace --stats summary of this same
    # command under profiles/ gives the per-kernel split).
    # the entry points the package calls (the loss nodes' backward launch also writes the reduced loss: ..._bwd_loss_ws_f32)
    pair = ("ftr_mutual_information_fwd_ws_f32",
            "ftr_mutual_information_bwd_loss_ws_f32" if "ftr_mutual_information_bwd_loss_ws_f32" in calls else "ftr_mutual_information_bwd_ws_f32")
    roofline = None
    traffic, traffic_file = pmc_traffic(args.config) if (first_pass == "simple" and not args.ragged) else (None, None)
    if traffic:
        for name in kernels:
            kernels[name]["pmc_hbm_MB"] = round(traffic[name] / 1e6, 1) if name in traffic else None
    if all(n in calls for n in pair):
        pair_us = sum(calls[n]["avg_us"] for n in pair)
        pair_bytes = sum(alg[n] for n in pair)
        achieved = pair_bytes / (pair_us * 1e-6) / 1e9
        roofline = dict(bound="hbm", kernel="+".join(pair), achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4),
                        traffic=(sum(traffic[n] for n in pair) if traffic and all(n in traffic for n in pair) else None),
                        traffic_source=traffic_file, avg_launch_us=round(pair_us, 2),
                        avg_launch_us_each={n: round(calls[n]["avg_us"], 2) for n in pair},
                        algorithmic_bytes=pair_bytes, launches_per_step=calls[pair[0]]["calls"] / len(sampled))
    # the streaming share of the step against the same roofline: all native calls together
    tot_alg = sum(alg[n] * calls[n]["calls"] for n in calls if n in alg) / len(sampled)
    tot_us = sum(calls[n]["total_ms"] for n in calls if n in alg) * 1e3 / len(sampled)
    joiner_us = sum(calls[n]["total_ms"] for n in calls if n.startswith("joiner_standin")) * 1e3 / len(sampled)
    native_us = sum(rec["total_ms"] for n, rec in calls.items() if not n.startswith("joiner_standin")) * 1e3 / len(sampled)
    if which == "simple":
        workload = (f"{args.config}: rnnt_loss_simple fwd+bwd (builder + recursion forward/backward + builder backward to am, lm), "
                    f"B={B}/GPU T={T} S={S} C={C}, regular, {'ragged' if args.ragged else 'full'} boundary")
    else:
        workload = (f"{args.config}: rnnt_loss_pruned fwd+bwd step, B={B}/GPU T={T} S={S} C={C} s_range={r}, "
                    f"regular, {'ragged' if args.ragged else 'full'} boundary, first pass rnnt_loss_{first_pass}")
    out = {
        "metric": "rnnt_loss_simple_fwd_bwd_throughput" if which == "simple" else "rnnt_loss_pruned_fwd_bwd_throughput",
        "value": round(value, 2),
        "unit": "utterances/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "us_per_step": round(1e3 * ms_per_step, 1),
        "peak_mem_mb": round(peak_mb, 1),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload,
                   "global_batch": world * B, "sharding": f"batch x{world}, one scalar all-reduce/step",
                   # do_rnnt_pruning returns am_pruned as a stride-0 broadcast view of am (values, shape and gradient as the
                   # reference's tf.broadcast_to; the reference's joiner stand-in consumes it by broadcasting)
                   "am_pruned": "broadcast view"},
        "roofline": roofline,
        "kernel_timing": f"HIP events around every native call on {len(sampled)} of the {args.steps} timed steps",
        "native_us_per_step": round(native_us, 1),
        # the joiner stand-in sigmoid(am_pruned + lm_pruned) is user code between do_rnnt_pruning and
        # rnnt_loss_pruned; it IS inside the timed step (value/ms_per_step include it) and is reported here so that the
        # loss-only time can be read off: ms_per_step - joiner_standin_us_per_step / 1000.
        "joiner_standin_us_per_step": round(joiner_us, 1),
        "loss_only_us_per_step": round(1e3 * ms_per_step - joiner_us, 1),
        "native_aggregate": {"algorithmic_MB_per_step": round(tot_alg / 1e6, 1), "us_per_step": round(tot_us, 1),
                             "GBps": round(tot_alg / (tot_us * 1e-6) / 1e9, 1) if tot_us > 0 else None,
                             "frac_of_peak": round(tot_alg / (tot_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if tot_us > 0 else None},
        "dense_am_pruned_ms_per_step": round(dense_ms, 4) if dense_ms is not None else None,
        "kernels": kernels,
        "loss": float(last.item()),
        "gemm_tuning": gemm_choices if not args.no_gemm_tuning else False,
    }
    if world == 1 and not args.no_graph:
        gr = graph_replay((lambda: simple_step(inp)) if which == "simple" else (lambda: pruned_step(inp, r, first_pass=first_pass)), args.steps)
        if gr.get("ms_per_step"):
            gr["value"] = round(B / (gr["ms_per_step"] * 1e-3), 2)
        out["graph_replay"] = gr
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(B, T, S, C, r, simple_only=(which == "simple"))
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
