#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X Ed25519 / X25519 engine.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--op verify|x25519|sign] [--log2n 20]

Metric (BASELINE.json): ed25519 verifies/s at batch 2^20 per GPU (configs[1]; --op x25519 and
--op sign measure configs[2] and configs[4]).  One "step" = one pass of the hot path over the
whole synthetic batch, inputs already resident in HBM.  N > 1 is launched by torch.distributed.run
with one rank per GPU: every rank owns its own 2^20-item shard (weak scaling, no data-path
collective) and the step ends with the RCCL all-gather of the result bytes.

Prints ONE JSON line on rank 0 (see the driver contract), with two extra objects:
  roofline      the dominant kernel (k_verify_main) against the integer-VALU multiply-issue
                roofline SURVEY 8(d) prescribes for this path, plus the HBM view of the same launch
  cpu_baseline  the reference itself (oracle/_ref, compiled from its own sources) timed on this
                box's host cores on a bounded sample of the same workload
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd as ed  # noqa: E402
import workload  # noqa: E402

# canonical 32x32->64 multiply counts per item (BASELINE.md "Work per item")
MUL32_VERIFY = 312370
MUL32_VERIFY_MAIN = 312370 - (255 * 55 + 19 * 100) - (254 * 55 + 13 * 100)  # minus ed_import, ed_export
MUL32_X25519 = 202050
MUL32_SIGN = 64570
BYTES = {"verify": 129, "x25519": 96, "sign": 160}      # algorithmic HBM bytes per item
# v_mad_u64_u32 issue peak: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz (profiles/r01_valu_rates.txt
# measures 36-37 T lane-MAC/s at the clock the chip holds under this load)
PEAK_TMUL32 = 256 * 4 * 16 * 2.4e9 / 1e12
PEAK_HBM_GBS = 8000.0


def make_workload(op, n, rank, device):
    """SURVEY 8(d) seeded synthetic batch from tools/workload.py (SplitMix64 streams; rank r owns
    items r*n .. (r+1)*n-1 of the stream).  verify: 32-byte messages, random keys, items with
    i % 16 == 5 corrupted by one flipped bit in R, S, A or the message (round-robin).  Signatures
    and public keys come from the engine's own sign/genpub kernels (parity-tested separately; the
    2^20 batches are the ones whose digests tests/golden/batch_digests.json pins)."""
    first = rank * n
    up = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    if op == "x25519":
        sc, pt = workload.x25519_inputs(n, first=first)
        return {"scalars": up(sc), "points": up(pt)}
    if op == "sign":
        sk, msg = workload.sign_inputs(n, first=first)
        sk, msg = up(sk), up(msg)
        return {"secs": sk, "pubs": ed.ed25519_genpub_batch(sk), "msgs": msg}
    sk, msg = workload.sign_inputs(n, seed=1, config=2, first=first)
    d_sk = up(sk)
    pk = ed.ed25519_genpub_batch(d_sk)
    sig = ed.ed25519_sign_batch(d_sk, pk, up(msg)).cpu().numpy()
    pk = pk.cpu().numpy()
    expect = workload.corrupt_for_verify(sig, pk, msg, first=first)
    return {"sigs": up(sig), "pubs": up(pk), "msgs": up(msg), "expect": up(expect)}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes of tools/profile.sh (same
    command, separate passes), as committed in profiles/: WRITE_SIZE (KB, exact) + 2 x FETCH_SIZE
    (KB; gfx950 reports half the bytes of 16-byte-per-lane reads, MI355X_MICROARCH.md, HBM).
    None when no profile has been committed."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        prof = json.load(open(path))
        ks = [prof["ed::" + name.strip()] for name in kernel.split("+")]
        fetch, write = sum(k["FETCH_SIZE"] for k in ks), sum(k["WRITE_SIZE"] for k in ks)
        return {"bytes": (2.0 * fetch + write) * 1024.0, "fetch_size_kb_raw": fetch, "write_size_kb_raw": write,
                "source": "profiles/pmc_summary.json (rocprofv3 --pmc, separate passes)"}
    except (OSError, KeyError, ValueError):
        return None


def pmc_valu_busy(kernel):
    """VALU-busy fraction of `kernel` from the same committed PMC passes: SQ_INSTS_VALU wave-instructions
    x 4 clocks (one VALU instruction per SIMD per 4 clocks, profiles/r01_valu_rates.txt) / (1024 SIMDs x
    GRBM_GUI_ACTIVE / 8 XCDs).  None when the profile lacks the counters."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
        k = prof["ed::" + kernel.split("+")[0].strip()]
        return 4.0 * k["SQ_INSTS_VALU"] / (1024.0 * k["GRBM_GUI_ACTIVE"] / 8.0)
    except (OSError, KeyError, ValueError, TypeError, ZeroDivisionError):
        return None


def run_step(op, w):
    if op == "verify":
        return ed.ed25519_verify_batch(w["sigs"], w["pubs"], w["msgs"], msg_len=32)
    if op == "x25519":
        return ed.x25519_batch(w["scalars"], w["points"])
    return ed.ed25519_sign_batch(w["secs"], w["pubs"], w["msgs"], msg_len=32)


def usable_cores():
    """host cores this process may actually use: affinity mask, capped by the cgroup CPU quota"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(op, w, gpu_out, sample):
    """Time the reference (oracle/_ref, kind "reference"; else the C restatement, kind "port") on
    all host cores over the first `sample` items of the same workload; check the GPU against it."""
    cores = usable_cores()
    refdrv = os.path.join(ROOT, "oracle", "_ref", "libref_driver.so")
    if os.path.exists(refdrv):
        lib, kind = ctypes.CDLL(refdrv), "reference"
        fn = {"verify": lib.refdrv_verify_batch, "x25519": lib.refdrv_x25519_batch, "sign": lib.refdrv_sign_batch}[op]
    else:
        lib, kind = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "port"
        fn = {"verify": lib.orc_ed25519_verify_batch, "x25519": lib.orc_x25519_batch, "sign": lib.orc_ed25519_sign_batch}[op]
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    host = {k: v[:sample].cpu().numpy().copy() for k, v in w.items() if k != "expect"}
    if op == "verify":
        out = np.zeros(sample, np.uint8)
        args = (P(out), P(host["sigs"]), P(host["pubs"]), P(host["msgs"]), ctypes.c_size_t(32), ctypes.c_size_t(sample), cores)
    elif op == "x25519":
        out = np.zeros((sample, 32), np.uint8)
        args = (P(out), P(host["scalars"]), P(host["points"]), ctypes.c_size_t(sample), cores)
    else:
        out = np.zeros((sample, 64), np.uint8)
        args = (P(out), P(host["secs"]), P(host["pubs"]), P(host["msgs"]), ctypes.c_size_t(32), ctypes.c_size_t(sample), cores)
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        fn(*args)
        best = min(best, time.perf_counter() - t0)
    same = bool(np.array_equal(out.reshape(sample, -1), gpu_out[:sample].cpu().numpy().reshape(sample, -1)))
    unit = {"verify": "verifies/s", "x25519": "ops/s", "sign": "signs/s"}[op]
    return {"value": sample / best, "unit": unit, "cores": cores, "kind": kind,
            "sample": f"first {sample} items of the same batch, {cores} pthreads, best of 3",
            "per_core": sample / best / cores, "gpu_matches_cpu_on_sample": same}


def gather_results(out, world):
    """N > 1: all-gather the result bytes of every rank's shard (the only exchange of the path)."""
    if world == 1:
        return out
    if dist.get_backend() == "gloo":                    # test hook (see main): stage through the host
        out = out.cpu()
    full = torch.empty((world * out.shape[0],) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
    dist.all_gather_into_tensor(full, out)              # concatenated along dim 0: valid on nccl and gloo
    return full.view((world,) + tuple(out.shape))


def timed_region(step, steps, world, sync, device):
    """EXACTLY `steps` steps between two (barrier + device sync) brackets; returns the MAX over
    ranks of the elapsed seconds and the last step's output."""
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--op", choices=("verify", "x25519", "sign"), default="verify")
    ap.add_argument("--log2n", type=int, default=20, help="items per GPU = 2^log2n")
    ap.add_argument("--cpu-sample", type=int, default=1 << 18)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    # Test hooks, so that the N > 1 code path can be exercised on a one-GPU box: EDDSA_BENCH_SHARE_GPU=1
    # maps every rank onto the visible devices round-robin, EDDSA_BENCH_BACKEND=gloo replaces RCCL
    # (which refuses two ranks on one device).  Neither is set by the driver's runs.
    if os.environ.get("EDDSA_BENCH_SHARE_GPU") == "1":
        local %= torch.cuda.device_count()
    backend = os.environ.get("EDDSA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    ed.init(local)

    op, n = args.op, 1 << args.log2n
    w = make_workload(op, n, rank, device)
    torch.cuda.synchronize()

    marks = []                                          # (start, end) events around each kernel-only part

    def step():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = run_step(op, w)
        e1.record()
        marks.append((e0, e1))
        gather_results(out, world)                      # the final result gather (RCCL over xGMI)
        return out

    for _ in range(args.warmup):
        out = step()
    marks.clear()
    if op == "verify":
        ed.set_profiling(True)
    elapsed, out = timed_region(step, args.steps, world, torch.cuda.synchronize, device)
    phases = ed.verify_phase_ms() if op == "verify" else None
    ed.set_profiling(False)

    correct = True
    if op == "verify":
        correct = bool(torch.equal(out, w["expect"]))
    if world > 1:                                       # every rank's shard must be right
        flag = torch.tensor([int(correct)], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        correct = bool(flag.item())
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n * args.steps / elapsed
        mul32 = {"verify": MUL32_VERIFY, "x25519": MUL32_X25519, "sign": MUL32_SIGN}[op]
        if op == "verify":
            kernel, k_ms, k_mul32 = "k_verify_main", phases[1], MUL32_VERIFY_MAIN
        else:
            k_ms = sum(a.elapsed_time(b) for a, b in marks) / len(marks)      # HIP events, launch stream
            kernel, k_mul32 = {"x25519": "k_x25519_ladder + k_x25519_finish", "sign": "k_sign_point + k_sign_finish"}[op], mul32
        achieved = n * k_mul32 / (k_ms * 1e-3) / 1e12
        roofline = {
            "bound": "valu", "kernel": kernel, "achieved": achieved, "peak": PEAK_TMUL32,
            "unit": "Tmul32/s", "frac": achieved / PEAK_TMUL32, "traffic": pmc_traffic(kernel),
            "valu_busy": pmc_valu_busy(kernel),
            "kernel_ms": k_ms, "canonical_mul32_per_item": k_mul32,
            "note": "integer-VALU multiply-issue roofline (SURVEY 8d): canonical 32x32->64 products of the "
                    "reference's radix-2^25.5 schoolbook per item / v_mad_u64_u32 issue peak; the path is not "
                    "HBM- or MFMA-bound",
            "hbm": {"achieved": n * BYTES[op] / (ms_per_step * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": n * BYTES[op] / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS,
                    "algorithmic_bytes_per_item": BYTES[op]},
        }
        if phases:
            roofline["phase_ms"] = {"k_verify_prepare": phases[0], "k_verify_main": phases[1], "k_verify_finish": phases[2]}
        # the CPU baseline is timed at N = 1 only (at N > 1 the other ranks' host threads share the cores)
        base = cpu_baseline(op, w, out, min(args.cpu_sample, n)) if world == 1 else None
        correct = correct and (base is None or base["gpu_matches_cpu_on_sample"])
        line = {
            "metric": {"verify": "ed25519 verifies/sec", "x25519": "x25519 ops/sec", "sign": "ed25519 signs/sec"}[op],
            "value": value, "unit": {"verify": "verifies/s", "x25519": "ops/s", "sign": "signs/s"}[op],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (radix-2^25.5 limbs, u64 accumulators)", "data": "synthetic",
            "config": {"workload": f"batch=2^{args.log2n} per GPU, ed25519 {op}" if op != "x25519" else
                       f"batch=2^{args.log2n} per GPU, x25519 variable-base", "msg_len": 32 if op != "x25519" else None,
                       "items_per_gpu": n, "parallelism": f"shard{world}+allgather" if world > 1 else "single"},
            "outputs_correct": correct,
            "roofline": roofline, "cpu_baseline": base,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    if not correct:
        raise SystemExit("bench.py: GPU outputs differ from the expected verdicts / CPU reference")


if __name__ == "__main__":
    main()
