#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric: fp32 sgemm GFLOP/s and fraction of the MI355X fp32
MFMA roofline at M=N=K=8192 (configs[1]), on 1/2/4/8 GPUs of one node.

  python bench.py --gpus N --steps K --warmup W                      (any N: ONE process drives the N GPUs through the
                                                                      C-ABI's laser_hip_gemm_strided_f32_sharded_dev)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W       (one process per GPU, RCCL through torch.distributed)

A "step" is one pass of the hot path: C <- A.B through laser_hip's device-resident entry point
(operands already resident in HBM; the PCIe-inclusive host-pointer rate is reported in DESIGN.md,
never here).  N = 1: the 8192^3 problem.  N > 1: weak scaling -- every rank owns 8192 rows of an
(8192 N) x 8192 x 8192 problem (configs[4] at N = 8), row panels dealt block-cyclically, C
all-gathered over RCCL/xGMI inside the timed region (laser_amd/distributed.py).

Prints ONE JSON line on rank 0.  `roofline` prices the GEMM kernel alone against the dense fp32
MFMA peak (157.3 TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md); `cpu_baseline` is the CPU
oracle's OpenMP restatement of Laser's algorithm timed on this host's cores on a bounded sample of
the same workload -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

# Thread placement for the CPU baseline's OpenMP team: must be in the environment before the first OpenMP runtime of this
# process initialises (torch / numpy load one on import).  Without it Laser's nest -- ceil(M/192) ic tasks + jr tasks over
# two sockets -- migrates between cores and its timing swings by an order of magnitude (r02: 291 vs 3772 GFLOP/s).
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz
# laser_hip_last_f32_asm() -> the hand-scheduled assembly kernel the last launch ran (laser_amd/csrc/gemm_f32_asm.cpp)
# (the symbol table of gemm_f32_asm.cpp: kKernels, in order; tests/test_host_logic_cpu.py checks the two lists agree)
ASM_KERNEL_SYMBOLS = [
    "lh_f32_exact_256x128x32", "lh_f32_fast_256x256x16", "lh_f32_exact_128x128x16", "lh_f32_fast_128x128x16",
    "lh_f32_exact_256x128x32_nt", "lh_f32_fast_256x256x16_nt", "lh_f32_exact_128x128x16_nt", "lh_f32_fast_128x128x16_nt",
    "lh_f32_fast_256x128x32", "lh_f32_fast_256x128x32_nt", "lh_f32_conv_exact_256x128x32", "lh_f32_conv_fast_256x128x32",
    "lh_f32_exact_64x64x32", "lh_f32_fast_64x64x32", "lh_f32_exact_64x64x32_nt", "lh_f32_fast_64x64x32_nt",
    "lh_f64_exact_128x128x16", "lh_f64_fast_128x128x16", "lh_f64_exact_64x64x16", "lh_f64_fast_64x64x16", "lh_i32_128x128x32",
    "lh_f32_conv_exact_128x128x32", "lh_f32_conv_fast_128x128x32", "lh_f32_conv_exact_64x128x32", "lh_f32_conv_fast_64x128x32",
    "lh_f64_exact_128x128x16_nt", "lh_f64_fast_128x128x16_nt", "lh_f64_exact_64x64x16_nt", "lh_f64_fast_64x64x16_nt", "lh_i64_64x64x32",
    "lh_f32_exact_128x128x32", "lh_f32_fast_128x128x32", "lh_f32_exact_128x128x32_nt", "lh_f32_fast_128x128x32_nt",
    "lh_f32_exact_256x128x32_pre", "lh_f32_exact_256x128x32_pre_nt", "lh_f32_fast_256x256x16_pre", "lh_f32_fast_256x256x16_pre_nt",
    "lh_f32_exact_128x128x16_pre", "lh_f32_exact_128x128x16_pre_nt", "lh_f32_fast_128x128x16_pre", "lh_f32_fast_128x128x16_pre_nt",
    "lh_f32_exact_64x64x32_pre", "lh_f32_exact_64x64x32_pre_nt", "lh_f32_fast_64x64x32_pre", "lh_f32_fast_64x64x32_pre_nt",
    "lh_f32x16_exact_96x96x32", "lh_f32x16_fast_96x96x32", "lh_f32x16_exact_96x96x32_nt", "lh_f32x16_fast_96x96x32_nt",
    "lh_f32x16_exact_160x96x32", "lh_f32x16_fast_160x96x32", "lh_f32x16_exact_160x96x32_nt", "lh_f32x16_fast_160x96x32_nt",
    "lh_f32x16_exact_128x96x32", "lh_f32x16_fast_128x96x32", "lh_f32x16_exact_128x96x32_nt", "lh_f32x16_fast_128x96x32_nt",
    "lh_f32x16_exact_192x96x32", "lh_f32x16_fast_192x96x32", "lh_f32x16_exact_192x96x32_nt", "lh_f32x16_fast_192x96x32_nt",
    "lh_f32x16_exact_160x160x32", "lh_f32x16_fast_160x160x32", "lh_f32x16_exact_160x160x32_nt", "lh_f32x16_fast_160x160x32_nt",
    "lh_f32_conv_exact_256x128x32_p", "lh_f32_conv_fast_256x128x32_p", "lh_f32_conv_exact_128x128x32_p", "lh_f32_conv_fast_128x128x32_p",
    "lh_f32_conv_exact_64x128x32_p", "lh_f32_conv_fast_64x128x32_p"]
ASM_KERNEL_NAMES = {1 + i: n + " (hand-scheduled assembly)" for i, n in enumerate(ASM_KERNEL_SYMBOLS)}
COMPILER_KERNEL_NAME = "gemm_mfma_kernel<float,...> (compiler-scheduled)"


def last_kernel_name(laser_amd):
    """Name of the kernel the last f32 GEMM launch of this process ran (laser_hip_get_option "last_f32_asm")."""
    return ASM_KERNEL_NAMES.get(laser_amd.last_f32_asm(), COMPILER_KERNEL_NAME)


SIZE = 8192


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_info():
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return os.cpu_count() or 1, model


def physical_cores():
    """Physical cores of this host (unique (package, core) pairs of /proc/cpuinfo); logical CPUs if that fails."""
    try:
        cores, pkg = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pkg = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cores.add((pkg, line.split(":", 1)[1].strip()))
        if cores:
            return len(cores)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline(samples=3):
    """Laser's OpenMP CPU path (the oracle's C restatement, kind "port") on the FULL 8192^3 job, the reference's own
    protocol (benchmarks/gemm/gemm_bench_float32.nim:8-40,57: warm-up, then timed samples of the whole product, mean /
    min / max / stddev like printStats), threads pinned (OMP_PROC_BIND=close, OMP_PLACES=cores).  Two team sizes with equal
    prominence: every physical core (`all_physical_cores`) and the best team of a short calibration (`best_team`) --
    Laser's nest exposes only ceil(M/192) = 43 ic tasks (+ jr tasks) behind a serial pc loop (gemm.nim:150-176), so on a
    128-core two-socket host "all cores" is far from its best team.  `value` / `cores` quote the best team."""
    import numpy as np
    from oracle import oracle
    oracle.build()
    rng = np.random.default_rng(42)
    n = SIZE
    B = rng.uniform(-0.1, 0.1, (n, n)).astype(np.float32)
    A = rng.uniform(-0.1, 0.1, (n, n)).astype(np.float32)
    isa = oracle.detect_isa(np.float32)
    flop = 2.0 * n * n * n

    def run(rows=n):
        C = np.zeros((rows, n), dtype=np.float32)
        t0 = time.perf_counter()
        oracle.gemm_strided(rows, n, n, 1.0, A, n, 1, B, n, 1, 0.0, C, n, 1, isa=isa)
        return time.perf_counter() - t0

    def stats(ts, threads):
        mean = sum(ts) / len(ts)
        sd = (sum((t - mean) ** 2 for t in ts) / max(1, len(ts) - 1)) ** 0.5
        return {"threads": threads, "gflops": round(flop / mean / 1e9, 1), "mean_s": round(mean, 4), "min_s": round(min(ts), 4),
                "max_s": round(max(ts), 4), "stddev_s": round(sd, 4), "samples": len(ts)}

    ncpu, model = host_info()
    cores = min(physical_cores(), oracle.num_threads())
    oracle.set_num_threads(cores)
    run(192)                      # page faults, thread pool
    run()                         # the warm-up sample
    all_cores = stats([run() for _ in range(samples)], cores)
    # team-size calibration on 1920 rows (10 ic tasks), then the same protocol at the best team size
    best = None
    for th in sorted({cores, max(1, cores // 2), max(1, cores // 4), min(cores, 32), min(cores, 16), min(cores, 8)}):
        oracle.set_num_threads(th)
        run(1920)
        t_cal = min(run(1920) for _ in range(2))
        if best is None or t_cal < best[1]:
            best = (th, t_cal)
    oracle.set_num_threads(best[0])
    run()
    best_team = stats([run() for _ in range(samples)], best[0]) if best[0] != cores else dict(all_cores)
    names = {0: "generic", 1: "sse", 2: "sse2", 3: "sse4.1", 4: "avx", 5: "avx+fma", 6: "avx2", 7: "avx512"}
    # the reference's own comparator ("vendor BLAS", gemm_bench_float32.nim:191-197): numpy == OpenBLAS
    _ = A[:512] @ B
    tb = []
    for _ in range(2):
        t0 = time.perf_counter()
        _ = A @ B
        tb.append(time.perf_counter() - t0)
    blas_gflops = flop / (sum(tb) / len(tb)) / 1e9
    log(f"[cpu_baseline] host: {ncpu} logical / {physical_cores()} physical CPUs, {model}; isa {names.get(isa)}; "
        f"{cores} threads: {all_cores['gflops']:.1f} GFLOP/s; best team {best[0]}: {best_team['gflops']:.1f} GFLOP/s; "
        f"numpy/OpenBLAS {blas_gflops:.1f} GFLOP/s")
    return {"value": best_team["gflops"], "unit": "GFLOP/s", "cores": best_team["threads"], "kind": "port",
            "best_team": best_team, "all_physical_cores": all_cores, "openblas_same_job_gflops": round(blas_gflops, 1),
            "omp": {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES")},
            "sample": f"the whole {n}^3 sgemm (M=N=K={n}), 1 warm-up + {samples} timed samples per team, Laser's algorithm restated in "
                      f"C (oracle/), ukernel {names.get(isa)}, threads pinned to cores; `value` = the best team of a calibration over "
                      f"{{all, 1/2, 1/4, 32, 16, 8}} threads ({best_team['threads']} threads, {best_team['mean_s']:.2f} s per sample), "
                      f"`all_physical_cores` = {cores} threads ({all_cores['mean_s']:.2f} s): Laser's loop nest offers only "
                      f"ceil(M/192) = {(n + 191) // 192} row-block tasks per kc slice, so large teams mostly wait; host {model}"}


def side_configs(budget_s=10.0):
    """BASELINE.json's other single-GPU configs and the reference's own published shapes, one compact line each (ms, TFLOP/s,
    fraction of the fp32 MFMA peak, and the oracle's CPU time for the same call beside it): C1 128^3, the reference's
    headline 1920^3 (benchmarks/gemm/gemm_bench_float32.nim:383-410), C3 4096^3 with B transposed, C4 conv pad 1, and the
    reference's conv bench geometry (16,3,224,224) * (20,3,3,3) pad 0 (benchmarks/convolution/conv2d_bench.nim:130-170)."""
    import numpy as np
    import torch
    import laser_amd
    from oracle import oracle
    oracle.build()
    out = []
    t_start = time.perf_counter()

    def gpu_ms_stats(fn, total_ms=120.0):
        """>= total_ms of back-to-back launches after the warm-up, every launch timed on its own (event pairs): min / median / max.
        For lines whose round-3 figure did not reproduce between boxes (C3: 0.92 ms in one harness, 1.09 - 1.15 ms in another --
        12 launches after a 30 ms warm-up caught the clocks mid-ramp)."""
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
            for _ in range(8):
                fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        n = max(50, int(total_ms / max(1e-3, e0.elapsed_time(e1))))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for i in range(n):
            fn()
            evs[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
        return {"launches": n, "min_ms": round(ts[0], 4), "median_ms": round(ts[n // 2], 4), "max_ms": round(ts[-1], 4),
                "mean_ms": round(sum(ts) / n, 4)}

    def side_roofline(flops, st_):
        """`roofline` of a side line: algorithmic flops / the AVERAGE duration of one call (every launch timed on its own with HIP
        events on the launch stream) against the fp32 MFMA peak; `kernel` = what the last launch ran."""
        tf_ = flops / (st_["mean_ms"] * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(tf_, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf_ / FP32_MFMA_PEAK_TFLOPS, 4), "kernel": last_kernel_name(laser_amd), "kernel_ms": st_["mean_ms"],
                "launches_timed": st_["launches"], "algorithmic_flops_per_launch": flops}

    def gpu_ms(fn, inner=4, reps=3):
        # steady state: the clocks ramp up over the first ~20 ms of work after an idle gap (the CPU oracle runs between the
        # lines); a C4 conv launch measured 470 us right after idle and 405 us fifty launches later
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.03:
            for _ in range(8):
                fn()
            torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / inner)
        return sorted(ts)[len(ts) // 2]

    def rnd(shape, seed, lo=-0.1, hi=0.1):
        g = torch.Generator(device="cuda").manual_seed(seed)
        return torch.rand(shape, generator=g, device="cuda") * (hi - lo) + lo

    def cpu_s(fn):
        if time.perf_counter() - t_start > budget_s:
            return None
        fn()
        t0 = time.perf_counter()
        fn()
        return round(time.perf_counter() - t0, 4)

    # Launch-bound lines (tens of microseconds of GPU work): the Python mirror's per-call argument checks cost as much as the
    # kernel, so the timed loop calls the C-ABI entry point itself, bound once through ctypes with its argument tuple prebuilt
    # (what a compiled caller does; the mirror has validated the same call once before).  `python_mirror_ms` is the mirror's rate.
    import ctypes
    from laser_amd import _lib as _lh
    L = _lh.lib()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def vp(t):
        return ctypes.c_void_p(t.data_ptr())

    def gemm_line(name, M, N, K, A, B, C, prebound=False, alpha=1, beta=0, stats=False):
        mirror = lambda: laser_amd.matmul(A, B, alpha, beta, C)
        extra = {}
        if prebound:
            mirror()
            cargs = (M, N, K, ctypes.c_float(1.0), vp(A), A.stride(0), A.stride(1), vp(B), B.stride(0), B.stride(1), ctypes.c_float(0.0),
                     vp(C), C.stride(0), C.stride(1), stream)
            fn = L.laser_hip_gemm_strided_f32_dev
            ms = gpu_ms(lambda: fn(*cargs), inner=16)
            extra = {"python_mirror_ms": round(gpu_ms(mirror, inner=16), 4), "timed": "C-ABI entry bound once via ctypes", "kernel": laser_amd.last_f32_asm()}
        elif stats:
            st_ = gpu_ms_stats(mirror)
            ms = st_["median_ms"]
            extra = {"timing": st_, "kernel": laser_amd.last_f32_asm(), "roofline": side_roofline(2.0 * M * N * K, st_)}
        else:
            ms = gpu_ms(mirror)
            extra = {"kernel": laser_amd.last_f32_asm()}
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        Ah, Bh = A.cpu().numpy(), B.cpu().numpy()
        cs = cpu_s(lambda: oracle.matmul(Ah, Bh)) if (alpha == 1 and beta == 0) else None
        out.append({"config": name, "ms": round(ms, 4), "tflops": round(tf, 2), "frac_mfma_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4),
                    "cpu_oracle_s": cs, **extra})

    try:
        gemm_line("C1 fp32 128^3 (device-resident; launch-bound)", 128, 128, 128, rnd((128, 128), 1), rnd((128, 128), 2),
                  torch.zeros((128, 128), device="cuda"), prebound=True)
        n = 1920
        # (a 0.11-ms launch: on a busy host the Python mirror's per-call work is the rate -- bench_v5.json read 0.1236 ms through it where
        # the C-ABI symbol bound once reads 0.1097 in the same call; timed like C1, the mirror's rate beside it)
        gemm_line("fp32 1920^3 (the reference's published shape)", n, n, n, rnd((n, n), 3), rnd((n, n), 4), torch.zeros((n, n), device="cuda"), prebound=True)
        n = 4096
        gemm_line("C3 fp32 4096^3, B transposed (rowStrideB=1, colStrideB=K)", n, n, n, rnd((n, n), 5), rnd((n, n), 6).t(),
                  torch.zeros((n, n), device="cuda"), stats=True)
        n = 8192
        gemm_line("C2 with alpha=0.5, beta=0.25 (fp32 8192^3; the running sum starts as beta*C0, every slice is scaled)", n, n, n,
                  rnd((n, n), 9), rnd((n, n), 10), rnd((n, n), 11), alpha=0.5, beta=0.25)
        for name, ishape, kshape, pad in (("C4 conv (32,128,56,56)*(256,128,3,3) pad 1", (32, 128, 56, 56), (256, 128, 3, 3), (1, 1)),
                                         ("reference conv bench (16,3,224,224)*(20,3,3,3) pad 0", (16, 3, 224, 224), (20, 3, 3, 3), (0, 0))):
            st = (1, 1)
            x, w = rnd(ishape, 7, 0, 1), rnd(kshape, 8, 0, 1)
            oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, st)
            o = torch.zeros(oshape, device="cuda")
            mirror = lambda: laser_amd.conv2d_im2col(o, oshape, x, ishape, w, kshape, pad, st, None)
            extra = {}
            if kshape[0] <= 32:          # the reference's own conv bench: ~27 us of GPU work
                mirror()
                cargs = (vp(o), vp(x), *ishape, vp(w), *kshape, *pad, *st, None, stream)
                fn = L.laser_hip_conv2d_im2col_f32_dev
                ms = gpu_ms(lambda: fn(*cargs), inner=16)
                extra = {"python_mirror_ms": round(gpu_ms(mirror, inner=16), 4), "timed": "C-ABI entry bound once via ctypes",
                         "hbm_frac": round(4.0 * (x.numel() + o.numel()) / (ms * 1e-3) / 8e12, 3)}
            fl = 2.0 * oshape[0] * oshape[1] * oshape[2] * oshape[3] * kshape[1] * kshape[2] * kshape[3]
            if kshape[0] > 32:           # C4: every call timed on its own, like C3 (one call = the main launch + any tail launches)
                st_ = gpu_ms_stats(mirror)
                ms = st_["median_ms"]
                extra = {"timing": st_, "roofline": side_roofline(fl, st_), "tail_cut_at_pixel": laser_amd.last_split()}
            tf = fl / (ms * 1e-3) / 1e12
            xh, wh = x.cpu().numpy(), w.cpu().numpy()
            cs = cpu_s(lambda: oracle.conv2d_im2col(xh, wh, pad, st))
            out.append({"config": name, "ms": round(ms, 4), "tflops": round(tf, 2), "frac_mfma_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4),
                        "cpu_oracle_s": cs, **extra})
    except Exception as e:      # a side line must never cost the headline
        out.append({"error": f"{type(e).__name__}: {e}"[:300]})
    return out


def hashed_rows(dev, rows, cols, salt):
    """Operand rows as a pure function of their GLOBAL (row, column) index (a 32-bit integer hash), uniform in [-0.1, 0.1) like the
    reference's bench inputs (gemm_bench_float32.nim:343-344): any rank / device can regenerate any row of the global A to verify
    rows it received through the gather.  (Random data is mandatory: zero-filled operands run at a higher clock.)"""
    import torch
    r = torch.as_tensor(rows, dtype=torch.int64, device=dev).view(-1, 1)
    c = torch.arange(cols, dtype=torch.int64, device=dev).view(1, -1)
    h = (r * 2654435761 + c * 40503 + salt) & 0xFFFFFFFF
    h = (h ^ (h >> 15)) * 2246822519 & 0xFFFFFFFF
    h = (h ^ (h >> 13)) * 3266489917 & 0xFFFFFFFF
    h = h ^ (h >> 16)
    return ((h & 0xFFFFFF).to(torch.float32) / 16777216.0 - 0.5) * 0.2


def single_process_primary(args):
    """`python bench.py --gpus N` without torchrun: ONE process drives the N GPUs through the product boundary --
    laser_hip_gemm_strided_f32_sharded_dev (block-cyclic row panels of A, B replicated, C gathered on every GPU inside
    the timed region; the parallelism lives inside one call, like gemm.nim:160-176).  Weak scaling: M = size * N."""
    import torch
    import laser_amd
    ndev, n = args.gpus, args.size
    one_gpu = os.environ.get("LASER_BENCH_ONE_GPU") == "1"   # test hook: every device slot on GPU 0 (timings meaningless)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; laser_amd has no CPU fallback")
    if not one_gpu and torch.cuda.device_count() < ndev:
        raise SystemExit(f"--gpus {ndev} but only {torch.cuda.device_count()} GPUs are visible")
    devices = [0] * ndev if one_gpu else list(range(ndev))
    laser_amd._lib.check(laser_amd.lib().laser_hip_init(0))
    if args.mode is not None:
        laser_amd.set_float_mode(0 if args.mode == "laser_order" else 1)
    laser_amd.set_f32_config(args.cfg)
    mode = "laser_order" if laser_amd.get_float_mode() == 0 else "fast"
    M, N, K = n * ndev, n, n
    tdev = [torch.device("cuda", d) for d in devices]

    hashed = hashed_rows

    Bs = [hashed(tdev[g], range(K), N, 7) for g in range(ndev)]
    gm = {"none": laser_amd.GATHER_NONE, "peer": laser_amd.GATHER_PEER, "rccl": laser_amd.GATHER_RCCL}

    def build(ppd):
        rows, ppd_used, padded = laser_amd.shard_plan(M, ndev, ppd)
        Ap, Cs = [], []
        for g in range(ndev):
            a = torch.zeros((ppd_used * rows, K), dtype=torch.float32, device=tdev[g])
            for s_ in range(ppd_used):
                start = (s_ * ndev + g) * rows
                valid = max(0, min(rows, M - start))
                if valid > 0:
                    a[s_ * rows: s_ * rows + valid] = hashed(tdev[g], range(start, start + valid), K, 1)
            Ap.append(a)
            Cs.append(torch.zeros((padded, N), dtype=torch.float32, device=tdev[g]))
        return rows, ppd_used, Ap, Cs

    def call(Ap, Cs, ppd, gather):
        laser_amd.gemm_strided_sharded_dev(devices, M, N, K, 1.0, Ap, K, 1, Bs, N, 1, 0.0, Cs, N, ppd, gm[gather], 0)

    # untimed calibration of the two box questions (panels per GPU; transport of the gather), 1 + 2 steps per candidate
    cands = [(args.panels_per_rank, "peer")] if args.panels_per_rank > 0 else [(4, "peer"), (8, "peer")]
    if ndev > 1 and not one_gpu and args.panels_per_rank <= 0:
        cands += [(4, "rccl"), (8, "rccl")]
    if ndev == 1:
        cands = [(1, "peer")]
    calibration = []
    for ppd, gather in cands:
        rec = {"panels_per_dev": ppd, "gather": gather}
        try:
            _, _, Ap, Cs = build(ppd)
            call(Ap, Cs, ppd, gather)
            t0 = time.perf_counter()
            for _ in range(2):
                call(Ap, Cs, ppd, gather)
            rec["ms_per_step"] = round((time.perf_counter() - t0) / 2 * 1e3, 4)
            del Ap, Cs
        except Exception as e:
            rec["error"] = f"{type(e).__name__}: {e}"[:200]
        calibration.append(rec)
    good = [r for r in calibration if "ms_per_step" in r]
    if not good:
        raise SystemExit(f"no sharded configuration ran: {calibration}")
    pick = min(good, key=lambda r: r["ms_per_step"])
    ppd, gather = pick["panels_per_dev"], pick["gather"]
    rows, ppd_used, Ap, Cs = build(ppd)
    for _ in range(args.warmup):
        call(Ap, Cs, ppd, gather)
    for d in set(devices):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        call(Ap, Cs, ppd, gather)          # synchronous: returns when every GPU holds all of C
    for d in set(devices):
        torch.cuda.synchronize(d)
    wall = time.perf_counter() - t0
    # self-check on every device: a few rows owned by EVERY slot (only right if the gather delivered them), vs fp64
    err = 0.0
    for g in range(ndev):
        chk = []
        for r_ in range(ndev):
            start = ((ppd_used - 1) * ndev + r_) * rows
            chk += list(range(start, min(M, start + 2))) + list(range(r_ * rows, min(M, r_ * rows + 2)))
        ref = hashed(tdev[g], chk, K, 1).double() @ Bs[g].double()
        err = max(err, (Cs[g][chk].double() - ref).abs().max().item())
    assert err < 1e-4, f"bench self-check failed: max abs err {err}"
    # The step, taken apart: the SAME panels with every transport of the gather, each a labelled, timed figure --
    # "none" (the local products only: what the step would cost if the gather were free), "peer" (each finished panel pushed to every
    # other GPU over xGMI) and "rccl" (ncclAllGather per slab on a communicator over the N devices: the collective north_star names).
    # exposed_gather_ms = that leg - the "none" leg.  Whatever the calibration picked as the headline transport, all three are in the line.
    leg_steps = max(3, min(args.steps, 10))
    transports = {}
    for tname in ("none", "peer", "rccl"):
        rec = {}
        if tname == "rccl" and one_gpu and ndev > 1:
            rec["skipped"] = "RCCL refuses two ranks on one device (LASER_BENCH_ONE_GPU test hook)"
            transports[tname] = rec
            continue
        try:
            call(Ap, Cs, ppd, tname)
            for d in set(devices):
                torch.cuda.synchronize(d)
            t0_ = time.perf_counter()
            for _ in range(leg_steps):
                call(Ap, Cs, ppd, tname)
            for d in set(devices):
                torch.cuda.synchronize(d)
            rec["ms_per_step"] = round((time.perf_counter() - t0_) / leg_steps * 1e3, 4)
            rec["steps"] = leg_steps
            if tname == "rccl":
                rec["backend"] = {"name": "rccl (librccl.so loaded by liblaser_hip.so; ncclCommInitAll + ncclAllGather per slab)",
                                  "communicator_ranks": laser_amd.get_option("shard_rccl_ranks"), "devices": devices}
        except Exception as e:
            rec["error"] = f"{type(e).__name__}: {e}"[:200]
        transports[tname] = rec
    none_ms = transports["none"].get("ms_per_step")
    for tname in ("peer", "rccl"):
        if none_ms is not None and "ms_per_step" in transports[tname]:
            transports[tname]["exposed_gather_ms"] = round(transports[tname]["ms_per_step"] - none_ms, 4)
    call(Ap, Cs, ppd, gather)       # (leave every GPU holding all of C again, as after the timed region)
    for d in set(devices):
        torch.cuda.synchronize(d)
    slot_kernel = last_kernel_name(laser_amd)      # every slot multiplies the same panel shape under the same pinned tile class
    # roofline: the GEMM kernel alone on device slot 0 (its n-row share as one launch), HIP events on the launch stream
    torch.cuda.set_device(devices[0])
    Cl = torch.zeros((n, N), dtype=torch.float32, device=tdev[0])
    Al = Ap[0][:n]
    for _ in range(2):
        laser_amd.matmul(Al, Bs[0], 1, 0, Cl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        laser_amd.matmul(Al, Bs[0], 1, 0, Cl)
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / args.steps
    fl = 2.0 * n * N * K
    ach = fl / (k_ms * 1e-3) / 1e12
    value = 2.0 * M * N * K / (wall / args.steps) / 1e9
    out = {
        "metric": "sgemm GFLOP/s and %MFMA-peak, M=N=K=8192, 1/2/4/8 MI355X",
        "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": ndev, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"fp32 sgemm M={M} N={N} K={K} row-panel sharded over {ndev}xMI355X in ONE process, C gathered on every GPU "
                        f"inside the timed region (BASELINE configs[4] shape at 8 GPUs)",
            "entry_point": "laser_hip_gemm_strided_f32_sharded_dev (C-ABI; one host thread per GPU inside the call)",
            "M": M, "N": N, "K": K, "accumulation": mode, "panels_per_dev": ppd_used, "rows_per_panel": rows, "gather": gather,
            "parallelism": f"row-panels x{ndev}, {ppd_used} block-cyclic panels/GPU",
            "pct_of_fp32_mfma_peak": round(100.0 * value / 1e3 / (FP32_MFMA_PEAK_TFLOPS * ndev), 2),
            "calibration_untimed": calibration, "max_abs_err_vs_fp64": err,
            "device_slots": "all on GPU 0 (LASER_BENCH_ONE_GPU test hook: timings meaningless)" if one_gpu else devices,
            "slots": [{"slot": g, "device": devices[g], "device_name": torch.cuda.get_device_name(devices[g]), "kernel": slot_kernel,
                       "rows": ppd_used * rows, "panels": ppd_used} for g in range(ndev)],
            "transports": transports,
            "compute_only_ms": none_ms,
            "exposed_gather_ms": None if none_ms is None else round(wall / args.steps * 1e3 - none_ms, 4),
            "compute_only_gflops": None if none_ms is None else round(2.0 * M * N * K / (none_ms * 1e-3) / 1e9, 1),
        },
        "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                     "kernel": last_kernel_name(laser_amd),
                     "kernel_ms": round(k_ms, 4), "algorithmic_flops_per_launch": fl,
                     "note": "per-GPU kernel alone (device slot 0's row share as one launch), timed after the sharded run"},
    }
    if not args.no_cpu_baseline:      # the reference's OpenMP path restated, on this box's host cores: stated beside every line, N > 1 too
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)


def headline_plan(laser_amd):
    """The launch plan the LAST assembly GEMM launch of this process ran under (diagnostics of laser_hip_get_option): workgroups, K slices
    per tile, raster group height / XCD chunking -- what decides which tiles share an XCD's L2, i.e. the HBM-side traffic."""
    return {"wgs": laser_amd.get_option("last_asm_wgs"), "slices": laser_amd.get_option("last_asm_slices"),
            "group_m": laser_amd.get_option("last_asm_group_m")}


def kernel_source_sha16(plans):
    """Identity of what the committed PMC traffic figure was measured on: the sha-256 of the two headline kernels' generated gfx950
    assembly text (what the assembler turns into the code object's .text; produced here by the same generator the build runs) AND the
    launch plan each ran under at the headline shape, as the launcher reports it after the launch (`plans` = {mode: headline_plan}).
    A change to these kernels or to the launcher's plan for this shape (persistent vs plain, workgroup count, raster group, XCD
    chunking -- ADVICE r4) invalidates the figure; a change to any other kernel or launcher path does not."""
    import hashlib
    from laser_amd.asmgen import f32_kernel as K
    h = hashlib.sha256()
    for name in ("exact_256x128x32", "fast_256x256x16"):
        g = K.make(name)
        g.build()
        h.update(K.kernel_text(g, "lh_f32_" + name).encode())
    h.update(json.dumps(plans, sort_keys=True).encode())
    return h.hexdigest()[:16]


def pmc_traffic(mode, plan):
    """HBM-side bytes per launch of the headline kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json; FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, in bytes).
    Counters cannot be collected inside this process, so this is the latest COMMITTED MEASUREMENT
    of the same kernel on the same shape under the same launch plan -- quoted ONLY while the kernel text and the plan this run
    observed (`plan` = headline_plan after the timed launches) match what was recorded with the measurement
    (scripts/update_pmc_traffic.py) -- or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        plans = dict(d.get("plans") or {})
        if plans.get(mode) != plan:
            return None      # another launch plan than the one the counters were collected under
        if d.get("kernel_source_sha16") != kernel_source_sha16(plans):
            return None      # measured on other kernel sources: stale, do not quote it
        return d.get(mode)
    except (OSError, ValueError):
        return None


def single_process_sharded(ndev, n, steps, warmup, mode, one_gpu=False):
    """The same weak-scaling workload through the C-ABI's single-process entry point
    (laser_hip_gemm_strided_f32_sharded_dev: one process, one host thread per GPU, block-cyclic row panels, C gathered on
    every GPU), swept over its knobs so that ONE multi-GPU lease answers the open questions: panels per GPU, transport of
    the gather (peer copies vs RCCL vs none = compute-only scaling), 128x128 tile pin beside RCCL.  Wall time per step is
    host-timed around the synchronous call (it returns when every GPU holds all of C)."""
    import torch
    import laser_amd
    devices = [0] * ndev if one_gpu else list(range(ndev))
    M, N, K = n * ndev, n, n
    if mode is not None:
        laser_amd.set_float_mode(0 if mode == "laser_order" else 1)

    hashed = hashed_rows

    tdev = [torch.device("cuda", d) for d in devices]
    Bs = [hashed(tdev[g], range(K), N, 7) for g in range(ndev)]
    runs = []
    sweep = [(4, "peer", 0)]
    if ndev > 1:
        sweep += [(p, "peer", 0) for p in (1, 2, 8, 16)] + [(4, "none", 0), (4, "rccl", 0), (4, "rccl", 1), (8, "rccl", 1)]
    gm = {"none": laser_amd.GATHER_NONE, "peer": laser_amd.GATHER_PEER, "rccl": laser_amd.GATHER_RCCL}
    for ppd, gather, pin in sweep:
        if one_gpu and gather == "rccl":
            continue                     # RCCL refuses two ranks on one device
        rows, ppd_used, padded = laser_amd.shard_plan(M, ndev, ppd)
        Ap, Cs = [], []
        for g in range(ndev):
            a = torch.zeros((ppd_used * rows, K), dtype=torch.float32, device=tdev[g])
            for s_ in range(ppd_used):
                start = (s_ * ndev + g) * rows
                valid = max(0, min(rows, M - start))
                if valid > 0:
                    a[s_ * rows: s_ * rows + valid] = hashed(tdev[g], range(start, start + valid), K, 1)
            Ap.append(a)
            Cs.append(torch.zeros((padded, N), dtype=torch.float32, device=tdev[g]))
        rec = {"panels_per_dev": ppd_used, "rows_per_panel": rows, "gather": gather, "pin_128x128": bool(pin)}
        try:
            call = lambda: laser_amd.gemm_strided_sharded_dev(devices, M, N, K, 1.0, Ap, K, 1, Bs, N, 1, 0.0, Cs, N, ppd,
                                                              gm[gather], laser_amd.SHARD_PIN_TILE if pin else 0)
            for _ in range(warmup):
                call()
            ts = []
            for _ in range(steps):
                t0 = time.perf_counter()
                call()
                ts.append(time.perf_counter() - t0)
            ms = sum(ts) / len(ts) * 1e3
            rec.update(ms_per_step=round(ms, 4), min_ms=round(min(ts) * 1e3, 4), gflops=round(2.0 * M * N * K / (ms * 1e-3) / 1e9, 1))
            # self-check on every device: its own first rows and (gathered) a few rows of every other slot, vs fp64
            err = 0.0
            for g in range(ndev):
                owners = range(ndev) if gather != "none" else [g]
                chk = []
                for r_ in owners:
                    start = ((ppd_used - 1) * ndev + r_) * rows
                    chk += list(range(start, min(M, start + 2))) + list(range(r_ * rows, min(M, r_ * rows + 2)))
                ref = hashed(tdev[g], chk, K, 1).double() @ Bs[g].double()
                err = max(err, (Cs[g][chk].double() - ref).abs().max().item())
            rec["max_abs_err_vs_fp64"] = err
            assert err < 1e-4, f"self-check failed: {rec}"
        except Exception as e:          # a transport that is unavailable must not hide the others
            rec["error"] = f"{type(e).__name__}: {e}"[:300]
        runs.append(rec)
        del Ap, Cs
    good = [r for r in runs if "gflops" in r and r["gather"] != "none"] or [r for r in runs if "gflops" in r]
    best = max(good, key=lambda r: r["gflops"]) if good else None
    return {"entry_point": "laser_hip_gemm_strided_f32_sharded_dev (one process, one host thread per GPU)", "n_gpus": ndev,
            "M": M, "N": N, "K": K, "steps": steps, "warmup": warmup, "runs": runs, "best": best,
            "pct_of_fp32_mfma_peak": round(100.0 * best["gflops"] / 1e3 / (FP32_MFMA_PEAK_TFLOPS * ndev), 2) if best else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["laser_order", "fast"], default=None, help="fp32 accumulation mode (default: library default)")
    ap.add_argument("--cfg", type=int, default=-1, help="force an f32 tile configuration (-1 = heuristic)")
    ap.add_argument("--size", type=int, default=SIZE)
    ap.add_argument("--panels-per-rank", type=int, default=0,
                    help="block-cyclic row panels per rank for N > 1 (0 = pick among {4, 8} x {128x128 tile pinned, heuristic tile} "
                         "from a short untimed calibration before the warm-up)")
    ap.add_argument("--gather", choices=["auto", "collective", "p2p"], default="auto",
                    help="N > 1: all-gather of C as one collective per slab, as grouped point-to-point sends, or (auto) whichever the "
                         "untimed calibration measures faster")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true", help="skip the compact per-config side field (C1, 1920^3, C3, C4, the reference's conv shape)")
    ap.add_argument("--single-process", type=int, default=0, metavar="NDEV",
                    help="run ONLY the single-process sharded entry point of the C-ABI over NDEV GPUs and print its JSON")
    ap.add_argument("--no-single-process", action="store_true", help="skip the single-process sharded side measurement")
    args = ap.parse_args()

    if args.single_process > 0:
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X; laser_amd has no CPU fallback")
        res = single_process_sharded(args.single_process, args.size, args.steps, args.warmup, args.mode,
                                     one_gpu=os.environ.get("LASER_BENCH_ONE_GPU") == "1")
        print(json.dumps(res), flush=True)
        return

    import torch
    import torch.distributed as dist
    import laser_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        return single_process_primary(args)     # plain `bench.py --gpus N`: one process, the C-ABI's sharded entry point
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; laser_amd has no CPU fallback")
    # LASER_BENCH_ONE_GPU=1 (test hook only): every rank on GPU 0 over gloo -- exercises the N > 1 code path of this
    # file on a single-GPU box (timings meaningless; RCCL refuses two ranks on one device)
    one_gpu_test = os.environ.get("LASER_BENCH_ONE_GPU") == "1"
    if one_gpu_test:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_test:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    L = laser_amd.lib()
    laser_amd._lib.check(L.laser_hip_init(local_rank))
    if args.mode is not None:
        laser_amd.set_float_mode(0 if args.mode == "laser_order" else 1)
    laser_amd.set_f32_config(args.cfg)
    mode = "laser_order" if laser_amd.get_float_mode() == 0 else "fast"

    n = args.size
    M_total, N, K = n * world, n, n
    from laser_amd.distributed import ShardedGemm, SHARDED_ASM_TILE, SHARDED_TILE_NAME

    # Operands: hashed_rows (uniform [-0.1, 0.1), a pure function of the global index)
    def hashed(rows, cols, salt):
        return hashed_rows(dev, rows, cols, salt)

    B = hashed(range(K), N, 7)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def build(ppr, tile, gather="collective"):
        """The sharded problem for `ppr` block-cyclic panels per rank; tile: None = the 128x128x16 ASSEMBLY tile pinned for sharded
        runs (option asm_tile), -1 = the library heuristic (the 256x128 assembly tile at this size), >= 0 = that compiler configuration; gather: one all-gather collective per slab, or the same
        exchange as grouped point-to-point sends.  Returns (ShardedGemm, this rank's A panels, full C)."""
        g = ShardedGemm(M_total, N, K, torch.float32, dev, None, ppr if world > 1 else 1, tile_config=tile, gather=gather)
        pl = g.plan
        a = torch.zeros((pl.panels_per_rank * pl.rows, K), dtype=torch.float32, device=dev)
        for s_ in range(pl.panels_per_rank):
            start, valid = pl.panel(s_, rank)
            if valid > 0:
                a[s_ * pl.rows: s_ * pl.rows + valid] = hashed(range(start, start + valid), K, 1)
        return g, a, g.alloc_C()

    # N > 1: how many panels per rank (a shorter exposed last all-gather vs larger local products) and whether the local
    # products share the CUs with RCCL better on the pinned 128x128 tile are box questions -- answered by a short UNTIMED
    # calibration (1 + 2 steps per candidate, max over ranks), like the tile heuristic answers the single-GPU ones.
    calibration = None
    tile_choice = args.cfg if args.cfg >= 0 else None
    ppr_choice = args.panels_per_rank if args.panels_per_rank > 0 else 4
    gather_choice = args.gather if args.gather != "auto" else "collective"
    if world > 1 and (args.panels_per_rank <= 0 or args.gather == "auto"):
        calibration = []
        pprs = [args.panels_per_rank] if args.panels_per_rank > 0 else [4, 8]
        tiles = [tile_choice] + ([-1] if args.cfg < 0 else [])
        gathers = [args.gather] if args.gather != "auto" else ["collective", "p2p"]
        cands = [(ppr, tile, ga) for ga in gathers for tile in tiles for ppr in pprs]
        for ppr, tile, ga in cands:
            rec = {"panels_per_rank": ppr, "tile": "128x128x16 assembly tile pinned" if tile is None else
                   ("heuristic" if tile == -1 else laser_amd.f32_configs()[tile]), "gather": ga}
            ok = 1.0
            try:
                g_, a_, c_ = build(ppr, tile, ga)
                g_.run(a_, B, c_)
                fence()
                t0 = time.perf_counter()
                for _ in range(2):
                    g_.run(a_, B, c_)
                fence()
                dt = (time.perf_counter() - t0) / 2
                del g_, a_, c_
            except Exception as e:       # a candidate one rank cannot run is dropped by every rank (flag all-reduced below)
                ok, dt = 0.0, 1e9
                rec["error"] = f"{type(e).__name__}: {e}"[:200]
            tt = torch.tensor([dt, -ok], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)          # max time; max of -ok = 0 if any rank failed
            rec["ms_per_step"] = round(float(tt[0].item()) * 1e3, 4) if float(tt[1].item()) < 0 else None
            calibration.append(rec)
        good = [i for i in range(len(cands)) if calibration[i]["ms_per_step"] is not None]
        if good:
            best = min(good, key=lambda i: calibration[i]["ms_per_step"])   # identical on every rank (all-reduced values)
            ppr_choice, tile_choice, gather_choice = cands[best]
    sg, A_local, C = build(ppr_choice, tile_choice, gather_choice)
    p = sg.plan

    def step():
        sg.run(A_local, B, C)

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if world == 1 else []
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        if marks:
            marks[i].record()     # per-step marks for mean/min/max/stddev like the reference's printStats
        step()
    if marks:
        marks[-1].record()
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())

    # self-check inside the bench: rows of C against an fp64 product on this GPU -- a few of this rank's own
    # rows and, through the hashed generator, a few rows owned by EVERY other rank (they can only be right
    # if the all-gather delivered them)
    check_rows = []
    for r_ in range(world):
        start, valid = p.panel(p.panels_per_rank - 1, r_)
        check_rows += list(range(start, start + min(valid, 4)))
        start, valid = p.panel(0, r_)
        check_rows += list(range(start + max(0, valid - 4), start + valid))
    ref = hashed(check_rows, K, 1).double() @ B.double()
    err = (C[check_rows].double() - ref).abs().max().item()
    assert err < 1e-4, f"bench self-check failed on rank {rank}: max abs err {err}"

    # N > 1: (i) the same panels WITHOUT the exchange (compute-only: what the step would cost if the gather were free), max over
    # ranks; (ii) the roofline object prices the GEMM kernel alone -- this rank's row share as ONE launch under the same tile pin
    # the sharded run used, HIP events on the launch stream; (iii) who took part: every rank reports its device, so the line shows
    # whether RCCL really spanned N GPUs
    k_ms_multi = compute_only_ms = ranks_seen = None
    if world > 1:
        def pinned(fn):
            prev = laser_amd.get_option("asm_tile")
            if tile_choice is None:
                laser_amd.set_option("asm_tile", SHARDED_ASM_TILE)
            elif tile_choice >= 0:
                laser_amd.set_f32_config(tile_choice)
            try:
                return fn()
            finally:
                laser_amd.set_option("asm_tile", prev)
                laser_amd.set_f32_config(args.cfg)

        def panels_only():
            for s_ in range(p.panels_per_rank):
                start, valid = p.panel(s_, rank)
                if valid > 0:
                    laser_amd.matmul(A_local[s_ * p.rows: s_ * p.rows + valid], B, 1, 0, C[start:start + valid])

        def timed_compute_only():
            panels_only()
            fence()
            t0_ = time.perf_counter()
            for _ in range(args.steps):
                panels_only()
            fence()
            return (time.perf_counter() - t0_) / args.steps * 1e3
        tc = torch.tensor([pinned(timed_compute_only)], dtype=torch.float64, device=dev)
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        compute_only_ms = float(tc.item())

        rows_local = min(n, A_local.shape[0])
        Cs = torch.zeros((rows_local, N), dtype=torch.float32, device=dev)

        def timed_kernel():
            for _ in range(2):
                laser_amd.matmul(A_local[:rows_local], B, 1, 0, Cs)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                laser_amd.matmul(A_local[:rows_local], B, 1, 0, Cs)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.steps, rows_local, last_kernel_name(laser_amd)
        k_ms_multi = pinned(timed_kernel)
        del Cs
        me = {"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(),
              "kernel": k_ms_multi[2], "kernel_ms": round(k_ms_multi[0], 4)}
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)

    if rank == 0:
        flops_step = 2.0 * M_total * N * K
        ms_per_step = wall / args.steps * 1e3
        value = flops_step / (wall / args.steps) / 1e9
        out = {
            "metric": "sgemm GFLOP/s and %MFMA-peak, M=N=K=8192, 1/2/4/8 MI355X",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": (f"fp32 sgemm M=N=K={n} contiguous row-major, alpha=1 beta=0, 1xMI355X (BASELINE configs[1])"
                             if world == 1 else
                             f"fp32 sgemm M={M_total} N={N} K={K} row-panel sharded over {world}xMI355X, "
                             f"RCCL all-gather of C ({'one collective per slab' if gather_choice == 'collective' else 'grouped point-to-point sends'}) "
                             f"inside the timed region (BASELINE configs[4] shape at 8 GPUs)"),
                "M": M_total, "N": N, "K": K, "accumulation": mode,
                "tile_config": (laser_amd.f32_configs()[args.cfg] if args.cfg >= 0 else
                                "heuristic" if (world == 1 or tile_choice == -1) else
                                SHARDED_TILE_NAME + " (pinned for sharded runs: shares the CUs with RCCL)"),
                "parallelism": f"row-panels x{world}" + (f", {sg.plan.panels_per_rank} block-cyclic panels/rank" if world > 1 else ""),
                "pct_of_fp32_mfma_peak": round(100.0 * value / 1e3 / (FP32_MFMA_PEAK_TFLOPS * world), 2),
            },
        }
        if calibration is not None:
            out["config"]["calibration_untimed"] = calibration   # the candidates the warm-up phase tried; the fastest one ran
        if world == 1:
            # one step == one launch of the GEMM kernel on torch's current stream, bracketed by HIP
            # events on that same stream: average launch duration = ev_ms / steps
            k_ms = ev_ms / args.steps
            ach = 2.0 * n * n * n / (k_ms * 1e-3) / 1e12
            per = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
            mean = sum(per) / len(per)
            out["config"]["step_ms_stats"] = {   # benchmarks/gemm/gemm_bench_float32.nim:20-27 (printStats)
                "mean": round(mean, 4), "min": round(min(per), 4), "max": round(max(per), 4),
                "stddev": round((sum((x - mean) ** 2 for x in per) / max(1, len(per) - 1)) ** 0.5, 4)}
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                               "kernel": last_kernel_name(laser_amd),
                               "kernel_ms": round(k_ms, 4), "algorithmic_flops_per_launch": 2.0 * n * n * n}
            plan_now = headline_plan(laser_amd)
            out["roofline"]["launch_plan"] = plan_now
            tr = pmc_traffic(mode, plan_now) if (n == SIZE and args.cfg < 0) else None   # the profiled shape / configuration only
            if tr is not None:
                out["roofline"]["traffic"] = tr["bytes_per_launch"]
                out["roofline"]["traffic_source"] = "committed measurement (not collected in this run): " + tr["source"]
            # the other accumulation mode, same operands, same protocol (reported, not `value`)
            other = 1 if mode == "laser_order" else 0
            laser_amd.set_float_mode(other)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                step()
            e1.record()
            torch.cuda.synchronize()
            o_ms = e0.elapsed_time(e1) / args.steps
            laser_amd.set_float_mode(1 - other)
            o_tf = 2.0 * n * n * n / (o_ms * 1e-3) / 1e12
            out["config"]["other_mode"] = {"accumulation": "fast" if other == 1 else "laser_order",
                                           "ms_per_step": round(o_ms, 4), "gflops": round(o_tf * 1e3, 1),
                                           "frac_mfma_peak": round(o_tf / FP32_MFMA_PEAK_TFLOPS, 4)}
            if not args.no_side_configs:
                out["configs"] = side_configs()
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
        else:
            k_ms, rows_local, k_name = k_ms_multi
            fl = 2.0 * rows_local * N * K
            ach = fl / (k_ms * 1e-3) / 1e12
            # the step, taken apart: the same panels without the exchange, and what the exchange left exposed
            out["config"]["compute_only_ms"] = round(compute_only_ms, 4)
            out["config"]["exposed_gather_ms"] = round(ms_per_step - compute_only_ms, 4)
            out["config"]["compute_only_gflops"] = round(flops_step / (compute_only_ms * 1e-3) / 1e9, 1)
            out["config"]["backend"] = {"name": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": ranks_seen,
                                        "distinct_devices": len({r_["device"] for r_ in ranks_seen})}
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                               "kernel": k_name, "kernel_ms": round(k_ms, 4),
                               "algorithmic_flops_per_launch": fl,
                               "note": "per-GPU kernel alone (rank 0's row share as one launch), timed after the sharded run"}
            # (traffic stays null: the committed PMC passes profiled the single-GPU run's tile configuration)
            if not args.no_cpu_baseline:      # stated beside every line, N > 1 too (the other ranks wait at the barrier below)
                out["cpu_baseline"] = cpu_baseline()
        side = None
        if not args.no_single_process:
            if world == 1:
                try:
                    side = single_process_sharded(1, n, min(args.steps, 5), 1, args.mode)
                except Exception as e:
                    side = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1:
            if side is not None:
                out["single_process_sharded"] = side
            print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            # every rank is past its GPU work: the same job through the C-ABI's single-process entry point, in a child
            # process with a time limit (a hung transport there must never cost the bench line above)
            if not args.no_single_process:
                import subprocess
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GROUP_RANK",
                                                                         "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
                cmd = [sys.executable, os.path.abspath(__file__), "--single-process", str(world), "--size", str(n), "--steps",
                       str(min(args.steps, 5)), "--warmup", "2"] + (["--mode", args.mode] if args.mode else [])
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
                    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    out["single_process_sharded"] = json.loads(lines[-1]) if lines else {"error": (r.stderr or r.stdout)[-400:]}
                except subprocess.TimeoutExpired:
                    out["single_process_sharded"] = {"error": "timed out after 240 s"}
                except Exception as e:
                    out["single_process_sharded"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
