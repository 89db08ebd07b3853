#!/usr/bin/env python3
"""bench.py -- voxels/s of the MI355X T2-spectrum hot path on BASELINE.json's configs.

Default = configs[1]: synthetic 128x128x64 volume (1 048 576 voxels), nTE=32, nT2=60, reg_method=X2, reg_matrix=L2, single
flip angle.  One "step" = one pass of the hot path over one rank's voxels (gates + FA bucketing [+ brute-force FA estimation
when the config has it] + per-voxel solve + metrics epilogue), inputs and outputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W [--config {0,1,2,3,4}] [--scaling {weak,strong}] [--gather {maps,all}]

  --config   0: 32x32x1 X2/I single FA (the reference's own CPU-runnable case; cpu_baseline also at 1 thread)
             1: 128x128x64 X2/L2 single FA (the metric's config; default)
             2: 200x200x128 L_curve/L1      3: 200x200x128 BayesReg/InvT2
             4: 200x200x128, nTE=48, nT2=120, GCV/L2, FA brute-force over 91 flip angles
  --scaling  weak   (default): every rank owns a volume of the config's size
             strong: ONE volume of the config's size, its voxel list dealt to the ranks in interleaved 4 096-voxel blocks
  --gather   maps (default): the step's single collective moves the six maps + reg_param (56 B/voxel) to rank 0
             all:  fsol + Est_Signal + reg_param + maps (SURVEY.md section 8e's payload: 792 B/voxel at 32x60)

N > 1: one rank per GPU over RCCL.  Launched by torch.distributed.run (the driver's way) the ranks are already there; started
plainly (`python bench.py --gpus N`) this process starts them itself -- as a CHILD `python -m torch.distributed.run ...` before
anything touches the GPU -- relays the child's JSON line and exits with its code.  Fewer than N visible devices is an error, never
a silent 1-rank run.

Besides `value` (region (i) of SURVEY.md section 8d: device-resident in -> device-resident out) the line carries
`dict_build_ms` (region (ii): EPG dictionary + Gram matrices + seeds, once per run) and `end_to_end` (region (iii): host volume ->
blocks H2D -> fit -> D2H of every output through the C ABI's host entry met2_fit_host, rank 0's volume).
  --driver host: ONE process drives all N devices through that entry (pinned host arrays in and out); see host_driver_main.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PKG = "multicomponent-t2-toolbox_amd"
BYTES_PER_VOXEL = {(32, 60): 1057, (48, 120): 1793}     # compulsory HBM bytes per voxel
HBM_PEAK_GBPS = 8000.0                                   # MI355X_MICROARCH.md: 8 TB/s spec

CONFIGS = {
    0: dict(dims=(32, 32, 1), nte=32, nt2=60, method="X2", penalty="I", fa="single"),
    1: dict(dims=(128, 128, 64), nte=32, nt2=60, method="X2", penalty="L2", fa="single"),
    2: dict(dims=(200, 200, 128), nte=32, nt2=60, method="L_curve", penalty="L1", fa="single"),
    3: dict(dims=(200, 200, 128), nte=32, nt2=60, method="BayesReg", penalty="InvT2", fa="single"),
    4: dict(dims=(200, 200, 128), nte=48, nt2=120, method="GCV", penalty="L2", fa="brute-force"),
}


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return int(os.environ.get("MET2_CPU_THREADS", n))


def source_sha(files=("met2_hip.hip", "fit_kernel.hpp", "nnls_wave.hpp", "nnls_big.hpp", "objectives.hpp", "wave_ops.hpp")):
    """Digest of the kernel sources: counter files under profiles/ are only quoted while they describe this build."""
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, PKG, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


FP64_VECTOR_PEAK_TFLOPS = 78.6       # AMD Instinct MI355X data sheet, peak FP64 vector; = 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz (not in the local guides)
KERNEL_METHOD_ID = {"NNLS": 0, "T2SPARC": 1, "X2": 2, "L_curve": 3, "GCV": 6, "BayesReg": 5}      # fit_kernel<METHOD, NB, SECOND> (GCV: the low-rank form)


def kernel_resources(method, nb):
    """Registers and spills of the dominant fit kernel from the newest profiles/*_kernel_resource_usage.csv whose `# src_sha` line matches the
    sources this run was built from (scripts/resource_usage.py regenerates it without a GPU); None otherwise."""
    import glob
    want = "void fit_kernel<%d, %d, false>(FitArgs)" % (KERNEL_METHOD_ID.get(method, -1), nb)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_resource_usage.csv")), reverse=True):
        try:
            lines = open(f).read().splitlines()
            sha = [l.split(":", 1)[1].strip() for l in lines if l.startswith("# src_sha:")]
            if not sha or sha[0] != source_sha():
                continue
            for l in lines:
                if l.startswith('"' + want + '"'):
                    v = [int(x) for x in l.rsplit('"', 1)[1].strip(", ").split(",")]
                    return {"file": "profiles/" + os.path.basename(f), "vgprs": v[0], "agprs": v[1], "sgprs": v[2], "sgpr_spills": v[3], "vgpr_spills": v[4],
                            "scratch_bytes_per_lane": v[5], "waves_per_simd": v[6]}
        except Exception:
            continue
    return None


def cpu_baseline(method, pen, brute, data_cpu, nte, nt2, T2s, T1s, alphas, lam_grid, cores, seconds=15.0):
    """The oracle ("port": the C restatement in oracle/) timed on this host's cores on a bounded sample of the same voxels
    (the same region the GPU times: FA estimation when the config has it + fit).  Checker code used as a reported baseline only."""
    from oracle import oracle
    oracle.build()
    D = oracle.dictionary_fa_major(nt2, T2s, T1s, nte, 10.0, alphas, 3000.0)
    L = oracle.penalty(nt2, pen, T2s)

    def run(n):
        d = data_cpu[:n]
        t = time.time()
        fa = oracle.fa_bruteforce(D, d, np.ones(n), nthreads=cores)[0] if brute else np.zeros(n)
        out = oracle.fit_batch(method, D, L, d, fa, np.ones(n), lambda_reg=lam_grid, nthreads=cores, want_lambda=True)
        return time.time() - t, out

    n0 = min(max(16 * cores, 16), data_cpu.shape[0])
    dt0, _ = run(n0)
    rate = n0 / max(dt0, 1e-6)
    n1 = int(min(data_cpu.shape[0], max(n0, rate * seconds)))
    dt, out = run(n1)
    return {"value": n1 / dt, "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": "first %d voxels of the same volume, %s/%s%s, %.1f s, OpenMP over voxels"
                      % (n1, method, pen, " after brute-force FA" if brute else "", dt)}, (out[0], out[4], n1)


def cpu_baseline_scipy(method, pen, data_cpu, nte, nt2, T2s, T1s, alphas, cores, nvox=512):
    """A second CPU baseline of the REFERENCE's shape: this repo's own SciPy restatement of the reference's per-voxel Python path
    (scipy.optimize.nnls on the augmented system inside scipy.optimize.fminbound, algorithms.py:211-233) run row by row through
    joblib worker processes as motor:427-441 does.  The reference's files never travel; this is oracle/scipy_restatement.py,
    validated against golden_S1 in the CPU suite.  Single-FA X2 configs only (the reference-shaped leg exists to show what the
    reference's own software stack does per core, not to cover every method)."""
    from oracle import scipy_restatement as sr
    from oracle import oracle
    D = oracle.dictionary_fa_major(nt2, T2s, T1s, nte, 10.0, alphas, 3000.0)[0]
    L = oracle.penalty(nt2, pen, T2s)
    n = min(nvox, data_cpu.shape[0])
    rows = max(cores, 1)
    t = time.time()
    sr.fit_rows(method, D, L, data_cpu[:n], n_rows=rows, n_jobs=cores)
    dt = time.time() - t
    return {"value": n / dt, "unit": "voxels/s", "cores": cores, "kind": "scipy-restatement",
            "sample": "first %d voxels of the same volume, %s/%s, %.1f s, scipy.optimize.nnls + fminbound per voxel, joblib "
                      "(multiprocessing) over %d image rows as motor:435" % (n, method, pen, dt, rows)}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a child torch.distributed.run, BEFORE any GPU call in this
    process (torch.cuda.device_count() does not initialise the GPU on this image), relay its output and return its exit code."""
    share = os.environ.get("MET2_BENCH_SHARE_GPU") == "1" or os.environ.get("MET2_BENCH_PLUMBING") == "1"
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not share:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible; refusing to run fewer ranks than asked\n" % (args.gpus, ndev))
        return 3
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, env=env)
    return p.returncode


def plumbing_main(args, rank, world, mdist):
    """MET2_BENCH_PLUMBING=1: the N-rank harness (spawn, rendezvous, shard sizes, the packed gather, max-over-ranks timing, the JSON
    line) WITHOUT any compute -- every rank fills its send buffer with a rank-dependent pattern and rank 0 checks what arrives.
    For the CPU test of the launcher only; the line says so and carries no throughput."""
    import torch.distributed as dist
    nvox = 10007
    W = 7 if args.gather == "maps" else 99
    counts = [mdist.shard_count(nvox, r, world) for r in range(world)]
    maxlen = max(counts)
    n = counts[rank]
    send = torch.zeros((maxlen, W), dtype=torch.float64)
    idx = mdist.shard_indices(nvox, rank, world)
    send[:n] = idx.to(torch.float64).unsqueeze(1) * 1000.0 + torch.arange(W, dtype=torch.float64).unsqueeze(0)
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1:
            dist.gather(send, bufs, dst=0)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ok = True
    if rank == 0 and world > 1:
        full = torch.empty((nvox, W), dtype=torch.float64)
        for r in range(world):
            full[mdist.shard_indices(nvox, r, world)] = bufs[r][: counts[r]]
        want = torch.arange(nvox, dtype=torch.float64).unsqueeze(1) * 1000.0 + torch.arange(W, dtype=torch.float64).unsqueeze(0)
        ok = bool(torch.equal(full, want))
    if rank == 0:
        print(json.dumps({"metric": "voxels/sec (whole node) at nTE=32, nT2=60; max |MWF-ref|", "value": None, "unit": "voxels/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True,
                          "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "none (launcher plumbing test, no compute)",
                          "config": {"workload": "plumbing test", "ranks_seen": world, "backend": dist.get_backend() if world > 1 else None,
                                     "gather": args.gather, "gather_bytes_per_voxel": 8 * W, "gather_ok": ok}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 4


def end_to_end(plan, pkg, method, data, brute, chunk_vox=262144, reps=3):
    """Region (iii) of SURVEY.md section 8d on this rank, through the C ABI's host entry (met2_fit_host: numpy arrays in and out, the block
    pipeline inside the library, no torch on the call path): the volume in host memory -> blocks H2D -> [FA estimation] -> fit -> D2H of fsol,
    Est_Signal, reg_param and the six maps, copies and kernels overlapped on three streams.  Once with every array in PINNED memory (the DMA
    engines reach the caller's arrays themselves), once with pageable numpy arrays (staged by the library under the fits)."""
    host_mod = importlib.import_module(PKG + ".host")
    host = torch.empty(data.shape, dtype=torch.float64, pin_memory=True)
    host.copy_(data)
    torch.cuda.synchronize()
    nvox, nte = data.shape
    pin = lambda shape, dt=torch.float64: torch.empty(shape, dtype=dt, pin_memory=True).numpy()
    pinned_out = {"fsol": pin((nvox, plan.n_t2)), "sig": pin((nvox, nte)), "reg": pin((nvox,)), "maps": pin((6, nvox)), "status": pin((nvox,), torch.int32),
                  "fa_index": pin((nvox,))}
    res = {}
    ref = None
    for label, src, outs in (("pinned", host.numpy(), pinned_out), ("pageable", np.array(host.numpy()), None)):
        b = None; first = None
        for i in range(reps + 1):
            t0 = time.perf_counter()
            outs = host_mod.fit_host(plan, method, src, estimate_fa=bool(brute), chunk=chunk_vox, out=outs)
            dt = time.perf_counter() - t0
            if i == 0:
                first = dt
            else:
                b = dt if b is None else min(b, dt)
        if ref is None:
            ref = outs
        res[label] = {"ms": 1e3 * b, "first_call_ms": 1e3 * first, "voxels_per_s": nvox / b,
                      "bit_equal_to_the_pinned_run": bool(np.array_equal(outs["fsol"], ref["fsol"], equal_nan=True) and np.array_equal(outs["maps"], ref["maps"], equal_nan=True))}
    nbytes = host.numel() * 8 + sum(int(a.nbytes) for a in pinned_out.values())
    best = res["pinned"]["ms"] * 1e-3
    return {"c_abi_met2_fit_host": res, "ms": res["pinned"]["ms"], "first_call_ms": res["pinned"]["first_call_ms"], "voxels_per_s": res["pinned"]["voxels_per_s"],
            "chunk_voxels": chunk_vox, "host_bytes_moved": nbytes, "pcie_GBps": nbytes / best / 1e9,
            "includes": "host volume -> H2D -> %sfit + metrics -> D2H of fsol, Est_Signal, reg_param, maps (met2_fit_host: three streams, blocks of %d voxels)"
                        % ("brute-force FA -> " if brute else "", chunk_vox)}


def tv_main(args, rank, local_rank, world):
    """--workload tv: the driver's TV step (motor:293-304) on a spatially organised phantom of the config's shape; one step = every echo
    volume through estimate_sigma + denoise_tv_chambolle (weight 2 sigma, eps 2e-4, at most 200 iterations), device-resident in and
    out.  The stencil kernel is HBM-bound: `roofline` prices its 56 algorithmic bytes per (voxel, echo) and iteration."""
    import ctypes as C
    import torch.distributed as dist
    synth = importlib.import_module(PKG + ".synth")
    tv = importlib.import_module(PKG + ".tv")
    lib = importlib.import_module(PKG + "._lib")
    cfg = CONFIGS[args.config]
    dims = tuple(int(v) for v in args.dims.split(",")) if args.dims else cfg["dims"]
    nte = args.nte or cfg["nte"]
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    vol, _ = synth.make_phantom(dims, nte=nte, seed=20260110 + rank, device=dev)
    nvox = int(np.prod(dims))

    def step():
        return tv.tv_chambolle(vol, return_info=True)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    iter_ms = []; launches = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, sig, its = step()
        ms = C.c_double(); nl = C.c_int32()
        lib.check(lib.lib().met2_tv_last_timing(C.byref(ms), C.byref(nl)))
        iter_ms.append(ms.value); launches.append(nl.value)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        it_sum = int(its.sum())
        bytes_iter = 56.0 * nvox * it_sum                     # read p (3) + f (1), write p (3) per active (voxel, echo) and iteration
        ims = float(np.mean(iter_ms))
        achieved = bytes_iter / (ims * 1e-3) / 1e9
        traffic = None; pmc_note = "no counter file for this workload"
        try:
            ent = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json"))).get("tv_%dx%dx%dx%d" % (dims + (nte,)))
            if ent is not None and ent.get("src_sha") == source_sha(("met2_tv.hip",)):
                traffic = ent.get("hbm_bytes_per_launch")
                pmc_note = "profiles/%s (tag %s)" % (ent.get("files"), ent.get("tag"))
        except Exception:
            pass
        line = {"metric": "voxels/sec through the TV denoising step (motor:293-304), all %d echo volumes" % nte,
                "value": world * nvox * args.steps / dt, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic",
                "config": {"workload": "TV denoising of a synthetic %dx%dx%d head phantom, nTE=%d (estimate_sigma + denoise_tv_chambolle, weight 2 sigma, "
                                       "eps 2e-4, <= 200 iterations); not a BASELINE config" % (dims + (nte,)),
                           "voxels_per_gpu": nvox, "ranks_seen": world, "iterations_per_echo": its.tolist(),
                           "sigma_per_echo_first_last": [float(sig[0]), float(sig[-1])], "iterations_enqueued": int(np.mean(launches)),
                           "sharding": "one volume per rank, no collective" if world > 1 else "single GPU"},
                "roofline": {"bound": "hbm", "kernel": "tv_iter_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "counters": pmc_note,
                             "iteration_phase_ms": ims, "launches": int(np.mean(launches)), "active_echo_iterations": it_sum,
                             "bytes_per_voxel_echo_iteration": 56,
                             "note": "achieved = 56 B x voxels x (sum over echoes of the iterations each ran) / HIP-event time from the first to the last "
                                     "Chambolle launch (tv_iter_kernel + the per-iteration tv_reduce_kernel, finished echoes' launches included)"}}
        if not args.no_cpu_baseline:
            from oracle import tv_oracle
            host = vol.cpu().numpy()
            t1 = time.time(); ne = 0; o0 = None; n0 = 0; s0 = 0.0; its_cpu = []
            while ne < nte and time.time() - t1 < args.cpu_seconds:          # whole echo volumes until the time budget is used
                v = np.ascontiguousarray(host[..., ne])
                sg = tv_oracle.estimate_sigma(v)
                o, n = tv_oracle.denoise_tv_chambolle(v, 2.0 * sg, return_iters=True)
                if ne == 0:
                    o0, n0, s0 = o, n, sg
                its_cpu.append(n); ne += 1
            d1 = time.time() - t1
            line["cpu_baseline"] = {"value": nvox * ne / d1 / nte, "unit": "voxels/s", "cores": 1, "kind": "port",
                                    "sample": "echo volumes 0..%d of the same phantom through the numpy restatement (oracle/tv_oracle.py), %d Chambolle iterations "
                                              "in all, %.1f s; voxel-echoes/s divided by the %d echoes of the metric's unit" % (ne - 1, sum(its_cpu), d1, nte)}
            g0 = out[..., 0].cpu().numpy()
            line["parity"] = {"against": "numpy restatement of scikit-image's functions (PARITY UNPINNED: scikit-image is not in the image)",
                              "echo": 0, "max_rel": float(np.max(np.abs(g0 - o0)) / np.max(np.abs(o0))), "iterations_hip_oracle": [int(its[0]), int(n0)],
                              "sigma_rel": float(abs(sig[0] / s0 - 1.0))}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def host_driver_main(args):
    """--driver host: ONE process drives N devices through the C ABI's host entry (met2_fit_host: one plan per device, one host thread per
    plan inside the call, the voxel list dealt in runs of 4 096 voxels, every device's copies over its own PCIe link, no collective) -- the shape
    of the reference's single Python process (motor:427-441).  Host arrays in PINNED memory in and out; `value` is host-to-host wall clock, so
    PCIe is inside it (the torch.distributed driver's `value` is device-resident).  weak: a volume of N x the config's voxels; strong: one volume."""
    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    host_mod = importlib.import_module(PKG + ".host")
    cfg = dict(CONFIGS[args.config])
    variant = False
    for k, v in (("method", args.method), ("penalty", args.penalty), ("nte", args.nte), ("nt2", args.nt2), ("fa", args.fa)):
        if v and v != cfg[k]:
            cfg[k] = v; variant = True
    if args.dims:
        d = tuple(int(v) for v in args.dims.split(","))
        variant = variant or d != cfg["dims"]
        cfg["dims"] = d
    N = args.gpus
    share = os.environ.get("MET2_BENCH_SHARE_GPU") == "1"
    ndev = torch.cuda.device_count()
    if ndev < N and not share:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible; refusing to run fewer devices than asked\n" % (N, ndev))
        return 3
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    devices = [0] * N if share else list(range(N))
    nx, ny, nz = cfg["dims"]
    nte, nt2, method, pen = cfg["nte"], cfg["nt2"], cfg["method"], cfg["penalty"]
    strong = args.scaling == "strong"
    nvox = nx * ny * nz * (1 if strong else N)
    T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2)
    brute = cfg["fa"] == "brute-force"
    alphas = np.linspace(90.0, 180.0, 91) if brute else np.array([150.0])
    plans = []
    t0 = time.perf_counter()
    for dv in devices:
        pl = pkg.Met2Plan(nte, nt2, alphas.shape[0], device=dv)
        pl.build_dictionary_epg(T2s, T1s, 10.0, alphas, 3000.0).set_penalty(pen, T2s).set_t2_grid(T2s)
        plans.append(pl)
    for dv in set(devices):
        torch.cuda.synchronize(dv)
    dict_build_ms = 1e3 * (time.perf_counter() - t0) / N
    # the volume, generated on device 0 in pieces of the config's size and kept in pinned host memory
    host = torch.empty((nvox, nte), dtype=torch.float64, pin_memory=True)
    per = nx * ny * nz
    for i in range(nvox // per):
        d_i, _, _ = synth.make_voxels(per, nte=nte, seed=20260102 + i, fa_deg=150.0, fa_values=alphas if brute else None, device="cuda:0")
        host[i * per:(i + 1) * per].copy_(d_i)
        del d_i
    torch.cuda.synchronize(0)
    pin = lambda shape, dt=torch.float64: torch.empty(shape, dtype=dt, pin_memory=True).numpy()
    outs = {"fsol": pin((nvox, nt2)), "sig": pin((nvox, nte)), "reg": pin((nvox,)), "lam": pin((nvox,)), "maps": pin((6, nvox)),
            "status": pin((nvox,), torch.int32), "fa_index": pin((nvox,))}
    src = host.numpy()
    chunk = int(os.environ.get("MET2_BENCH_HOST_CHUNK", "0"))

    def step():
        return host_mod.fit_host(plans, method, src, estimate_fa=bool(brute), chunk=chunk, want_lambda=True, out=outs)

    for _ in range(args.warmup):
        step()
    plan_ms = []; kernel_ms = []; second_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        plan_ms.append(res["plan_ms"].copy())
        kernel_ms.append([pl.last_kernel_ms() for pl in plans]); second_ms.append([pl.last_second_pass_ms() for pl in plans])
    dt = time.perf_counter() - t0
    pm = np.mean(plan_ms, axis=0)
    fitted = int((res["status"] > 0).sum())
    bpv = BYTES_PER_VOXEL.get((nte, nt2), 8 * (2 * nte + nt2) + 65)
    achieved = fitted * bpv * args.steps / dt / 1e9
    moved = nvox * (8 * nte + 8 * (nt2 + nte + 2 + 6) + 4 + 8)
    workload = "%s: synthetic %s%dx%dx%d volume, nTE=%d, nT2=%d, reg_method=%s, reg_matrix=%s, %s" % (
        ("configs[%d]" % args.config) if not variant else "variant of configs[%d]" % args.config, ("%d x " % N) if (not strong and N > 1) else "", nx, ny, nz, nte, nt2, method, pen,
        "FA brute-force over 91 flip angles" if brute else "single FA (150 deg)")
    line = {
        "metric": "voxels/sec (whole node) at nTE=32, nT2=60; max |MWF-ref|",
        "value": nvox * args.steps / dt, "unit": "voxels/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "driver": "host: one process, met2_fit_host with one plan per device (C ABI 6), pinned host arrays in and out",
                   "voxels": nvox, "fitted_voxels": fitted, "devices": devices, "ranks_seen": 1, "backend": None,
                   "sharding": "runs of 4096 voxels dealt round-robin to the plans, blocks of %s voxels per DMA" % (chunk or "the library's default number of"),
                   "collective": None, "host_bytes_moved_per_step": moved},
        "dict_build_ms": dict_build_ms,
        "host_driver": {"plan_ms_per_device": [float(v) for v in pm], "plan_ms_min": float(pm.min()), "plan_ms_max": float(pm.max()),
                        "plan_ms_spread": float((pm.max() - pm.min()) / pm.max()) if pm.max() > 0 else 0.0,
                        "last_block_kernel_ms_per_device": [float(v) for v in np.mean(kernel_ms, axis=0)],
                        "last_block_spill_kernel_ms_per_device": [float(v) for v in np.mean(second_ms, axis=0)],
                        "pcie_GBps_whole_job": moved * args.steps / dt / 1e9,
                        "note": "plan_ms: wall ms of every plan's host thread inside the call (uploads, FA step, fits, downloads of its share); "
                                "value is host-to-host, PCIe inside"},
        "roofline": {"bound": "hbm", "kernel": "fit_kernel<%s>" % method, "achieved": achieved, "peak": HBM_PEAK_GBPS * N, "unit": "GB/s",
                     "frac": achieved / (HBM_PEAK_GBPS * N), "traffic": None, "kernel_ms": None, "bytes_per_voxel": bpv,
                     "note": "algorithmic bytes of the fitted voxels over the host-to-host wall time of the whole call (all devices); the kernel's own "
                             "roofline is in the default driver's line"},
    }
    # the binding roofline, as far as this driver can see it: k measured on the outputs, the dominant kernel's registers; the counter-derived rate
    # belongs to one launch on resident data and is in the default driver's line
    mean_k = float((res["fsol"] > 0).sum()) / max(fitted, 1)
    info = plans[0].launch_info(method)
    line["roofline_fp64"] = {"bound": "fp64 vector issue", "peak": FP64_VECTOR_PEAK_TFLOPS * N, "unit": "TFLOP/s", "achieved": None, "frac": None,
                             "waves_per_simd_launched": info["block"] // 64 / 4.0, "mean_final_passive_set_k": mean_k,
                             "live_lane_frac_position_phases": mean_k / 64.0,
                             "registers": kernel_resources(method, 2 if nt2 > 64 else 1) or "no profiles/*_kernel_resource_usage.csv for these sources",
                             "note": "achieved (issued fp64 FLOP/s of one launch on device-resident data) is in the default driver's line: python bench.py"}
    print(json.dumps(line), flush=True)
    for pl in plans:
        pl.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", type=str, default="fit", choices=["fit", "tv"],
                    help="fit (default): BASELINE.json's metric; tv: the driver's TV denoising step on a phantom of the config's shape")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--driver", type=str, default="dist", choices=["dist", "host"],
                    help="dist (default): one process per GPU (torch.distributed / RCCL), device-resident volume, one gather; host: ONE process, "
                         "met2_fit_host with one plan per device on pinned host arrays (the reference's single-process shape), no collective")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument("--scaling", type=str, default="weak", choices=["weak", "strong"])
    ap.add_argument("--gather", type=str, default="maps", choices=["maps", "all"])
    ap.add_argument("--dims", type=str, default="", help="override the config's volume (marks the line as a variant)")
    ap.add_argument("--method", type=str, default="")
    ap.add_argument("--penalty", type=str, default="")
    ap.add_argument("--nte", type=int, default=0)
    ap.add_argument("--nt2", type=int, default=0)
    ap.add_argument("--fa", type=str, default="", choices=["", "single", "brute-force"],
                    help="single: constant FA 150 deg; brute-force: per-voxel FA drawn from the 91-grid and estimated on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true", help="N > 1: run rank 0's CPU baseline all the same (default there: skipped, the other ranks would idle)")
    ap.add_argument("--end-to-end", action="store_true", help="N > 1: run rank 0's host-to-host leg all the same")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--parity-sample", type=int, default=1 << 17, help="upper bound on the voxels the oracle is run on for the parity block")
    ap.add_argument("--dump-fail", type=str, default="", help="npz path: inputs/outputs of sample voxels whose fsol is >1e-5 off the oracle")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1:       # the CPU baseline and the host-to-host leg are rank 0's alone (15-25 s with N - 1 GPUs idle): single-GPU runs report them
        args.no_cpu_baseline = not args.cpu_baseline
        args.no_end_to_end = not args.end_to_end

    if args.driver == "host":                # one process for all devices: nothing is spawned, and nothing has touched a GPU before this point
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--driver host is ONE process for all devices: start it plainly, not under torch.distributed.run")
        sys.exit(host_driver_main(args))

    # ---- N > 1 without a torchrun environment: this process only starts the ranks (no GPU call before this point)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))

    cfg = dict(CONFIGS[args.config])
    variant = False
    for k, v in (("method", args.method), ("penalty", args.penalty), ("nte", args.nte), ("nt2", args.nt2), ("fa", args.fa)):
        if v and v != cfg[k]:
            cfg[k] = v; variant = True
    if args.dims:
        d = tuple(int(v) for v in args.dims.split(","))
        variant = variant or d != cfg["dims"]
        cfg["dims"] = d

    mdist = importlib.import_module(PKG + ".dist")
    plumbing = os.environ.get("MET2_BENCH_PLUMBING") == "1"
    rank, local_rank, world = mdist.init(backend="gloo" if plumbing else None)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if plumbing:
        sys.exit(plumbing_main(args, rank, world, mdist))
    if args.workload == "tv":
        if os.environ.get("MET2_BENCH_SHARE_GPU") == "1":
            local_rank = 0
        sys.exit(tv_main(args, rank, local_rank, world))
    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # MET2_BENCH_SHARE_GPU=1 (+ MET2_DIST_BACKEND=gloo) rehearses the N>1 code path on a one-GPU box: every
    # rank uses cuda:0 and the collective runs over gloo on host copies.  Never set for measurements.
    share = os.environ.get("MET2_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("WORLD_SIZE=%d but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    nx, ny, nz = cfg["dims"]
    nvox_total = nx * ny * nz
    nte, nt2, method, pen = cfg["nte"], cfg["nt2"], cfg["method"], cfg["penalty"]
    T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2)
    brute = cfg["fa"] == "brute-force"
    alphas = np.linspace(90.0, 180.0, 91) if brute else np.array([150.0])   # "single FA" = index 60 of the 91-grid
    plan = pkg.Met2Plan(nte, nt2, alphas.shape[0], device=local_rank)
    # region (ii): dictionary (EPG) + Gram matrices + penalty + plan-level seeds, once per run; timed on the second build
    # (the first one also pays for loading the code object)
    plan.build_dictionary_epg(T2s, T1s, 10.0, alphas, 3000.0).set_penalty(pen, T2s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.build_dictionary_epg(T2s, T1s, 10.0, alphas, 3000.0).set_penalty(pen, T2s)
    torch.cuda.synchronize()
    dict_build_ms = 1e3 * (time.perf_counter() - t0)
    strong = args.scaling == "strong"
    # weak: a volume per rank (its own seed).  strong: the one volume (same seed everywhere), this rank's interleaved blocks of it
    data, _, _ = synth.make_voxels(nvox_total, nte=nte, seed=20260102 + (0 if strong else rank), fa_deg=150.0,
                                   fa_values=alphas if brute else None, device=dev)
    if strong and world > 1:
        idx = mdist.shard_indices(nvox_total, rank, world, device=dev)
        data = data[idx].contiguous()
    nvox = data.shape[0]
    out = {k: torch.empty(s, dtype=torch.float64, device=dev) for k, s in
           (("fsol", (nvox, nt2)), ("sig", (nvox, nte)), ("reg", (nvox,)), ("lam", (nvox,)), ("maps", (6, nvox)))}
    out["status"] = torch.empty((nvox,), dtype=torch.int32, device=dev)

    fa_ms = []
    import torch.distributed as dist
    backend = dist.get_backend() if world > 1 else None
    gloo = backend == "gloo"
    counts = [mdist.shard_count(nvox_total, r, world) for r in range(world)] if strong else [nvox] * world
    maxlen = max(counts)
    gather_all = args.gather == "all"
    WIDTH = {"maps": 7, "all": nt2 + nte + 1 + 6}
    W = WIDTH[args.gather]
    gdev = "cpu" if gloo else dev
    sends = {}; gbufs = {}

    def do_gather(res, kind):
        """The path's single collective: ONE packed buffer per rank to the root (direct peer -> root over xGMI; gloo moves host copies)."""
        if kind not in sends:
            sends[kind] = torch.zeros((maxlen, WIDTH[kind]), dtype=torch.float64, device=gdev)
            gbufs[kind] = [torch.empty_like(sends[kind]) for _ in range(world)] if rank == 0 else None
        send = sends[kind]
        if kind == "all":
            send[:nvox, :nt2].copy_(res["fsol"])
            send[:nvox, nt2:nt2 + nte].copy_(res["sig"])
            send[:nvox, nt2 + nte].copy_(res["reg"])
            send[:nvox, nt2 + nte + 1:].copy_(res["maps"].t())
        else:
            send[:nvox, :6].copy_(res["maps"].t())
            send[:nvox, 6].copy_(res["reg"])
        dist.gather(send, gbufs[kind], dst=0)

    gather_ms = []

    class GatherTimer:
        """ms of one packed gather: HIP events on the launch stream under RCCL, host clock under gloo (the collective runs on the host there)."""
        def __enter__(self):
            if gloo:
                torch.cuda.synchronize(); self.t0 = time.perf_counter()
            else:
                self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); self.e0.record()
            return self
        def __exit__(self, *a):
            if gloo:
                self.t1 = time.perf_counter()
            else:
                self.e1.record()
        def ms(self):
            return 1e3 * (self.t1 - self.t0) if gloo else self.e0.elapsed_time(self.e1)

    def step():
        fa_idx = None
        if brute:       # driver step 2 (motor:349-373) on the device, then step 3+4
            fa_idx, _, _ = plan.fa_bruteforce(data)
            fa_ms.append(plan.last_kernel_ms())
        res = plan.fit(method, data, fa_index=fa_idx, out=out, want_lambda=True)
        if world > 1:
            with GatherTimer() as gt:
                do_gather(res, args.gather)
            gather_ms.append(gt)
        return res

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    kernel_ms = []
    pass2_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(plan.last_kernel_ms())
        pass2_ms.append(plan.last_second_pass_ms())
    sync()
    spill_voxels = plan.last_spill_count()
    dt = time.perf_counter() - t0
    multi = None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=gdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # per-rank numbers of the timed steps: the solver kernel and the collective (packing + gather, HIP events on the launch stream)
        g_timed = float(np.mean([g_.ms() for g_ in gather_ms[-args.steps:]]))
        mine = torch.tensor([float(np.mean(kernel_ms)), g_timed], dtype=torch.float64, device=gdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        # both payloads once more, untimed for `value`: the maps' 56 B/voxel and SURVEY section 8e's full 8 (nT2 + nTE + 7) B/voxel in the same run
        each = {}
        for kind in ("maps", "all"):
            do_gather(out, kind)                       # (allocates the buffers of the kind the timed steps did not use)
            sync()
            with GatherTimer() as gt:
                do_gather(out, kind)
            sync()
            t1 = torch.tensor([gt.ms()], dtype=torch.float64, device=gdev)
            dist.all_reduce(t1, op=dist.ReduceOp.MAX)
            each[kind] = {"ms_max_over_ranks": float(t1.item()), "bytes_per_voxel": 8 * WIDTH[kind], "bytes_per_rank": 8 * WIDTH[kind] * maxlen}
        multi = {"kernel_ms_per_rank": [float(v) for v in allr[:, 0]], "kernel_ms_min": float(allr[:, 0].min()), "kernel_ms_max": float(allr[:, 0].max()),
                 "gather_ms_in_timed_steps_per_rank": [float(v) for v in allr[:, 1]], "gather_ms": each}

    if rank == 0:
        fitted = int((out["status"] > 0).sum().item())
        units = nvox_total if strong else world * nvox
        value = units * args.steps / dt
        bpv = BYTES_PER_VOXEL.get((nte, nt2), 8 * (2 * nte + nt2) + 65)
        kms = float(np.mean(kernel_ms))
        achieved = fitted * bpv / (kms * 1e-3) / 1e9
        # counters (HBM bytes from FETCH/WRITE_SIZE, SQ instruction counts) come from separate rocprofv3 --pmc passes of this
        # very command, stored by scripts/collect_pmc.py with the digest of the kernel sources they were taken on; they are
        # quoted only while that digest matches the sources this run was built from
        traffic = None; valu = None; ent = None; pmc_note = "no counter file for this workload"
        key = "config%d" % args.config
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json")))
            ent = pmc.get(key) if not variant else None
            if ent is not None and ent.get("src_sha") == source_sha() and ent.get("voxels") == nvox:
                traffic = ent.get("hbm_bytes_per_launch")
                pmc_note = "profiles/%s (tag %s, sources %s)" % (ent.get("files", "pmc_counters.json"), ent.get("tag"), ent.get("src_sha"))
                if ent.get("valu_insts_per_launch"):
                    n_valu = float(ent["valu_insts_per_launch"])
                    simds = 4 * torch.cuda.get_device_properties(0).multi_processor_count
                    valu = {"valu_insts_per_launch": n_valu, "valu_insts_per_voxel": n_valu / max(fitted, 1), "simds": simds, "clock_ghz": 2.4,
                            "valu_issue_frac": n_valu * 4.0 / (simds * kms * 1e-3 * 2.4e9),
                            "note": "4 cycles per wave64 VALU instruction and a 2.4 GHz clock assumed"}
                    for k2 in ("salu_insts_per_launch", "lds_insts_per_launch", "mfma_insts_per_launch"):
                        if ent.get(k2) is not None:
                            valu[k2] = ent[k2]
            elif ent is not None:
                pmc_note = "profiles/pmc_counters.json is stale for this build (sources %s, counters taken on %s): dropped" % (source_sha(), ent.get("src_sha"))
        except Exception:
            pass
        workload = "%s: synthetic %dx%dx%d volume, nTE=%d, nT2=%d, reg_method=%s, reg_matrix=%s, %s" % (
            ("configs[%d]" % args.config) if not variant else "variant of configs[%d]" % args.config, nx, ny, nz, nte, nt2, method, pen,
            "FA brute-force over 91 flip angles" if brute else "single FA (150 deg)")
        line = {
            "metric": "voxels/sec (whole node) at nTE=32, nT2=60; max |MWF-ref|",
            "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "voxels_per_gpu": nvox, "fitted_voxels_per_gpu": fitted,
                       "ranks_seen": world, "backend": ("nccl (RCCL)" if backend == "nccl" else backend),
                       "sharding": ("one volume, interleaved 4096-voxel blocks over the ranks" if strong else "one volume per rank") if world > 1 else "single GPU",
                       "gather": args.gather, "gather_bytes_per_voxel": 8 * W,
                       "collective": ("one gather of %s (%d B/voxel) to rank 0" % ("[fsol | Est_Signal | reg_param | maps]" if gather_all else "[maps | reg_param]", 8 * W)) if world > 1 else None},
            "dict_build_ms": dict_build_ms,
            "roofline": {"bound": "hbm", "kernel": "fit_kernel<%s>" % method, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel_ms": kms, "second_pass_ms": float(np.mean(pass2_ms)), "spill_voxels": spill_voxels, "bytes_per_voxel": bpv, "counters": pmc_note,
                         "note": "fp64 VALU-issue-bound active-set iteration, not HBM-bound (DESIGN.md section 6)"},
        }
        # ---- the roofline that binds: fp64 vector issue.  Counter-derived numbers are quoted only while the counter file describes this build
        # (sha-gated like `traffic`); k and the launch's waves per SIMD are measured live.
        info = plan.launch_info(method)
        waves_per_simd = info["block"] // 64 / 4.0
        # the passive set of a voxel's final solve = its positive bins: counted on the outputs (a device-side counter cost the X2 kernel its L2-resident
        # scratch: 112 -> 120 bytes per lane and 1.4x -> 3.5x of the algorithmic HBM bytes, profiles/r05_ab.txt)
        mean_k = float((out["fsol"] > 0).sum().item()) / max(fitted, 1)
        fp = {"bound": "fp64 vector issue", "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
              "peak_source": "AMD Instinct MI355X data sheet, peak FP64 vector (= 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz); not in /opt/skills/guides",
              "achieved": None, "frac": None, "fp64_share_of_valu_insts": None, "simd_busy_frac": None,
              "waves_per_simd_launched": waves_per_simd, "mean_final_passive_set_k": mean_k, "live_lane_frac_position_phases": mean_k / 64.0,
              "registers": kernel_resources(method, 2 if nt2 > 64 else 1) or "no profiles/*_kernel_resource_usage.csv for these sources (python3 scripts/resource_usage.py)",
              "counters": pmc_note,
              "note": "achieved = (2 FMA + MUL + ADD fp64 instructions) x 64 lanes / kernel time: ISSUED lanes, of which about live_lane_frac carry a passive bin in the "
                      "position-indexed phases; simd_busy = SQ_ACTIVE_INST_VALU x waves per SIMD / SQ_WAVE_CYCLES"}
        try:
            if ent is not None and ent.get("src_sha") == source_sha() and ent.get("voxels") == nvox and ent.get("valu_fma_f64") is not None:
                flops = 64.0 * (2.0 * ent["valu_fma_f64"] + ent.get("valu_mul_f64", 0.0) + ent.get("valu_add_f64", 0.0))
                fp["achieved"] = flops / (kms * 1e-3) / 1e12
                fp["frac"] = fp["achieved"] / FP64_VECTOR_PEAK_TFLOPS
                f64 = ent["valu_fma_f64"] + ent.get("valu_mul_f64", 0.0) + ent.get("valu_add_f64", 0.0) + ent.get("valu_trans_f64", 0.0)
                fp["fp64_share_of_valu_insts"] = f64 / ent["valu_insts_per_launch"]
                if ent.get("sq_wave_cycles"):
                    fp["simd_busy_frac"] = ent["sq_active_inst_valu"] * waves_per_simd / ent["sq_wave_cycles"]
        except Exception:
            pass
        line["roofline_fp64"] = fp
        if multi:
            line["multi_gpu"] = multi
        if valu:
            line["roofline"]["valu"] = valu
        if brute:
            line["roofline"]["fa_kernel_ms"] = float(np.mean(fa_ms[-args.steps:]))
        if not args.no_cpu_baseline:
            cores = host_cores()
            sample = data[: min(nvox, args.parity_sample)].cpu().numpy()
            cb, (fs_ref, lam_ref, n1) = cpu_baseline(method, pen, brute, sample, nte, nt2, T2s, T1s, alphas, synth.lambda_grid(), cores, seconds=args.cpu_seconds)
            line["cpu_baseline"] = cb
            if args.config == 0 and cores > 1:      # configs[0] names the reference's 1-core path: the same port on one thread beside it
                cb1, _ = cpu_baseline(method, pen, brute, sample, nte, nt2, T2s, T1s, alphas, synth.lambda_grid(), 1, seconds=min(args.cpu_seconds, 10.0))
                line["cpu_baseline_1core"] = cb1
            if method == "X2" and not brute:
                try:
                    line["cpu_baseline_scipy"] = cpu_baseline_scipy(method, pen, sample, nte, nt2, T2s, T1s, alphas, cores)
                except Exception as e:          # joblib or scipy missing on the box: say so, the port baseline stands
                    line["cpu_baseline_scipy"] = {"value": None, "kind": "scipy-restatement", "error": repr(e)[:200]}
            got = out["fsol"][:n1].cpu().numpy()
            den = np.max(np.abs(fs_ref), axis=1); den[den == 0] = 1.0
            rel = np.max(np.abs(got - fs_ref), axis=1) / den
            mwf_ref = fs_ref[:, T2s <= 40.0].sum(axis=1) / (fs_ref.sum(axis=1) + 1e-16)
            dm = np.abs(out["maps"][0, :n1].cpu().numpy() - mwf_ref)
            inside = rel <= 1e-5
            line["parity"] = {"sample": n1, "against": "oracle (pinned to the reference: tests/test_oracle_golden.py, tests/test_tail_parity.py; the reference "
                                                        "itself leaves the oracle at 3e-5 of 65 536 voxels, profiles/parity_r03.json)",
                              "max_rel_fsol": float(rel.max()), "frac_over_1e-5": float((rel > 1e-5).mean()), "n_over_1e-5": int((rel > 1e-5).sum()),
                              "max_abs_MWF": float(dm.max()), "max_abs_MWF_within_tol": float(dm[inside].max()) if inside.any() else None,
                              "median_abs_MWF": float(np.median(dm)), "p99_abs_MWF": float(np.quantile(dm, 0.99))}
            if args.dump_fail:
                bad = np.nonzero(rel > 1e-5)[0]
                np.savez(args.dump_fail, idx=bad, data=data[:n1].cpu().numpy()[bad], got=got[bad], ref=fs_ref[bad],
                         reg=out["reg"][:n1].cpu().numpy()[bad], lam_hip=out["lam"][:n1].cpu().numpy()[bad], lam_oracle=lam_ref[bad],
                         n_sample=n1, src_sha=source_sha())
        if not args.no_end_to_end:
            try:
                line["end_to_end"] = end_to_end(plan, pkg, method, data, brute)
            except Exception as e:
                line["end_to_end"] = {"error": repr(e)[:300]}
        print(json.dumps(line), flush=True)
    plan.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
