#!/usr/bin/env python3
"""bench.py -- voxels/s of the MI355X T2-spectrum hot path on BASELINE.json's configs[1]:
synthetic 128x128x64 volume (1 048 576 voxels), nTE=32, nT2=60, reg_method=X2, reg_matrix=L2,
single flip angle.  One "step" = one pass of met2_fit (gates + FA bucketing + per-voxel X2 solve +
metrics epilogue) over one rank's volume, inputs and outputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU; weak scaling -- every rank owns a
   volume of the same size; the step ends with the single gather of the output maps on rank 0)

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
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


def cpu_baseline(method, pen, data_cpu, nte, nt2, T2s, T1s, alphas, lam_grid, seconds=15.0):
    """The oracle ("port": the C restatement in oracle/) timed on this host's cores on a bounded
    sample of the same voxels.  Checker code used as a reported baseline only."""
    from oracle import oracle
    oracle.build()
    cores = host_cores()
    D = oracle.dictionary_fa_major(nt2, T2s, T1s, nte, 10.0, alphas, 3000.0)
    L = oracle.penalty(nt2, pen, T2s)
    n0 = min(256 * cores, data_cpu.shape[0])
    t = time.time()
    oracle.fit_batch(method, D, L, data_cpu[:n0], np.zeros(n0), np.ones(n0), lambda_reg=lam_grid, nthreads=cores)
    rate = n0 / max(time.time() - t, 1e-6)
    n1 = int(min(data_cpu.shape[0], max(n0, rate * seconds)))
    t = time.time()
    fs, sg, rg, st = oracle.fit_batch(method, D, L, data_cpu[:n1], np.zeros(n1), np.ones(n1), lambda_reg=lam_grid, nthreads=cores)
    dt = time.time() - t
    return {"value": n1 / dt, "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": "first %d voxels of the same volume, %s/%s, %.1f s, OpenMP over voxels" % (n1, method, pen, dt)}, (fs, n1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dims", type=str, default="128,128,64")
    ap.add_argument("--method", type=str, default="X2")
    ap.add_argument("--penalty", type=str, default="L2")
    ap.add_argument("--nte", type=int, default=32)
    ap.add_argument("--nt2", type=int, default=60)
    ap.add_argument("--fa", type=str, default="single", choices=["single", "brute-force"],
                    help="single: constant FA 150 deg (configs[1]); brute-force: per-voxel FA drawn from the 91-grid and estimated on the device (config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--dump-fail", type=str, default="", help="npz path: inputs/outputs of sample voxels whose fsol is >1e-5 off the oracle")
    args = ap.parse_args()

    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    mdist = importlib.import_module(PKG + ".dist")
    rank, local_rank, world = mdist.init()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # MET2_BENCH_SHARE_GPU=1 (+ MET2_DIST_BACKEND=gloo) rehearses the N>1 code path on a one-GPU box: every
    # rank uses cuda:0 and the collective runs over gloo on host copies.  Never set for measurements.
    share = os.environ.get("MET2_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    nx, ny, nz = (int(v) for v in args.dims.split(","))
    nvox = nx * ny * nz
    nte, nt2 = args.nte, args.nt2
    T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2)
    brute = args.fa == "brute-force"
    alphas = np.linspace(90.0, 180.0, 91) if brute else np.array([150.0])   # "single FA" = index 60 of the 91-grid
    plan = pkg.Met2Plan(nte, nt2, alphas.shape[0], device=local_rank)
    plan.build_dictionary_epg(T2s, T1s, 10.0, alphas, 3000.0).set_penalty(args.penalty, T2s)
    data, fa_true, _ = synth.make_voxels(nvox, nte=nte, seed=20260102 + rank, fa_deg=150.0, fa_values=alphas if brute else None, device=dev)
    out = {k: torch.empty(s, dtype=torch.float64, device=dev) for k, s in
           (("fsol", (nvox, nt2)), ("sig", (nvox, nte)), ("reg", (nvox,)), ("maps", (6, nvox)))}
    out["status"] = torch.empty((nvox,), dtype=torch.int32, device=dev)

    fa_ms = []

    def step():
        fa_idx = None
        if brute:       # driver step 2 (motor:349-373) on the device, then step 3+4
            fa_idx, _, _ = plan.fa_bruteforce(data)
            fa_ms.append(plan.last_kernel_ms())
        res = plan.fit(args.method, data, fa_index=fa_idx, out=out)
        if world > 1:   # the path's single collective: output maps to the root over xGMI
            _gather(res["maps"])
        return res

    gather_bufs = None

    def _gather(maps):
        nonlocal gather_bufs
        import torch.distributed as dist
        if dist.get_backend() == "gloo":
            maps = maps.cpu()
        if rank == 0 and gather_bufs is None:
            gather_bufs = [torch.empty_like(maps) for _ in range(world)]
        dist.gather(maps, gather_bufs if rank == 0 else None, dst=0)

    def sync():
        if world > 1:
            import torch.distributed as dist
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
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        fitted = int((out["status"] > 0).sum().item())
        value = world * nvox * args.steps / dt
        bpv = BYTES_PER_VOXEL.get((nte, nt2), 8 * (2 * nte + nt2) + 65)
        kms = float(np.mean(kernel_ms))
        achieved = fitted * bpv / (kms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("%s_%s_bytes_per_launch" % (args.method, args.penalty)) if (nte, nt2, nvox) == (32, 60, 1048576) else None
            except Exception:
                traffic = None
        # the resource that actually binds: fp64 VALU issue.  Instructions per launch from the committed PMC pass of this
        # workload (SQ_INSTS_VALU; one wave64 VALU instruction = 4 cycles of a 16-lane SIMD), time measured live.
        valu = None
        sqfile = os.path.join(ROOT, "profiles", "r01j_x2l2_pmc_sq.csv")
        if os.path.exists(sqfile) and (nte, nt2, nvox, args.method, args.penalty, brute) == (32, 60, 1048576, "X2", "L2", False):
            try:
                n_valu = [float(l.split(",")[-1]) for l in open(sqfile) if ",SQ_INSTS_VALU," in l][0]
                simds = 4 * torch.cuda.get_device_properties(0).multi_processor_count
                valu = {"valu_insts_per_launch": n_valu, "simds": simds, "clock_ghz": 2.4,
                        "valu_issue_frac": n_valu * 4.0 / (simds * kms * 1e-3 * 2.4e9)}
            except Exception:
                valu = None
        line = {
            "metric": "voxels/sec (whole node) at nTE=32, nT2=60; max |MWF-ref|",
            "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: synthetic %dx%dx%d volume, nTE=%d, nT2=%d, reg_method=%s, reg_matrix=%s, %s"
                                   % ("configs[1]" if (nte, nt2, args.method, args.penalty, brute) == (32, 60, "X2", "L2", False) else "variant",
                                      nx, ny, nz, nte, nt2, args.method, args.penalty,
                                      "FA brute-force over 91 flip angles" if brute else "single FA (150 deg)"),
                       "voxels_per_gpu": nvox, "fitted_voxels_per_gpu": fitted, "sharding": "voxel blocks, one per rank"},
            "roofline": {"bound": "hbm", "kernel": "fit_kernel<%s>" % args.method, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel_ms": kms, "second_pass_ms": float(np.mean(pass2_ms)), "bytes_per_voxel": bpv,
                         "note": "fp64 VALU-issue-bound active-set iteration (70 % of VALU issue slots, profiles/r01j_x2l2_pmc_sq.csv), not HBM-bound (DESIGN.md section 6)"},
        }
        if valu:
            line["roofline"]["valu"] = valu
        if brute:
            line["roofline"]["fa_kernel_ms"] = float(np.mean(fa_ms[-args.steps:]))
        if not args.no_cpu_baseline and not brute:
            cb, (fs_ref, n1) = cpu_baseline(args.method, args.penalty, data[: min(nvox, 1 << 17)].cpu().numpy(), nte, nt2, T2s, T1s,
                                            alphas, synth.lambda_grid(), seconds=args.cpu_seconds)
            line["cpu_baseline"] = cb
            got = out["fsol"][:n1].cpu().numpy()
            den = np.max(np.abs(fs_ref), axis=1); den[den == 0] = 1.0
            rel = np.max(np.abs(got - fs_ref), axis=1) / den
            mwf_ref = fs_ref[:, T2s <= 40.0].sum(axis=1) / (fs_ref.sum(axis=1) + 1e-16)
            line["parity"] = {"sample": n1, "max_rel_fsol": float(rel.max()), "frac_over_1e-5": float((rel > 1e-5).mean()),
                              "max_abs_MWF": float(np.max(np.abs(out["maps"][0, :n1].cpu().numpy() - mwf_ref)))}
            if args.dump_fail:
                bad = np.nonzero(rel > 1e-5)[0]
                np.savez(args.dump_fail, idx=bad, data=data[:n1].cpu().numpy()[bad], got=got[bad], ref=fs_ref[bad],
                         reg=out["reg"][:n1].cpu().numpy()[bad])
        print(json.dumps(line), flush=True)
    plan.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
