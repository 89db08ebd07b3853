#!/usr/bin/env python3
"""bench.py -- voxels/s of the MI355X T2-spectrum hot path on BASELINE.json's configs.

Default = configs[1]: synthetic 128x128x64 volume (1 048 576 voxels), nTE=32, nT2=60, reg_method=X2, reg_matrix=L2, single
flip angle.  One "step" = one pass of the hot path over one rank's voxels (gates + FA bucketing [+ brute-force FA estimation
when the config has it] + per-voxel solve + metrics epilogue), inputs and outputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W [--config {0,1,2,3,4}] [--scaling {weak,strong}]

  --config   0: 32x32x1 X2/I single FA (the reference's own CPU-runnable case; cpu_baseline also at 1 thread)
             1: 128x128x64 X2/L2 single FA (the metric's config; default)
             2: 200x200x128 L_curve/L1      3: 200x200x128 BayesReg/InvT2
             4: 200x200x128, nTE=48, nT2=120, GCV/L2, FA brute-force over 91 flip angles
  --scaling  weak   (default): every rank owns a volume of the config's size
             strong: ONE volume of the config's size, its voxel list dealt to the ranks in interleaved 4 096-voxel blocks
  N>1: launched by torch.distributed.run, one rank per GPU; the step ends with the path's single collective, the gather of the
  output maps (+ reg_param) on rank 0.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
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


def source_sha():
    """Digest of the kernel sources: counter files under profiles/ are only quoted while they describe this build."""
    h = hashlib.sha256()
    for f in ("met2_hip.hip", "nnls_wave.hpp", "objectives.hpp", "wave_ops.hpp"):
        h.update(open(os.path.join(ROOT, PKG, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


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
        out = oracle.fit_batch(method, D, L, d, fa, np.ones(n), lambda_reg=lam_grid, nthreads=cores)
        return time.time() - t, out

    n0 = min(max(16 * cores, 16), data_cpu.shape[0])
    dt0, _ = run(n0)
    rate = n0 / max(dt0, 1e-6)
    n1 = int(min(data_cpu.shape[0], max(n0, rate * seconds)))
    dt, out = run(n1)
    return {"value": n1 / dt, "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": "first %d voxels of the same volume, %s/%s%s, %.1f s, OpenMP over voxels"
                      % (n1, method, pen, " after brute-force FA" if brute else "", dt)}, (out[0], n1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument("--scaling", type=str, default="weak", choices=["weak", "strong"])
    ap.add_argument("--dims", type=str, default="", help="override the config's volume (marks the line as a variant)")
    ap.add_argument("--method", type=str, default="")
    ap.add_argument("--penalty", type=str, default="")
    ap.add_argument("--nte", type=int, default=0)
    ap.add_argument("--nt2", type=int, default=0)
    ap.add_argument("--fa", type=str, default="", choices=["", "single", "brute-force"],
                    help="single: constant FA 150 deg; brute-force: per-voxel FA drawn from the 91-grid and estimated on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--dump-fail", type=str, default="", help="npz path: inputs/outputs of sample voxels whose fsol is >1e-5 off the oracle")
    args = ap.parse_args()

    cfg = dict(CONFIGS[args.config])
    variant = False
    for k, v in (("method", args.method), ("penalty", args.penalty), ("nte", args.nte), ("nt2", args.nt2), ("fa", args.fa)):
        if v and v != cfg[k]:
            cfg[k] = v; variant = True
    if args.dims:
        d = tuple(int(v) for v in args.dims.split(","))
        variant = variant or d != cfg["dims"]
        cfg["dims"] = d

    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    mdist = importlib.import_module(PKG + ".dist")
    rank, local_rank, world = mdist.init()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # MET2_BENCH_SHARE_GPU=1 (+ MET2_DIST_BACKEND=gloo) rehearses the N>1 code path on a one-GPU box: every
    # rank uses cuda:0 and the collective runs over gloo on host copies.  Never set for measurements.
    if os.environ.get("MET2_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    nx, ny, nz = cfg["dims"]
    nvox_total = nx * ny * nz
    nte, nt2, method, pen = cfg["nte"], cfg["nt2"], cfg["method"], cfg["penalty"]
    T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2)
    brute = cfg["fa"] == "brute-force"
    alphas = np.linspace(90.0, 180.0, 91) if brute else np.array([150.0])   # "single FA" = index 60 of the 91-grid
    plan = pkg.Met2Plan(nte, nt2, alphas.shape[0], device=local_rank)
    plan.build_dictionary_epg(T2s, T1s, 10.0, alphas, 3000.0).set_penalty(pen, T2s)
    strong = args.scaling == "strong"
    # weak: a volume per rank (its own seed).  strong: the one volume (same seed everywhere), this rank's interleaved blocks of it
    data, _, _ = synth.make_voxels(nvox_total, nte=nte, seed=20260102 + (0 if strong else rank), fa_deg=150.0,
                                   fa_values=alphas if brute else None, device=dev)
    if strong and world > 1:
        idx = mdist.shard_indices(nvox_total, rank, world, device=dev)
        data = data[idx].contiguous()
    nvox = data.shape[0]
    out = {k: torch.empty(s, dtype=torch.float64, device=dev) for k, s in
           (("fsol", (nvox, nt2)), ("sig", (nvox, nte)), ("reg", (nvox,)), ("maps", (6, nvox)))}
    out["status"] = torch.empty((nvox,), dtype=torch.int32, device=dev)

    fa_ms = []
    import torch.distributed as dist
    gloo = world > 1 and dist.get_backend() == "gloo"
    counts = [mdist.shard_count(nvox_total, r, world) for r in range(world)] if strong else [nvox] * world
    maxlen = max(counts)
    gather_bufs = None
    send = torch.zeros((maxlen, 7), dtype=torch.float64, device="cpu" if gloo else dev) if world > 1 else None

    def step():
        fa_idx = None
        if brute:       # driver step 2 (motor:349-373) on the device, then step 3+4
            fa_idx, _, _ = plan.fa_bruteforce(data)
            fa_ms.append(plan.last_kernel_ms())
        res = plan.fit(method, data, fa_index=fa_idx, out=out)
        if world > 1:   # the path's single collective: output maps + reg_param, one packed buffer, to the root over xGMI
            nonlocal gather_bufs
            send[:nvox, :6].copy_(res["maps"].t())
            send[:nvox, 6].copy_(res["reg"])
            if rank == 0 and gather_bufs is None:
                gather_bufs = [torch.empty_like(send) for _ in range(world)]
            dist.gather(send, gather_bufs if rank == 0 else None, dst=0)
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
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if gloo else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

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
        traffic = None; valu = None; pmc_note = "no counter file for this workload"
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
                       "sharding": ("one volume, interleaved 4096-voxel blocks over the ranks" if strong else "one volume per rank") if world > 1 else "single GPU",
                       "collective": "one gather of [maps | reg_param] (56 B/voxel) to rank 0" if world > 1 else None},
            "roofline": {"bound": "hbm", "kernel": "fit_kernel<%s>" % method, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel_ms": kms, "second_pass_ms": float(np.mean(pass2_ms)), "bytes_per_voxel": bpv, "counters": pmc_note,
                         "note": "fp64 VALU-issue-bound active-set iteration, not HBM-bound (DESIGN.md section 6)"},
        }
        if valu:
            line["roofline"]["valu"] = valu
        if brute:
            line["roofline"]["fa_kernel_ms"] = float(np.mean(fa_ms[-args.steps:]))
        if not args.no_cpu_baseline:
            cores = host_cores()
            sample = data[: min(nvox, 1 << 17)].cpu().numpy()
            cb, (fs_ref, n1) = cpu_baseline(method, pen, brute, sample, nte, nt2, T2s, T1s, alphas, synth.lambda_grid(), cores, seconds=args.cpu_seconds)
            line["cpu_baseline"] = cb
            if args.config == 0 and cores > 1:      # configs[0] names the reference's 1-core path: the same port on one thread beside it
                cb1, _ = cpu_baseline(method, pen, brute, sample, nte, nt2, T2s, T1s, alphas, synth.lambda_grid(), 1, seconds=min(args.cpu_seconds, 10.0))
                line["cpu_baseline_1core"] = cb1
            got = out["fsol"][:n1].cpu().numpy()
            den = np.max(np.abs(fs_ref), axis=1); den[den == 0] = 1.0
            rel = np.max(np.abs(got - fs_ref), axis=1) / den
            mwf_ref = fs_ref[:, T2s <= 40.0].sum(axis=1) / (fs_ref.sum(axis=1) + 1e-16)
            dm = np.abs(out["maps"][0, :n1].cpu().numpy() - mwf_ref)
            line["parity"] = {"sample": n1, "against": "oracle (pinned to the reference: tests/test_oracle_golden.py, tests/test_tail_parity.py)",
                              "max_rel_fsol": float(rel.max()), "frac_over_1e-5": float((rel > 1e-5).mean()),
                              "max_abs_MWF": float(dm.max()), "median_abs_MWF": float(np.median(dm)), "p99_abs_MWF": float(np.quantile(dm, 0.99))}
            if args.dump_fail:
                bad = np.nonzero(rel > 1e-5)[0]
                np.savez(args.dump_fail, idx=bad, data=data[:n1].cpu().numpy()[bad], got=got[bad], ref=fs_ref[bad],
                         reg=out["reg"][:n1].cpu().numpy()[bad])
        print(json.dumps(line), flush=True)
    plan.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
