#!/usr/bin/env python3
"""Parity rates on the large reference fixtures (tests/golden/golden_tail_{S1,S2}.npz: a few thousand voxels through the
reference's own functions).  For every (method, penalty) pair the three comparisons

    reference <-> oracle     (CPU, runs anywhere)
    reference <-> HIP        (needs the GPU)
    oracle    <-> HIP        (needs the GPU)

are reduced to {n, frac_over_1e-5, p50, p99, max (per-voxel max-norm relative error of fsol), max_abs_MWF, lambda
agreement}.  Output: one JSON document (profiles/parity_r02.json is a committed run of this script on the GPU box).

    python tests/tools/parity_report.py [--out FILE] [--cpu-only]
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
PKG = "multicomponent-t2-toolbox_amd"

PAIRS = {"S1": [("NNLS", "I"), ("X2", "L2"), ("X2", "I"), ("L_curve", "L1"), ("BayesReg", "InvT2"), ("BayesReg", "I"), ("GCV", "L2")],
         "S2": [("X2", "L2"), ("L_curve", "L1"), ("BayesReg", "InvT2"), ("GCV", "L2")],
         "S2b": [("BayesReg", "I")]}


def rel_rows(a, b):
    den = np.max(np.abs(b), axis=1)
    den = np.where(den > 0, den, 1.0)
    return np.max(np.abs(a - b), axis=1) / den


def mwf_of(f, T2s, cut=40.0):
    return f[:, T2s <= cut].sum(axis=1) / (f.sum(axis=1) + 1e-16)


def stats(fa, fb, T2s, lam_a=None, lam_b=None):
    r = rel_rows(fa, fb)
    d = np.abs(mwf_of(fa, T2s) - mwf_of(fb, T2s))
    out = {"n": int(r.shape[0]), "frac_over_1e-5": float((r > 1e-5).mean()), "n_over_1e-5": int((r > 1e-5).sum()),
           "p50": float(np.quantile(r, 0.5)), "p99": float(np.quantile(r, 0.99)), "max": float(r.max()),
           "max_abs_MWF": float(d.max()), "p99_abs_MWF": float(np.quantile(d, 0.99)), "median_abs_MWF": float(np.median(d))}
    if lam_a is not None and lam_b is not None:
        rl = np.abs(lam_a - lam_b) / np.maximum(np.abs(lam_b), 1e-300)
        out["lambda_identical_frac"] = float((lam_a == lam_b).mean())
        out["lambda_rel_p99"] = float(np.quantile(rl, 0.99))
    return out


def oracle_fit(oracle, g, meth, pen, n):
    D = g["D150"][None, :, :]
    L = oracle.penalty(int(g["npc"]), pen, g["T2s"])
    data = g["data"][:n]
    fs, sg, rg, st, lam = oracle.fit_batch(meth, D, L, data, np.zeros(n), np.ones(n), lambda_reg=g["lambda_grid"], nthreads=os.cpu_count() or 1,
                                           want_lambda=True)
    return fs / data[:, :1], lam


def hip_fit(pkg, torch, g, meth, pen, n):
    nte, npc = int(g["nte"]), int(g["npc"])
    plan = pkg.Met2Plan(nte, npc, 1)
    plan.set_dictionary(g["D150"][:, :, None]).set_t2_grid(g["T2s"])
    plan.set_penalty(pen, g["T2s"])
    plan.set_lambda_grid(g["lambda_grid"])
    data = torch.as_tensor(g["data"][:n], device="cuda")
    out = plan.fit(meth, data, want_lambda=True)
    f = out["fsol"].cpu().numpy() / g["data"][:n, :1]
    lam = out["lam"].cpu().numpy()
    plan.close()
    return f, lam


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--cpu-only", action="store_true")
    args = ap.parse_args()
    from oracle import oracle
    oracle.build()
    gpu = False
    if not args.cpu_only:
        import torch
        gpu = torch.cuda.is_available()
        if gpu:
            pkg = importlib.import_module(PKG)
    doc = {"tolerance": 1e-5, "relative": "per-voxel max-norm: max_j |a_j - b_j| / max_j |b_j| on the first-echo-normalised spectrum",
           "fixtures": {}, "rows": []}
    for tag, pairs in PAIRS.items():
        path = os.path.join(GOLDEN, "golden_tail_%s.npz" % tag)
        g = np.load(path)
        doc["fixtures"][tag] = {"file": os.path.relpath(path, ROOT), "nte": int(g["nte"]), "npc": int(g["npc"]), "voxels": int(g["data"].shape[0])}
        for meth, pen in pairs:
            base = "%s_%s" % (meth, pen)
            fref = g[base + "_f"]
            n = fref.shape[0]
            lref = g[base + "_lam"] if (base + "_lam") in g.files else None
            fo, lo = oracle_fit(oracle, g, meth, pen, n)
            row = {"shape": tag, "method": meth, "penalty": pen,
                   "reference_vs_oracle": stats(fo, fref, g["T2s"], lo if lref is not None else None, lref)}
            if gpu:
                fh, lh = hip_fit(pkg, torch, g, meth, pen, n)
                row["reference_vs_hip"] = stats(fh, fref, g["T2s"], lh if lref is not None else None, lref)
                row["oracle_vs_hip"] = stats(fh, fo, g["T2s"], lh, lo)
            doc["rows"].append(row)
            print(json.dumps(row), flush=True)
    # ---- the 65 536-voxel X2/L2 fixture (configs[1]'s method; make_goldens.py tailX2: inputs float32-representable, reference spectra
    #      stored as float32) and the reference run on the voxels where HIP and oracle disagree (make_goldens.py x2fail)
    #      (make_goldens.py tailX2S2: the same at 48 x 120, 8 192 voxels -- the two-bins-per-lane kernels)
    for shape, fname in (("X2tail", "golden_tail_X2.npz"), ("X2tailS2", "golden_tail_X2_S2.npz")):
        path = os.path.join(GOLDEN, fname)
        if not os.path.exists(path):
            continue
        g0 = np.load(path)
        g = {k: g0[k] for k in g0.files}
        g["data"] = g0["data"].astype(np.float64); g["lambda_grid"] = np.zeros(50); fref = g0["X2_L2_f"].astype(np.float64)
        n = fref.shape[0]
        doc["fixtures"][shape] = {"file": os.path.relpath(path, ROOT), "nte": int(g0["nte"]), "npc": int(g0["npc"]), "voxels": int(n),
                                  "note": "p50/p99 of ~3e-8 are the float32 storage of the reference spectra"}
        fo, lo = oracle_fit(oracle, g, "X2", "L2", n)
        row = {"shape": shape, "method": "X2", "penalty": "L2", "reference_vs_oracle": stats(fo, fref, g["T2s"], lo, g["X2_L2_lam"])}
        if gpu:
            fh, lh = hip_fit(pkg, torch, g, "X2", "L2", n)
            row["reference_vs_hip"] = stats(fh, fref, g["T2s"], lh, g["X2_L2_lam"])
            row["oracle_vs_hip"] = stats(fh, fo, g["T2s"], lh, lo)
        doc["rows"].append(row)
        print(json.dumps(row), flush=True)
    path = os.path.join(GOLDEN, "golden_x2_failset.npz")
    if os.path.exists(path):
        z = np.load(path)
        rel = lambda a, b: np.max(np.abs(a - b), axis=1) / np.max(np.abs(b), axis=1)
        e_hip, e_or = rel(z["got"], z["ref_f"]), rel(z["ref"], z["ref_f"])
        doc["x2_failset"] = {"file": os.path.relpath(path, ROOT), "sample_voxels": int(z["n_sample"]), "kernel_sources": str(z["src_sha"]),
                             "voxels_where_hip_and_oracle_differ_by_more_than_1e-5": int(z["idx"].shape[0]),
                             "reference_agrees_with_hip": int((e_hip < 1e-9).sum()), "reference_agrees_with_oracle": int((e_or < 1e-9).sum()),
                             "reference_agrees_with_neither": int(((e_hip >= 1e-9) & (e_or >= 1e-9)).sum()),
                             "max_abs_lambda_difference": float(np.max(np.abs(z["lam_hip"] - z["lam_oracle"]))), "brent_xtol": 1e-5,
                             "table": [{"voxel": int(i), "lambda_reference": float(a), "lambda_hip": float(b), "lambda_oracle": float(c),
                                        "rel_hip_vs_reference": float(d), "rel_oracle_vs_reference": float(e)}
                                       for i, a, b, c, d, e in zip(z["idx"], z["ref_lam"], z["lam_hip"], z["lam_oracle"], e_hip, e_or)]}
        doc["x2_failset"]["note"] = ("reference_agrees_with_hip / lambda_hip / rel_hip_vs_reference describe the ROUND-3 kernel whose disagreement with the oracle "
                                     "defined this set (stored in the fixture); `current_kernel` below is the library as built now")
        if gpu:      # the library as it is now, on the same 13 voxels (round 4: Brent's near-ties decided on refined objective values)
            synth = importlib.import_module(PKG + ".synth")
            nte, npc = z["data"].shape[1], z["ref_f"].shape[1]
            T2s = synth.t2_grid(npc)
            plan = pkg.Met2Plan(nte, npc, 1)
            plan.build_dictionary_epg(T2s, 1000.0 * np.ones(npc), 10.0, np.array([150.0]), 3000.0).set_penalty("L2", T2s)
            out = plan.fit("X2", torch.as_tensor(z["data"], device="cuda"), want_lambda=True)
            e_now = rel(out["fsol"].cpu().numpy(), z["ref_f"])
            lam_now = out["lam"].cpu().numpy()
            plan.close()
            doc["x2_failset"]["current_kernel"] = {"reference_agrees_with_hip": int((e_now < 1e-7).sum()), "of": int(e_now.shape[0]),
                                                   "max_abs_lambda_difference_to_reference": float(np.max(np.abs(lam_now - z["ref_lam"]))),
                                                   "rel_hip_vs_reference": [float(v) for v in e_now]}
        print(json.dumps({k: v for k, v in doc["x2_failset"].items() if k != "table"}), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(doc, f, indent=1)
    return doc


if __name__ == "__main__":
    main()
