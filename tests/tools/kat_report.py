#!/usr/bin/env python3
"""Statistical known-answer run (tests/kat.py) of all ten methods of the reference's Monte-Carlo table through the HIP path
(or, with --oracle, the CPU oracle): prints one JSON line per method with MAE(MWF) and mean lambda next to the reference's
committed values and the 4-standard-error tolerance.   python tests/tools/kat_report.py [-n 20000] [--oracle] [--out FILE]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import kat  # noqa: E402

PKG = "multicomponent-t2-toolbox_amd"


def run(n, use_oracle, seed=20260131, rows=None, fa_step=0.05):
    synth = importlib.import_module(PKG + ".synth")
    import torch
    dev = "cpu" if use_oracle else "cuda"
    T2s = kat.t2_grid(); T1s = 1000.0 * np.ones(60); alphas = np.linspace(90.0, 180.0, 91)
    fine = np.linspace(90.0, 180.0, int(round(90.0 / fa_step)) + 1)         # the script draws FA ~ U(90, 180); here from a fine grid
    data, _, par = synth.make_voxels(n, nte=32, seed=seed, fa_values=fine, device=dev)
    tm = kat.true_mwf(par, T2s)
    out = []
    if use_oracle:
        from oracle import oracle
        oracle.build()
        nthr = os.cpu_count() or 1
        D = oracle.dictionary_fa_major(60, T2s, T1s, 32, 10.0, alphas, 3000.0)
        d = data.numpy()
        idx = oracle.fa_bruteforce(D, d, np.ones(n), nthreads=nthr)[0]
    else:
        pkg = importlib.import_module(PKG)
        plan = pkg.Met2Plan(32, 60, 91)
        plan.build_dictionary_epg(T2s, T1s, 10.0, alphas, 3000.0)
        plan.set_lambda_grid(kat.kat_lambda_grid())
        idx, _, _ = plan.fa_bruteforce(data)
    for label, meth, pen, mae_ref, lam_ref, lam_sd in (rows or kat.REF_ROWS):
        t = time.time()
        if use_oracle:
            L = oracle.penalty(60, pen, T2s)
            fs, _, _, _, lam = oracle.fit_batch(meth, D, L, d, idx, np.ones(n), lambda_reg=kat.kat_lambda_grid(), nthreads=nthr, want_lambda=True)
            mwf = fs[:, T2s <= 40.0].sum(axis=1) / fs.sum(axis=1)
        else:
            plan.set_penalty(pen, T2s)
            o = plan.fit(meth, data, fa_index=idx, want_lambda=True, want_sig=False)
            mwf = o["maps"][0].cpu().numpy(); lam = o["lam"].cpu().numpy()
        mae = float(np.mean(np.abs(mwf - tm)))
        out.append({"method": label, "n": n, "mae": mae, "mae_ref": mae_ref, "mae_tol": kat.tolerance(label, n, "mae"),
                    "mean_lambda": float(lam.mean()), "mean_lambda_ref": lam_ref, "lambda_tol": kat.tolerance(label, n, "lam"),
                    "seconds": time.time() - t, "path": "oracle" if use_oracle else "hip"})
    if not use_oracle:
        plan.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=20000)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = run(a.n, a.oracle)
    for r in res:
        print(json.dumps(r), flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
