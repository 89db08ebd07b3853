"""Which voxels of the 2k parity case differ from the oracle, and is it a Brent path flip (lambda differs) or a solver error?"""
import importlib, sys
import numpy as np, torch
sys.path.insert(0, ".")
pkg = importlib.import_module("multicomponent-t2-toolbox_amd")
synth = importlib.import_module("multicomponent-t2-toolbox_amd.synth")
from oracle import oracle
oracle.build()
pen = sys.argv[1] if len(sys.argv) > 1 else "I"
nte, nt2, nvox = 32, 60, 2048
T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2); alphas = np.linspace(90.0, 180.0, 91)
plan = pkg.Met2Plan(nte, nt2, 91)
plan.build_dictionary_epg(T2s, T1s, 10.0, alphas, 3000.0).set_penalty(pen, T2s)
data, fa, _ = synth.make_voxels(nvox, nte=nte, seed=11, fa_values=alphas, device="cuda")
out = plan.fit("X2", data, fa_index=fa, want_lambda=True)
D = np.ascontiguousarray(np.transpose(plan.get_dictionary(), (2, 0, 1)))
L = oracle.penalty(nt2, pen, T2s)
d = data.cpu().numpy(); f = fa.cpu().numpy()
fs, sg, rg, st = oracle.fit_batch("X2", D, L, d, f, np.ones(nvox), nthreads=8)
got = out["fsol"].cpu().numpy(); lam = out["lam"].cpu().numpy()
e = np.max(np.abs(got - fs), axis=1) / np.max(np.abs(fs), axis=1)
bad = np.where(e > 1e-5)[0]
print("voxels > 1e-5:", bad, e[bad])
for v in bad[:5]:
    # oracle solution at the GPU's lambda: if it reproduces the GPU spectrum the solver is right and Brent took another path
    Dv = D[int(f[v])]
    x_at = oracle.nnls_tik(Dv, d[v] / d[v, 0], L, lam[v]) * d[v, 0] if hasattr(oracle, "nnls_tik") else None
    print("voxel", v, "gpu lam", lam[v], "gpu k_est", out["reg"][v].item(), "oracle k_est", rg[v],
          "| gpu-vs-oracle(at gpu lam)", None if x_at is None else np.max(np.abs(x_at - got[v])) / np.max(np.abs(got[v])))
