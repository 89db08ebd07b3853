"""X2 tie-guard experiment, run on the GPU box with the library built with / without -DMET2_TIE_GUARD=2:
the 65 536-voxel reference fixture and the 13-voxel fail set against the REFERENCE's own answers, and the share of voxels that took a
refined evaluation (status bit 64)."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import parity_report as pr
PKG = "multicomponent-t2-toolbox_amd"
pkg = importlib.import_module(PKG)
G = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(G, "golden_tail_X2.npz"))
gg = {k: g[k] for k in g.files}
gg["data"] = g["data"].astype(np.float64); gg["lambda_grid"] = np.zeros(50); gg["X2_L2_f"] = g["X2_L2_f"].astype(np.float64)
n = gg["data"].shape[0]
nte, nt2 = gg["data"].shape[1], gg["T2s"].shape[0]
plan = pkg.Met2Plan(nte, nt2, 1)
plan.set_dictionary(gg["D150"][:, :, None]).set_t2_grid(gg["T2s"]).set_penalty("L2", gg["T2s"])       # the reference's own dictionary, as tests/tools/parity_report.py
out = plan.fit("X2", torch.as_tensor(gg["data"], device="cuda"), want_lambda=True)
f = out["fsol"].cpu().numpy() / gg["data"][:, :1]; lam = out["lam"].cpu().numpy(); st = out["status"].cpu().numpy()
s = pr.stats(f, gg["X2_L2_f"], gg["T2s"], lam, gg["X2_L2_lam"])
res = {"defines": os.environ.get("MET2_BUILD_DEFINES", ""), "tail65536_vs_reference": {k: s[k] for k in ("n_over_1e-5", "max", "max_abs_MWF")},
       "refined_voxels": int(((st & 64) != 0).sum()), "refined_frac": float(((st & 64) != 0).mean()),
       "cause_frac": {name: float(((st & (1 << bit)) != 0).mean()) for name, bit in (("|p| vs |q r / 2|", 8), ("p vs q (a - xf)", 9), ("p vs q (b - xf)", 10), ("fx vs fnfc / ffulc", 11), ("fu vs retained", 12))}}
z = np.load(os.path.join(G, "golden_x2_failset.npz"))
plan.build_dictionary_epg(gg["T2s"], 1000.0 * np.ones(nt2), 10.0, np.array([150.0]), 3000.0).set_penalty("L2", gg["T2s"])      # (bench.py's dictionary: the fail set came from it)
o2 = plan.fit("X2", torch.as_tensor(z["data"], device="cuda"), want_lambda=True)
f2 = o2["fsol"].cpu().numpy()
rel = lambda a, b: np.max(np.abs(a - b), axis=1) / np.max(np.abs(b), axis=1)
e_hip = rel(f2, z["ref_f"])
res["failset13"] = {"hip_equals_reference": int((e_hip < 1e-7).sum()), "of": int(e_hip.shape[0]), "refined": int(((o2["status"].cpu().numpy() & 64) != 0).sum()),
                    "round3_hip_equals_reference": int((rel(z["got"], z["ref_f"]) < 1e-7).sum()), "cpu_checker_equals_reference": int((rel(z["ref"], z["ref_f"]) < 1e-7).sum())}
print(json.dumps(res))
