"""GPU parity at shape S2 (nTE=48, nT2=120: two T2 bins per lane, D/B read from L2) against the
reference goldens of tests/golden/golden_S2.npz -- the shape of BASELINE.json's config 5."""
import importlib

import numpy as np
import pytest

from conftest import relmax, relmax_rows

pytestmark = pytest.mark.gpu
PKG = "multicomponent-t2-toolbox_amd"
TOL = 1e-5


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available()
    from oracle import oracle
    oracle.build()
    return importlib.import_module(PKG)


@pytest.fixture(scope="module")
def plan91(pkg, gS2):
    g = gS2
    plan = pkg.Met2Plan(int(g["nte"]), int(g["npc"]), 91)
    plan.build_dictionary_epg(g["T2s"], g["T1s"], float(g["tau"]), g["alpha_values"], float(g["TR"]))
    plan.set_lambda_grid(g["lambda_grid"])
    return plan


def _single(pkg, g, pen):
    plan = pkg.Met2Plan(int(g["nte"]), int(g["npc"]), 1)
    plan.set_dictionary(np.ascontiguousarray(g["D150"][:, :, None])).set_t2_grid(g["T2s"]).set_penalty(g["L_" + pen])
    plan.set_lambda_grid(g["lambda_grid"])
    return plan


def test_epg_dictionary_s2(pkg, gS2, plan91):
    g = gS2
    p = pkg.Met2Plan(int(g["nte"]), int(g["npc"]), g["fa_sel"].shape[0])
    p.build_dictionary_epg(g["T2s"], g["T1s"], float(g["tau"]), g["fa_sel"], float(g["TR"]))
    assert relmax(p.get_dictionary(), g["Dic_sel"]) < 1e-12
    assert relmax(plan91.get_dictionary()[:, :, 60], g["D150"]) < 1e-12


@pytest.mark.parametrize("pen", ["I", "L1", "L2", "InvT2"])
def test_x2_lcurve_nnls_tik_s2(pkg, gS2, pen):
    import torch
    g = gS2
    plan = _single(pkg, g, pen)
    data = torch.as_tensor(g["data"], device="cuda")
    km = g["data"][:, :1]
    out = plan.fit("X2", data, want_lambda=True)
    assert np.max(relmax_rows(out["fsol"].cpu().numpy() / km, g["x2_f_" + pen])) < TOL
    assert np.allclose(out["reg"].cpu().numpy(), g["x2_kest_" + pen], rtol=1e-5)
    assert np.allclose(out["lam"].cpu().numpy(), g["x2_lam_" + pen], rtol=1e-4, atol=1e-9)
    out = plan.fit("L_curve", data)
    assert np.array_equal(out["reg"].cpu().numpy(), g["lc_lam_" + pen])
    assert np.max(relmax_rows(out["fsol"].cpu().numpy() / km, g["lc_f_" + pen])) < TOL
    out = plan.fit("NNLS", data)
    assert np.max(relmax_rows(out["fsol"].cpu().numpy() / km, g["nnls_x"])) < TOL
    for i, lam in enumerate(g["tik_lams"]):
        plan.set_options(t2sparc_lambda=float(lam))
        out = plan.fit("T2SPARC", data)
        assert np.max(relmax_rows(out["fsol"].cpu().numpy() / km, g["tik_" + pen][i])) < TOL


@pytest.mark.parametrize("pen", ["I", "L1", "L2", "InvT2"])
def test_bayesreg_s2(pkg, gS2, pen):
    import torch
    g = gS2
    plan = _single(pkg, g, pen)
    nv = g["bayes_lam_" + pen].shape[0]
    out = plan.fit("BayesReg", torch.as_tensor(g["data"][:nv], device="cuda"), want_lambda=True)
    f = out["fsol"].cpu().numpy() / g["data"][:nv, :1]
    e = relmax_rows(f, g["bayes_f_" + pen])
    if pen == "L2":
        assert np.all(out["lam"].cpu().numpy() == 1.9999959949686712)
    assert e.max() < (2e-4 if pen == "InvT2" else TOL), e
    T2s = g["T2s"]
    mwf = lambda x: x[:, T2s <= 40.0].sum(axis=1) / (x.sum(axis=1) + 1e-16)
    assert np.max(np.abs(mwf(f) - mwf(g["bayes_f_" + pen]))) < TOL


@pytest.mark.parametrize("pen", ["I", "L2"])
def test_gcv_s2(pkg, gS2, pen):
    # config 5's method at config 5's shape: objective goldens + near-optimality + MWF agreement
    import torch
    from oracle import oracle
    g = gS2
    plan = _single(pkg, g, pen)
    lams = g["obj_grid"]
    got = plan.objective_grid("GCV", torch.as_tensor(g["data"][:2], device="cuda"), lams).cpu().numpy()
    d = np.abs(got - g["gcvobj_" + pen])
    assert np.median(d) < 1e-3 and np.max(d) < 0.3, (got, g["gcvobj_" + pen])
    plan.set_lambda_grid(g["lambda_grid"])
    nv = g["gcv_lam_" + pen].shape[0]
    out = plan.fit("GCV", torch.as_tensor(g["data"][:nv], device="cuda"), want_lambda=True)
    assert not (out["status"].cpu().numpy() & 32).any()
    lam = out["lam"].cpu().numpy()
    f = out["fsol"].cpu().numpy() / g["data"][:nv, :1]
    M = g["data"] / g["data"][:, :1]
    T2s = g["T2s"]
    mwf = lambda x: x[T2s <= 40.0].sum() / (x.sum() + 1e-16)
    for v in range(nv):
        o2 = oracle.objective("GCV", g["D150"], M[v], g["L_" + pen], np.array([lam[v], g["gcv_lam_" + pen][v]]))
        print("MEASURED gcv_s2 %s v=%d dobj=%.3e dMWF=%.2e" % (pen, v, o2[0] - o2[1], abs(mwf(f[v]) - mwf(g["gcv_f_" + pen][v]))))
        assert o2[0] <= o2[1] + 0.12, (v, o2)          # measured max 1.1e-5; 0.12 allows one step of the rank staircase
        assert abs(mwf(f[v]) - mwf(g["gcv_f_" + pen][v])) < 1e-3        # measured max 7.2e-5


def test_fa_and_rows_s2(pkg, gS2, plan91):
    # F1 at S2 + V1 rows with per-voxel FA and gating, every method
    import torch
    g = gS2
    fa, km, _ = plan91.fa_bruteforce(torch.as_tensor(g["fa_data"], device="cuda"))
    assert np.array_equal(fa.cpu().numpy(), g["fa_idx"])
    assert np.allclose(km.cpu().numpy(), g["fa_km"], rtol=1e-7)
    data = torch.as_tensor(g["row_data"], device="cuda")
    fai = torch.as_tensor(g["row_fa_index"], device="cuda")
    msk = torch.as_tensor(g["row_mask"], device="cuda")
    for meth, pen, tol in (("NNLS", "I", TOL), ("X2", "L2", TOL), ("X2", "I", TOL), ("L_curve", "L1", TOL), ("BayesReg", "InvT2", 2e-4),
                           ("BayesReg", "L2", TOL)):
        plan91.set_penalty(g["L_" + pen])
        out = plan91.fit(meth, data, fa_index=fai, mask=msk)
        key = "row_%s_%s_" % (meth, pen)
        fs = out["fsol"].cpu().numpy()
        assert np.max(relmax_rows(fs, g[key + "fsol"])) < tol, (meth, pen)
        assert np.max(relmax_rows(out["sig"].cpu().numpy(), g[key + "sig"])) < tol
        for v in (2, 5, 7):
            assert not fs[v].any()


def test_s2_vs_oracle_maps(pkg, gS2, plan91):
    # metrics epilogue with two bins per lane (32/36/52 bins in the three windows at nT2=120)
    import torch
    from oracle import oracle
    synth = importlib.import_module(PKG + ".synth")
    g = gS2
    nvox = 256
    data, fa, _ = synth.make_voxels(nvox, nte=48, seed=31, fa_values=g["alpha_values"], device="cuda")
    plan91.set_penalty("L2")
    out = plan91.fit("X2", data, fa_index=fa)
    D = np.ascontiguousarray(np.transpose(plan91.get_dictionary(), (2, 0, 1)))
    fs, sg, rg, st = oracle.fit_batch("X2", D, oracle.penalty(120, "L2"), data.cpu().numpy(), fa.cpu().numpy(), np.ones(nvox), nthreads=8)
    e = relmax_rows(out["fsol"].cpu().numpy(), fs)
    print("MEASURED s2_maps n_over=%d of %d max=%.2e p99=%.2e" % (int((e >= TOL).sum()), nvox, e.max(), np.quantile(e, 0.99)))
    assert e.max() < 1e-6, (e.max(), np.quantile(e, 0.99))           # measured max 3.8e-8, none of 256 over 1e-5
    mo = oracle.metrics(fs, g["T2s"], np.ones(nvox))
    maps = out["maps"].cpu().numpy()
    ok = e < TOL
    for i, name in enumerate(pkg.MAP_NAMES):
        assert np.max(np.abs(maps[i][ok] - mo[name][ok])) / max(1.0, np.max(np.abs(mo[name]))) < TOL, name
    m2 = plan91.metrics(out["fsol"]).cpu().numpy()
    assert np.allclose(m2, maps, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("meth,pen", [("X2", "L2"), ("L_curve", "L1"), ("GCV", "L2"), ("BayesReg", "InvT2")])
def test_capacity_ladder_leaves_no_voxel_behind_s2(pkg, gS2, meth, pen):
    # At two bins per lane the first pass runs at the capacity that lets eight waves share the LDS (71 of 120 bins); voxels whose
    # passive set outgrows it leave the pass at once and are solved again by the clean-up passes (GCV: 116, then 120; the others:
    # 120).  8 192 synthetic voxels are enough for a few hundred to take that route: none may keep MET2_ST_KOVERFLOW, every one must
    # be a KKT point of its own lambda, and the result must not depend on the voxel's place in the list (the passes re-sort it).
    import torch
    synth = importlib.import_module(PKG + ".synth")
    g = gS2
    plan = _single(pkg, g, pen)
    nvox = 8192
    data, _, _ = synth.make_voxels(nvox, nte=int(g["nte"]), seed=4242, device="cuda")
    out = plan.fit(meth, data, want_lambda=True)
    st = out["status"].cpu().numpy()
    assert (st & 1).all() and not (st & 32).any(), np.unique(st)
    assert plan.last_spill_count() > 0 or meth == "BayesReg"                                 # (round 5: those voxels go on in the spill-over slots, one launch)
    f = out["fsol"].cpu().numpy()
    assert np.isfinite(f).all() and (f >= 0).all()
    perm = torch.randperm(nvox, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    out2 = plan.fit(meth, data[perm].contiguous(), want_lambda=True)
    assert torch.equal(out2["fsol"], out["fsol"][perm]) and torch.equal(out2["lam"], out["lam"][perm])
