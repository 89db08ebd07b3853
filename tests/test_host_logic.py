"""CPU-only tests: the C-ABI library loads and exports every symbol include/met2_hip.h declares
(no compute without a GPU), host-side helpers, the driver-level oracle pipeline against the
reference's end-to-end goldens, and the N>1 sharding path over gloo."""
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import torch
import pytest

from conftest import GOLDEN, ROOT, relmax

PKG = "multicomponent-t2-toolbox_amd"


def test_library_builds_and_exports_header_symbols():
    b = importlib.import_module(PKG + "._build")
    b.build()
    lib = importlib.import_module(PKG + "._lib")
    L = lib.lib()
    header = open(os.path.join(ROOT, "include", "met2_hip.h")).read()
    declared = set(re.findall(r"\b(met2_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), "libmet2_hip.so does not export %s" % name
    assert set(lib.SYMBOLS) == declared
    assert L.met2_abi_version() == 6


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pkg = importlib.import_module(PKG)
    with pytest.raises(pkg.Met2Error):
        pkg.Met2Plan(32, 60, 1)


def test_product_does_not_import_oracle():
    # the oracle is test infrastructure: nothing under the package, the public header, the root alias module or scripts/
    # may import or load it (only tests/, __graft_entry__.smoke() and bench.py's CPU-baseline leg do)
    roots = [os.path.join(ROOT, PKG), os.path.join(ROOT, "include"), os.path.join(ROOT, "scripts")]
    files = [os.path.join(ROOT, "met2_amd.py")]
    for r in roots:
        for dirpath, _, fs in os.walk(r):
            files += [os.path.join(dirpath, f) for f in fs if f.endswith((".py", ".hip", ".hpp", ".h", ".sh"))]
    for f in files:
        src = open(f).read()
        assert "oracle" not in src.replace("test infrastructure", ""), f


def test_motor_helpers_match_golden(gS1):
    motor = importlib.import_module(PKG + ".motor")
    for order, name in ((0, "I"), (1, "L1"), (2, "L2")):
        assert np.array_equal(motor.create_Laplacian_matrix(60, order), gS1["L_" + name])
    assert np.array_equal(motor.create_InvT2_matrix(gS1["T2s"]), gS1["L_InvT2"])
    synth = importlib.import_module(PKG + ".synth")
    assert np.allclose(synth.t2_grid(60), gS1["T2s"], rtol=1e-15)
    assert np.allclose(synth.lambda_grid(), gS1["lambda_grid"], rtol=1e-15)


@pytest.mark.parametrize("tag,meth,pen", [("x2_l2_bf", "X2", "L2")])
def test_oracle_pipeline_vs_driver_golden(oracle, tag, meth, pen):
    # M1 + F1 + V1 chained as the driver does (motor:336-472), against the reference's own end-to-end run
    g = np.load(os.path.join(GOLDEN, "golden_motor_%s.npz" % tag))
    data = g["data"]; mask = g["mask"]
    shp = mask.shape
    nt = data.shape[-1]
    d2 = (data * mask[..., None]).reshape(-1, nt)
    d2 = np.where(d2 < 0, 0.0, d2)
    m1 = (mask.reshape(-1) > 0).astype(float)
    T2s = np.logspace(1, np.log10(2000.0), 60); T1s = 1000.0 * np.ones(60); alphas = np.linspace(90.0, 180.0, 91)
    D = oracle.dictionary_fa_major(60, T2s, T1s, nt, 10.0, alphas, 3000.0)
    idx, km, sse, f = oracle.fa_bruteforce(D, d2, m1, nthreads=4)
    fitted = (m1 > 0) & (d2.sum(axis=1) > 0)
    assert np.array_equal(np.where(fitted, alphas[idx.astype(int)], 0.0).reshape(shp), g["FA"])
    L = oracle.penalty(60, pen, T2s)
    fs, sg, rg, st = oracle.fit_batch(meth, D, L, d2, idx, m1, nthreads=4)
    assert relmax(fs.reshape(shp + (60,)), g["fsol_4D"]) < 1e-8
    assert relmax(sg.reshape(shp + (nt,)), g["Est_Signal"]) < 1e-8
    assert np.allclose(rg.reshape(shp), g["reg_param"], rtol=1e-7, atol=1e-12)
    maps = oracle.metrics(fs, T2s, m1)
    for name in ("MWF", "IEWF", "FWF", "T2_M", "T2_IE", "TWC"):
        assert np.allclose(maps[name].reshape(shp), g[name], rtol=1e-8, atol=1e-12), name


def test_oracle_spline_pipeline_vs_driver_golden(oracle):
    # spline FA (fa_estimation.py:35-70, the CLI default) -> L-curve/L1 -> metrics, against the reference's own run
    g = np.load(os.path.join(GOLDEN, "golden_motor_lcurve_l1_spline.npz"))
    data = g["data"]; mask = g["mask"]
    shp = mask.shape; nt = data.shape[-1]
    d2 = (data * mask[..., None]).reshape(-1, nt)
    d2 = np.where(d2 < 0, 0.0, d2)
    m1 = (mask.reshape(-1) > 0).astype(float)
    T2s = np.logspace(1, np.log10(2000.0), 60); T1s = 1000.0 * np.ones(60)
    ah = np.linspace(90.0, 180.0, 273); al = np.linspace(90.0, 180.0, 15)
    Dh = oracle.dictionary_fa_major(60, T2s, T1s, nt, 10.0, ah, 3000.0)
    Dl = oracle.dictionary_fa_major(60, T2s, T1s, nt, 10.0, al, 3000.0)
    idx, km, xm = oracle.fa_spline(Dl, al, Dh, ah, d2, m1, nthreads=4)
    fitted = (m1 > 0) & (d2.sum(axis=1) > 0)
    assert np.array_equal(np.where(fitted, ah[idx.astype(int)], 0.0).reshape(shp), g["FA"])
    lam_grid = np.zeros(50); lam_grid[1:] = np.logspace(-8, 1, 49)
    fs, sg, rg, st = oracle.fit_batch("L_curve", Dh, oracle.penalty(60, "L1"), d2, idx, m1, lambda_reg=lam_grid, nthreads=4)
    assert relmax(fs.reshape(shp + (60,)), g["fsol_4D"]) < 1e-8
    assert np.array_equal(rg.reshape(shp), g["reg_param"])
    maps = oracle.metrics(fs, T2s, m1)
    for name in ("MWF", "IEWF", "FWF", "T2_M", "T2_IE", "TWC"):
        assert np.allclose(maps[name].reshape(shp), g[name], rtol=1e-8, atol=1e-12), name


def test_oracle_nesma_vs_driver_golden(oracle):
    # the NESMA filter (motor:305-333).  The golden holds the rows the reference's driver handed
    # to its FA step after filtering, and its end-to-end maps (denoise='NESMA', X2/L2, brute force; mask has a 2)
    g = np.load(os.path.join(GOLDEN, "golden_nesma.npz"))
    data = g["data"]; mask = g["mask"]
    shp = mask.shape; nt = data.shape[-1]
    dm = data * mask[..., None]
    dm = np.where(dm < 0, 0.0, dm)
    den = oracle.nesma(dm, mask, nthreads=4)
    assert np.array_equal(den, g["denoised"], equal_nan=True)
    assert np.all(den[mask != 1] == 0) and np.count_nonzero(np.abs(den - dm).sum(axis=-1)) > 300
    d2 = den.reshape(-1, nt); m1 = (mask.reshape(-1) > 0).astype(float)
    T2s = np.logspace(1, np.log10(2000.0), 60); T1s = 1000.0 * np.ones(60); alphas = np.linspace(90.0, 180.0, 91)
    D = oracle.dictionary_fa_major(60, T2s, T1s, nt, 10.0, alphas, 3000.0)
    idx, km, sse, f = oracle.fa_bruteforce(D, d2, m1, nthreads=4)
    fitted = (m1 > 0) & (d2.sum(axis=1) > 0)
    assert np.array_equal(np.where(fitted, alphas[idx.astype(int)], 0.0).reshape(shp), g["FA"])
    fs, sg, rg, st = oracle.fit_batch("X2", D, oracle.penalty(60, "L2", T2s), d2, idx, m1, nthreads=4)
    assert relmax(fs.reshape(shp + (60,)), g["fsol_4D"]) < 1e-8
    assert np.allclose(rg.reshape(shp), g["reg_param"], rtol=1e-7, atol=1e-12)
    assert np.allclose(oracle.metrics(fs, T2s, m1)["MWF"].reshape(shp), g["MWF"], rtol=1e-8, atol=1e-12)


def test_oracle_nesma_edge_cases(oracle):
    # an all-zero signal inside the mask has no similar voxel (0/0 in the relative distance): mean of nothing = nan,
    # as numpy gives the reference; a 1-voxel volume is its own neighbour only for x-6 <= x < x+6
    rng = np.random.default_rng(5)
    d = rng.uniform(1.0, 2.0, size=(3, 2, 2, 11)); m = np.ones((3, 2, 2)); d[1, 1, 1] = 0.0
    out = oracle.nesma(d, m)
    assert np.all(np.isnan(out[1, 1, 1])) and np.isfinite(np.delete(out.reshape(-1, 11), 7, axis=0)).all()
    one = oracle.nesma(d[:1, :1, :1], m[:1, :1, :1])
    assert np.array_equal(one, d[:1, :1, :1])
    # numpy's own evaluation of the same expression (pairwise np.sum over 11 and over 130 echoes)
    for nt in (5, 11, 40, 130):
        d = rng.uniform(1.0, 1.05, size=(4, 3, 2, nt)); m = np.ones((4, 3, 2))
        ref = np.zeros_like(d); flat = d.reshape(-1, nt)
        for i in range(flat.shape[0]):
            RE = 100 * np.sum(np.abs(flat - flat[i]), axis=1) / np.sum(flat[i])
            ref.reshape(-1, nt)[i] = np.mean(flat[RE < 2.5, :], axis=0)
        assert np.array_equal(oracle.nesma(d, m), ref), nt


def test_oracle_gaussian_smooth_vs_scipy_and_driver(oracle):
    # SURVEY.md §8f item 1, second half: the Gaussian pre-smoothing of the FA step (motor:337-343).  The restatement is
    # pinned to scipy.ndimage.gaussian_filter itself (the reference's dependency, present here) and to the rows the
    # reference's driver handed to its spline FA step in a run with the CLI defaults (spline, FA_smooth='yes', X2)
    import scipy.ndimage as filt
    rng = np.random.default_rng(3)
    for shp in [(3, 1, 2, 4), (20, 17, 9, 5), (1, 1, 1, 3), (9, 30, 4, 2), (2, 2, 40, 1)]:
        d = rng.standard_normal(shp)
        ref = np.stack([filt.gaussian_filter(d[..., c], 2.0, 0) for c in range(shp[-1])], axis=-1)
        assert np.array_equal(oracle.gaussian_smooth(d, 2.0, nthreads=2), ref), shp
    g = np.load(os.path.join(GOLDEN, "golden_motor_default_smooth.npz"))
    dm = g["data"] * g["mask"][..., None]
    dm = np.where(dm < 0, 0.0, dm)
    sm = oracle.gaussian_smooth(dm, 2.0, nthreads=4)
    assert np.array_equal(sm, g["smoothed"])
    # the default pipeline end to end: spline FA on the smoothed rows, X2/L2 on the unsmoothed ones
    shp = g["mask"].shape; nt = dm.shape[-1]
    m1 = (g["mask"].reshape(-1) > 0).astype(float)
    T2s = np.logspace(1, np.log10(2000.0), 60); T1s = 1000.0 * np.ones(60)
    ah = np.linspace(90.0, 180.0, 273); al = np.linspace(90.0, 180.0, 15)
    Dh = oracle.dictionary_fa_major(60, T2s, T1s, nt, 10.0, ah, 3000.0)
    Dl = oracle.dictionary_fa_major(60, T2s, T1s, nt, 10.0, al, 3000.0)
    idx, km, xm = oracle.fa_spline(Dl, al, Dh, ah, sm.reshape(-1, nt), m1, nthreads=4)
    fitted = (m1 > 0) & (sm.reshape(-1, nt).sum(axis=1) > 0)
    assert np.array_equal(np.where(fitted, ah[idx.astype(int)], 0.0).reshape(shp), g["FA"])
    fs, sg, rg, st = oracle.fit_batch("X2", Dh, oracle.penalty(60, "L2", T2s), dm.reshape(-1, nt), idx, m1, nthreads=4)
    assert relmax(fs.reshape(shp + (60,)), g["fsol_4D"]) < 1e-8
    assert np.allclose(oracle.metrics(fs, T2s, m1)["MWF"].reshape(shp), g["MWF"], rtol=1e-8, atol=1e-12)


def test_spline_weights_match_scipy(oracle):
    # interp1d(kind='cubic') is the not-a-knot cubic spline: same interpolant from the slope form
    from scipy.interpolate import interp1d
    x = np.linspace(90.0, 180.0, 15)
    rng = np.random.default_rng(3)
    y = rng.uniform(0.5, 2.0, 15) + 0.002 * (x - 140.0) ** 2
    s = oracle.spline_weights(x) @ y
    f2 = interp1d(x, y, kind="cubic")
    for xx in np.linspace(90.0, 180.0, 257):
        i = min(np.searchsorted(x, xx, side="right") - 1, 13)
        h = x[i + 1] - x[i]; t = (xx - x[i]) / h
        v = (1 + 2 * t) * (1 - t) ** 2 * y[i] + t * (1 - t) ** 2 * h * s[i] + t * t * (3 - 2 * t) * y[i + 1] + t * t * (t - 1) * h * s[i + 1]
        assert abs(v - float(f2(xx))) < 1e-12


def test_metrics_edge_cases(oracle):
    # motor:448-468: masked-in but unfitted voxel -> fractions 0, T2_M = T2_IE = 1, TWC = 1e-16; masked-out -> 0
    T2s = np.logspace(1, np.log10(2000.0), 60)
    fs = np.zeros((3, 60)); fs[2, 5] = 2.0; fs[2, 30] = 6.0
    m = oracle.metrics(fs, T2s, np.array([1.0, 0.0, 1.0]))
    assert m["MWF"][0] == 0 and m["T2_M"][0] == 1.0 and m["T2_IE"][0] == 1.0 and m["TWC"][0] == 1e-16
    assert all(m[k][1] == 0 for k in m)
    assert abs(m["MWF"][2] - 0.25) < 1e-15 and abs(m["IEWF"][2] - 0.75) < 1e-15 and abs(m["T2_M"][2] - T2s[5]) < 1e-12


def test_shard_ranges():
    d = importlib.import_module(PKG + ".dist")
    for n in (0, 1, 7, 8, 1000, 1048576, 5120000):
        for w in (1, 2, 3, 8):
            r = [d.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_interleaved_shards_partition_the_voxel_list():
    # block b of 4 096 voxels -> rank b mod world; every voxel exactly once, counts known up front
    d = importlib.import_module(PKG + ".dist")
    for n in (0, 1, 4095, 4096, 4097, 100000, 1048576, 5120000):
        for w in (1, 2, 3, 8):
            parts = [d.shard_indices(n, k, w) for k in range(w)]
            assert [int(p.numel()) for p in parts] == [d.shard_count(n, k, w) for k in range(w)]
            allv = torch.cat(parts)
            assert allv.numel() == n and (n == 0 or bool((torch.sort(allv)[0] == torch.arange(n)).all()))
            for k, p in enumerate(parts):
                assert bool((((p // d.BLOCK) % w) == k).all())
            if n >= w * d.BLOCK:
                assert max(p.numel() for p in parts) - min(p.numel() for p in parts) <= d.BLOCK
    small = [d.shard_indices(37, k, 3, block=4) for k in range(3)]
    assert small[0].tolist() == [0, 1, 2, 3, 12, 13, 14, 15, 24, 25, 26, 27, 36]


def test_sharding_without_a_process_group_is_an_error(monkeypatch):
    d = importlib.import_module(PKG + ".dist")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(RuntimeError):
        d.fit_sharded(lambda idx: {"reg": torch.zeros(idx.numel())}, 10, gather=("reg",), device=torch.device("cpu"))


_GLOO_WORKER = r"""
import os, sys, importlib
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
d = importlib.import_module(%(pkg)r + ".dist")
rank, local, world = d.init(backend="gloo")
from oracle import oracle
nvox, nte, nt2 = 37, 32, 60
rng = np.random.default_rng(5)
T2s = np.logspace(1, np.log10(2000.0), nt2); T1s = 1000.0 * np.ones(nt2)
D = oracle.dictionary_fa_major(nt2, T2s, T1s, nte, 10.0, [150.0], 3000.0)
L = oracle.penalty(nt2, "L2")
x = np.zeros((nvox, nt2)); x[:, 12] = rng.uniform(0.1, 0.3, nvox); x[:, 28] = rng.uniform(0.5, 1.0, nvox)
data = (x @ D[0].T) * 1000.0 * (1 + 0.01 * rng.standard_normal((nvox, nte)))
data = torch.as_tensor(np.abs(data))
calls = {"n": 0}
orig_gather = dist.gather
def counting_gather(*a, **k):
    calls["n"] += 1
    return orig_gather(*a, **k)
dist.gather = counting_gather
def fit_fn(idx):      # stand-in compute for the CPU rehearsal of the N>1 path (the checker, not the product)
    blk = data[idx]
    fs, sg, rg, st = oracle.fit_batch("X2", D, L, blk.numpy(), np.zeros(blk.shape[0]), np.ones(blk.shape[0]))
    maps = oracle.metrics(fs, T2s, np.ones(blk.shape[0]))
    return {"maps": torch.as_tensor(np.stack([maps[k] for k in ("MWF", "IEWF", "FWF", "T2_M", "T2_IE", "TWC")])), "reg": torch.as_tensor(rg),
            "fsol": torch.as_tensor(fs)}
out, res = d.fit_sharded(fit_fn, nvox, gather=("maps", "reg", "fsol"), block=4, device=torch.device("cpu"))
assert calls["n"] == 1, "the path has ONE collective, saw %%d" %% calls["n"]
if rank == 0:
    full = fit_fn(torch.arange(nvox))
    assert res["maps"].shape == (6, nvox) and torch.equal(res["maps"], full["maps"]), "maps mismatch"
    assert torch.equal(res["reg"], full["reg"]) and torch.equal(res["fsol"], full["fsol"])
    print("GLOO_OK", world)
else:
    assert res is None
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fit_and_gather_gloo(tmp_path, world, oracle):
    # the N>1 path (interleaved voxel blocks per rank + the single packed gather) rehearsed on CPU over gloo
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT, "pkg": PKG})
    port = 29600 + world + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "GLOO_OK %d" % world in p.stdout


def test_ctypes_argtypes_match_the_header_prototypes():
    # every prototype of include/met2_hip.h against the argtypes the Python binding declares: same argument count, pointers
    # where the header has pointers, 64-bit integers where it has int64_t (a drift here is a silent stack/register mismatch)
    import ctypes as C
    lib = importlib.import_module(PKG + "._lib")
    L = lib.lib()
    header = open(os.path.join(ROOT, "include", "met2_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = re.findall(r"\b(?:int|int64_t|void|const char \*)\s*(met2_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S)
    assert len(protos) >= 28
    checked = 0
    for name, args in protos:
        fn = getattr(L, name)
        if fn.argtypes is None:
            continue
        params = [a.strip() for a in args.replace("\n", " ").split(",") if a.strip() and a.strip() != "void"]
        assert len(params) == len(fn.argtypes), (name, params, fn.argtypes)
        for p, t in zip(params, fn.argtypes):
            is_ptr = "*" in p
            if is_ptr:
                assert t is C.c_void_p or hasattr(t, "contents") or t is C.c_char_p, (name, p, t)
            elif p.startswith("int64_t"):
                assert t is C.c_int64, (name, p, t)
            elif p.startswith("int32_t"):
                assert t is C.c_int32, (name, p, t)
            elif p.startswith("double"):
                assert t is C.c_double, (name, p, t)
        checked += 1
    assert checked >= 20


def test_fit_host_argument_checks_need_no_gpu():
    """met2_fit_host / host.fit_host reject malformed calls before anything touches a device (no compute: runs on the CPU box)"""
    import ctypes as C
    host = importlib.import_module(PKG + ".host")
    lib = importlib.import_module(PKG + "._lib")

    class FakePlan:                      # shape only: the checks below fail before the handle is used
        n_te, n_t2, _h = 32, 60, C.c_void_p(0)

    with pytest.raises(ValueError, match="unknown reg_method"):
        host.fit_host(FakePlan(), "X3", np.zeros((4, 32)))
    with pytest.raises(ValueError, match="float64"):
        host.fit_host(FakePlan(), "X2", np.zeros((4, 32), dtype=np.float32))
    with pytest.raises(ValueError, match="n_te=32"):
        host.fit_host(FakePlan(), "X2", np.zeros((4, 31)))
    with pytest.raises(ValueError, match="one entry per voxel"):
        host.fit_host(FakePlan(), "X2", np.zeros((4, 32)), mask=np.ones(5))
    with pytest.raises(ValueError, match="no plan"):
        host.fit_host([], "X2", np.zeros((4, 32)))
    L = lib.lib()
    assert L.met2_fit_host(None, 0, 2, 0, None, None, 32, 1, None, None, None, 0, None, None, None, None, None, None, None, None, 0, None) == -1
    assert b"1 to 64 plans" in L.met2_last_error()
    arr = (C.c_void_p * 1)(None)
    assert L.met2_fit_host(arr, 1, 2, 0, None, None, 32, 1, None, None, None, 0, None, None, None, None, None, None, None, None, 0, None) == -1
    assert b"NULL plan" in L.met2_last_error()
