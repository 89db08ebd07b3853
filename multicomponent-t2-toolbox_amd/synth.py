"""Seeded synthetic MET2 volumes, following the reference's own Monte-Carlo recipe
(scripts_synthetic_data_evaluation/Paper_Comparison/evaluate_all_methods_two_lobes_SNR50_150.py:156-170,
:385-394): two Gaussian lobes on a 1000-point T2 grid, EPG signal, Rician noise.  Data generation is
plumbing around the hot path: numpy for the (tiny) high-resolution EPG table, torch for the bulk."""
import math

import numpy as np
import torch


def t2_grid(npc, t2_min=10.0, t2_max=2000.0):
    # motor/motor_recon_met2_real_data.py:215-220
    return np.logspace(math.log10(t2_min), math.log10(t2_max), num=npc, endpoint=True, base=10.0)


def lambda_grid(num=50, lo=1e-8, hi=10.0):
    # motor:248-251
    lam = np.zeros(num)
    lam[1:] = np.logspace(math.log10(lo), math.log10(hi), num=num - 1, endpoint=True, base=10.0)
    return lam


def epg_table(nte, tau, T2, T1, alpha_deg):
    """Echo amplitudes [nte, len(T2)] of the CPMG train of epg/epg.py:64-153 (vectorised over T2).
    Includes the (1 - exp(-TR/T1)) factor's companion only when the caller multiplies it in."""
    T2 = np.asarray(T2, dtype=np.float64); T1 = np.asarray(T1, dtype=np.float64)
    a = alpha_deg * np.pi / 180.0
    aexc = alpha_deg / 2.0 * np.pi / 180.0
    E2 = np.exp(-(tau / 2.0) / T2); E1 = np.exp(-(tau / 2.0) / T1)
    c2, s2, sa, ca = math.cos(a / 2) ** 2, math.sin(a / 2) ** 2, math.sin(a), math.cos(a)
    nr = T2.shape[0]
    F0 = np.full(nr, math.sin(aexc))
    Fp = np.zeros((nte + 2, nr)); Fm = np.zeros((nte + 2, nr)); Z = np.zeros((nte + 2, nr))
    Fm[1] = math.cos(aexc)
    out = np.zeros((nte, nr))
    for e in range(nte):
        for half in range(2):
            nF0 = Fm[1].copy()
            Fp[2:nte + 1] = Fp[1:nte].copy(); Fp[1] = F0
            Fm[1:nte] = Fm[2:nte + 1].copy(); Fm[nte] = 0.0
            F0 = nF0 * E2
            Fp[1:nte + 1] *= E2; Fm[1:nte + 1] *= E2; Z[1:nte + 1] *= E1
            if half == 0:
                A, B, Zz = Fp[1:nte + 1].copy(), Fm[1:nte + 1].copy(), Z[1:nte + 1].copy()
                Fp[1:nte + 1] = c2 * A + s2 * B + sa * Zz
                Fm[1:nte + 1] = s2 * A + c2 * B - sa * Zz
                Z[1:nte + 1] = -0.5 * sa * A + 0.5 * sa * B + ca * Zz
        out[e] = F0
    return out


def make_voxels(nvox, nte=32, seed=20260102, fa_deg=150.0, fa_values=None, snr=(50.0, 150.0), te=10.0, TR=3000.0,
                device="cuda", chunk=65536, params=None):
    """Returns (data [nvox,nte] float64 tensor on `device`, fa_index float64 tensor or None, truth dict).
    fa_deg: constant flip angle; or fa_values (array of the dictionary's FA grid) -> per-voxel FA drawn
    uniformly from that grid, fa_index = its index."""
    rng = np.random.default_rng(seed)
    dev = torch.device(device)
    T2g = np.linspace(1.0, 300.0, 1000)
    T1g = 1000.0 * np.ones_like(T2g)
    par = {
        "MWF": rng.uniform(0.05, 0.25, nvox), "T2m": rng.uniform(15.0, 35.0, nvox), "T2ie": rng.uniform(60.0, 90.0, nvox),
        "SNR": rng.uniform(snr[0], snr[1], nvox), "sm": rng.uniform(1.0, 3.0, nvox), "sie": rng.uniform(6.0, 12.0, nvox),
    }
    if params is not None:                                           # spatially organised parameters (make_phantom)
        par.update({k: np.asarray(v, dtype=np.float64).reshape(nvox) for k, v in params.items()})
    if fa_values is None:
        fa_idx = None
        fas = [float(fa_deg)]
        which = np.zeros(nvox, dtype=np.int64)
    else:
        fa_values = np.asarray(fa_values, dtype=np.float64)
        which = rng.integers(0, fa_values.shape[0], nvox)
        fa_idx = which.astype(np.float64)
        fas = list(fa_values)
    gen = torch.Generator(device=dev); gen.manual_seed(int(seed))
    data = torch.empty((nvox, nte), dtype=torch.float64, device=dev)
    grid = torch.as_tensor(T2g, device=dev)
    tabs = {}
    for s in range(0, nvox, chunk):
        e = min(nvox, s + chunk)
        p = {k: torch.as_tensor(v[s:e], device=dev).unsqueeze(1) for k, v in par.items()}
        pdf = lambda mu, sg: torch.exp(-0.5 * ((grid - mu) / sg) ** 2) / (sg * math.sqrt(2.0 * math.pi))
        dist = p["MWF"] * pdf(p["T2m"], p["sm"]) + (1.0 - p["MWF"]) * pdf(p["T2ie"], p["sie"])
        dist = dist / dist.sum(dim=1, keepdim=True)
        S = torch.empty((e - s, nte), dtype=torch.float64, device=dev)
        w = torch.as_tensor(which[s:e], device=dev)
        for fi in np.unique(which[s:e]):
            if fi not in tabs:
                tab = (1.0 - np.exp(-TR / T1g)) * epg_table(nte, te, T2g, T1g, fas[int(fi)])
                tabs[fi] = torch.as_tensor(tab.T.copy(), device=dev)        # [1000, nte]
            sel = (w == int(fi))
            S[sel] = 1000.0 * (dist[sel] @ tabs[fi])
        sg = S[:, :1] / p["SNR"]
        n1 = torch.randn(S.shape, dtype=torch.float64, device=dev, generator=gen) * sg
        n2 = torch.randn(S.shape, dtype=torch.float64, device=dev, generator=gen) * sg
        data[s:e] = torch.sqrt((S + n1) ** 2 + n2 ** 2)
    fa_t = torch.as_tensor(fa_idx, device=dev) if fa_idx is not None else None
    return data, fa_t, par


def make_phantom(shape, nte=32, seed=20260110, snr=80.0, device="cuda"):
    """A spatially organised volume [nx, ny, nz, nte] for the filters that look at neighbourhoods (TV, NESMA, Gaussian smoothing):
    an ellipsoidal head of 'grey matter' with a 'white matter' core (higher myelin fraction) and smooth gradients of the
    compartment T2s, zero outside, Rician noise at the given SNR of the first echo.  Returns (data tensor, mask uint8 tensor)."""
    nx, ny, nz = shape
    x, y, z = np.meshgrid(np.linspace(-1, 1, nx), np.linspace(-1, 1, ny), np.linspace(-1, 1, nz), indexing="ij")
    r2 = (x / 0.9) ** 2 + (y / 0.8) ** 2 + (z / 0.85) ** 2
    head = r2 < 1.0
    wm = ((x / 0.55) ** 2 + (y / 0.5) ** 2 + (z / 0.5) ** 2) < 1.0
    n = nx * ny * nz
    params = {"MWF": np.where(wm, 0.16 + 0.04 * x, 0.07 + 0.02 * y), "T2m": 22.0 + 4.0 * z, "T2ie": np.where(wm, 70.0, 82.0) + 4.0 * x * y,
              "SNR": np.full(n, float(snr)), "sm": np.full(n, 2.0), "sie": np.full(n, 9.0)}
    data, _, _ = make_voxels(n, nte=nte, seed=seed, device=device, params=params)
    mask = torch.as_tensor(head.reshape(-1), device=data.device)
    data = data * mask.unsqueeze(1).to(data.dtype)
    return data.reshape(nx, ny, nz, nte), mask.reshape(nx, ny, nz).to(torch.uint8)
