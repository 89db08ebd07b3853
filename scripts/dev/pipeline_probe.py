"""One timing table for the reference's example pipeline (example_script_run_MET2_preproc_and_recon.sh: --denoise TV, FA_method spline, FA_smooth
yes, X2 / L2) on a 128 x 128 x 64 x 32 head phantom: every device step on device-resident tensors (HIP events, best of three), and the wall
clock of whole recon_met2_arrays calls from a pageable numpy volume with each denoising option.
    python3 scripts/dev/pipeline_probe.py [nx,ny,nz]"""
import importlib, json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
PKG = "multicomponent-t2-toolbox_amd"
motor = importlib.import_module(PKG + ".motor")
synth = importlib.import_module(PKG + ".synth")
tv = importlib.import_module(PKG + ".tv")
pkg = importlib.import_module(PKG)
dims = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "128,128,64").split(","))
nte, nt2 = 32, 60
nvox = int(np.prod(dims))
vol, mask = synth.make_phantom(dims, nte=nte, device="cuda:0")


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best, out


rows = []
ms, den = timed(lambda: tv.tv_chambolle(vol))
_, sg, it = tv.tv_chambolle(vol, return_info=True)
rows.append(("TV denoising, all 32 echoes (motor:293-304)", ms, {"iterations_min_max": [int(it.min()), int(it.max())]}))
ms, _ = timed(lambda: motor.nesma_filter(vol, mask))
rows.append(("NESMA filter (motor:305-333)", ms, {}))
ms, sm = timed(lambda: motor.gaussian_smooth(vol, 2.0))
rows.append(("Gaussian pre-smoothing of the FA step, sigma 2 (motor:337-343)", ms, {}))
T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2)
flat = vol.reshape(nvox, nte); mflat = mask.reshape(nvox)
a91 = np.linspace(90.0, 180.0, 91)
p91 = pkg.Met2Plan(nte, nt2, 91); p91.build_dictionary_epg(T2s, T1s, 10.0, a91, 3000.0).set_penalty("L2", T2s)
ms, fa = timed(lambda: p91.fa_bruteforce(flat, mflat)[0])
rows.append(("brute-force FA over 91 flip angles (fa_estimation.py:74-111)", ms, {}))
ms, _ = timed(lambda: p91.fit("X2", flat, fa_index=fa, mask=mflat))
rows.append(("fit X2/L2 + metrics, per-voxel FA from the 91-grid (motor:427-472)", ms, {}))
for r in rows:
    print(json.dumps({"step": r[0], "dims": list(dims) + [nte], "ms": r[1], "voxels_per_s": nvox / (r[1] * 1e-3), **r[2]}))
p91.close()
host = vol.cpu().numpy(); hmask = mask.cpu().numpy().astype(np.int64)
TE = 10.0 * np.arange(1, nte + 1)
for denoise, fa_method, smooth in (("None", "spline", "no"), ("TV", "spline", "no"), ("TV", "spline", "yes"), ("NESMA", "spline", "yes"), ("TV", "brute-force", "no")):
    best = None
    for rep in range(3):
        res = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = motor.recon_met2_arrays(host, hmask, TE, 3000.0, "X2", "L2", fa_method, 40.0, denoise=denoise, FA_smooth=smooth)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(json.dumps({"driver": "recon_met2_arrays (pageable numpy volume in, ten numpy outputs back; plan construction included)", "dims": list(dims) + [nte],
                      "denoise": denoise, "FA_method": fa_method, "FA_smooth": smooth, "seconds": best, "voxels_per_s": nvox / best,
                      "MWF_mean_in_mask": float(res["MWF"][hmask > 0].mean())}))
