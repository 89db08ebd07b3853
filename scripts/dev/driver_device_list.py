"""recon_met2_arrays(devices=[0]) -- the one-process driver through met2_fit_host -- next to the default (torch-pipelined) driver on the
128 x 128 x 64 x 32 head phantom: wall clock from a pageable numpy volume to the ten numpy outputs, plan construction included."""
import importlib, json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
PKG = "multicomponent-t2-toolbox_amd"
motor = importlib.import_module(PKG + ".motor")
synth = importlib.import_module(PKG + ".synth")
dims = (128, 128, 64); nte = 32
nvox = int(np.prod(dims))
vol, mask = synth.make_phantom(dims, nte=nte, device="cuda:0")
host = vol.cpu().numpy(); hmask = mask.cpu().numpy().astype(np.int64)
del vol
TE = 10.0 * np.arange(1, nte + 1)
for denoise, fa_method, smooth in (("None", "spline", "no"), ("TV", "spline", "yes"), ("None", "brute-force", "no")):
    for devs in (None, [0], [0, 0]):
        best = None
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = motor.recon_met2_arrays(host, hmask, TE, 3000.0, "X2", "L2", fa_method, 40.0, denoise=denoise, FA_smooth=smooth, devices=devs)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if rep:
                best = dt if best is None else min(best, dt)
        print(json.dumps({"driver": "recon_met2_arrays", "devices": devs, "denoise": denoise, "FA_method": fa_method, "FA_smooth": smooth, "seconds": round(best, 4),
                          "voxels_per_s": round(nvox / best), "MWF_mean_in_mask": float(res["MWF"][hmask > 0].mean())}), flush=True)
