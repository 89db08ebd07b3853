"""cProfile of one warm recon_met2_arrays call (settings from argv): which host call holds the time?"""
import importlib, sys, time, os, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
PKG = "multicomponent-t2-toolbox_amd"
motor = importlib.import_module(PKG + ".motor"); synth = importlib.import_module(PKG + ".synth")
denoise, fa_method, smooth = (sys.argv[1:4] + ["None", "spline", "no"][len(sys.argv) - 1:])[:3]
vol, mask = synth.make_phantom((128, 128, 64), nte=32, device="cuda:0")
host = vol.cpu().numpy(); hmask = mask.cpu().numpy().astype(np.int64)
TE = 10.0 * np.arange(1, 33)
run = lambda: motor.recon_met2_arrays(host, hmask, TE, 3000.0, "X2", "L2", fa_method, 40.0, denoise=denoise, FA_smooth=smooth)
for _ in range(2):
    res = run(); res = None
pr = cProfile.Profile()
torch.cuda.synchronize(); t0 = time.perf_counter()
pr.enable(); res = run(); pr.disable()
torch.cuda.synchronize(); print("wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
