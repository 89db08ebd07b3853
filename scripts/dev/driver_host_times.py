"""Host-side durations of the calls recon_met2_arrays makes per chunk (no synchronisation added): where does the host block?"""
import importlib, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
PKG = "multicomponent-t2-toolbox_amd"
motor = importlib.import_module(PKG + ".motor"); synth = importlib.import_module(PKG + ".synth"); planm = importlib.import_module(PKG + ".plan")
vol, mask = synth.make_phantom((128, 128, 64), nte=32, device="cuda:0")
host = vol.cpu().numpy(); hmask = mask.cpu().numpy().astype(np.int64)
TE = 10.0 * np.arange(1, 33)
log = []
def wrap(cls, name):
    f = getattr(cls, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); log.append((name, (time.perf_counter() - t0) * 1e3, t0)); return r
    setattr(cls, name, g)
for n in ("fa_spline", "fa_bruteforce", "fit", "finish", "build_dictionary_epg", "set_penalty", "close"):
    wrap(planm.Met2Plan, n)
for rep in range(3):
    log.clear(); res = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = motor.recon_met2_arrays(host, hmask, TE, 3000.0, "X2", "L2", "spline", 40.0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("wall %.1f ms" % ((t1 - t0) * 1e3))
    for name, ms, ts in log:
        print("   +%6.1f  %-22s %6.2f ms" % ((ts - t0) * 1e3, name, ms))
