"""recon_met2_arrays wall clock against the driver's chunk size (plain run and the example pipeline), 128x128x64x32 phantom and an unmasked volume."""
import importlib, sys, time, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
PKG = "multicomponent-t2-toolbox_amd"
motor = importlib.import_module(PKG + ".motor"); synth = importlib.import_module(PKG + ".synth")
vol, mask = synth.make_phantom((128, 128, 64), nte=32, device="cuda:0")
host = vol.cpu().numpy(); hmask = mask.cpu().numpy().astype(np.int64)
full = np.ones_like(hmask)
TE = 10.0 * np.arange(1, 33)
for chunk in (65536, 131072, 262144, 524288):
    motor.PIPELINE_CHUNK = chunk
    row = {"chunk": chunk}
    for name, m, kw in (("plain", hmask, {}), ("example", hmask, {"denoise": "TV", "FA_smooth": "yes"}), ("plain_unmasked", full, {})):
        best = None
        for rep in range(4):
            res = None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = motor.recon_met2_arrays(host, m, TE, 3000.0, "X2", "L2", "spline", 40.0, **kw)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        row[name] = round(best * 1e3, 1)
    print(json.dumps(row))
