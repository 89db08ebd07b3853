"""One pass of the CLI-default pipeline (spline FA on Gaussian-smoothed echoes, X2/L2) on a configs[1]-sized volume with
device-resident inputs; run under `rocprofv3 --kernel-trace --stats` to see every kernel of the pipeline."""
import importlib, sys
import numpy as np, torch
sys.path.insert(0, ".")
pkg = importlib.import_module("multicomponent-t2-toolbox_amd")
motor = importlib.import_module("multicomponent-t2-toolbox_amd.motor")
synth = importlib.import_module("multicomponent-t2-toolbox_amd.synth")
dims = (128, 128, 64); nvox = int(np.prod(dims)); nt = 32
T2s = synth.t2_grid(60); T1s = 1000.0 * np.ones(60)
ah = np.linspace(90.0, 180.0, 273); al = np.linspace(90.0, 180.0, 15)
ph = pkg.Met2Plan(nt, 60, 273); ph.build_dictionary_epg(T2s, T1s, 10.0, ah, 3000.0).set_penalty("L2", T2s)
pl = pkg.Met2Plan(nt, 60, 15); pl.build_dictionary_epg(T2s, T1s, 10.0, al, 3000.0)
data, _, _ = synth.make_voxels(nvox, nte=nt, seed=9, fa_values=np.linspace(90.0, 180.0, 91), device="cuda")
mask = torch.ones(nvox, dtype=torch.uint8, device="cuda")
for rep in range(2):
    sm = motor.gaussian_smooth(data.reshape(dims + (nt,)), 2.0).reshape(-1, nt)
    fa, km, _ = ph.fa_spline(pl, al, ah, sm, mask)
    out = ph.fit("X2", data, fa_index=fa, mask=mask)
torch.cuda.synchronize()
print("default pipeline done, MWF mean", float(out["maps"][0].mean()))
