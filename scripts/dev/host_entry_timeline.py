"""met2_fit_host on configs[1]'s volume (pinned arrays, one plan, default blocks) three times, for a rocprofv3 --kernel-trace
--memory-copy-trace timeline (scripts/dev/timeline_report.py digests the last call)."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
PKG = "multicomponent-t2-toolbox_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); host = importlib.import_module(PKG + ".host")
nte, nt2, nvox = 32, 60, 128 * 128 * 64
T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2)
p = pkg.Met2Plan(nte, nt2, 1); p.build_dictionary_epg(T2s, T1s, 10.0, np.array([150.0]), 3000.0).set_penalty("L2", T2s)
data, _, _ = synth.make_voxels(nvox, nte=nte, seed=20260102, device="cuda")
pin = torch.empty(data.shape, dtype=torch.float64, pin_memory=True).copy_(data); torch.cuda.synchronize()
P = lambda shape, dt=torch.float64: torch.empty(shape, dtype=dt, pin_memory=True).numpy()
out = {"fsol": P((nvox, nt2)), "sig": P((nvox, nte)), "reg": P((nvox,)), "maps": P((6, nvox)), "status": P((nvox,), torch.int32), "fa_index": P((nvox,))}
for rep in range(3):
    t0 = time.perf_counter(); out = host.fit_host(p, "X2", pin.numpy(), out=out); print("wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
