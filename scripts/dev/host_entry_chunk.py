import importlib, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
PKG = "multicomponent-t2-toolbox_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); host = importlib.import_module(PKG + ".host")
nte, nt2, nvox = 32, 60, 128 * 128 * 64
T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2); alphas = np.array([150.0])
p = pkg.Met2Plan(nte, nt2, 1); p.build_dictionary_epg(T2s, T1s, 10.0, alphas, 3000.0).set_penalty("L2", T2s)
data, _, _ = synth.make_voxels(nvox, nte=nte, seed=20260102, device="cuda")
pin = torch.empty(data.shape, dtype=torch.float64, pin_memory=True).copy_(data); torch.cuda.synchronize()
page = np.array(pin.numpy())
P = lambda shape, dt=torch.float64: torch.empty(shape, dtype=dt, pin_memory=True).numpy()
pinned_out = {"fsol": P((nvox, nt2)), "sig": P((nvox, nte)), "reg": P((nvox,)), "maps": P((6, nvox)), "status": P((nvox,), torch.int32), "fa_index": P((nvox,))}
for label, src, outs in (("pinned", pin.numpy(), pinned_out), ("pageable", page, None)):
    for chunk in (262144, 349526, 524288, 1048576):
        best = None
        for i in range(4):
            t0 = time.perf_counter(); outs = host.fit_host(p, "X2", src, chunk=chunk, out=outs); dt = time.perf_counter() - t0
            if i: best = dt if best is None else min(best, dt)
        print(json.dumps({"arrays": label, "chunk": chunk, "ms": round(1e3 * best, 2)}), flush=True)
