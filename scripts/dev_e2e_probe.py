import importlib, sys, time, json
import numpy as np, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("multicomponent-t2-toolbox_amd"); motor = importlib.import_module("multicomponent-t2-toolbox_amd.motor"); synth = importlib.import_module("multicomponent-t2-toolbox_amd.synth")
nte, nt2 = 32, 60
T2s = synth.t2_grid(nt2)
plan = pkg.Met2Plan(nte, nt2, 1); plan.build_dictionary_epg(T2s, 1000.0 * np.ones(nt2), 10.0, np.array([150.0]), 3000.0).set_penalty("L2", T2s)
data, _, _ = synth.make_voxels(1 << 20, nte=nte, seed=20260102, device="cuda")
host = torch.empty(data.shape, dtype=torch.float64, pin_memory=True); host.copy_(data); torch.cuda.synchronize()
hostp = host.numpy().copy()
free = len(sys.argv) > 1
if free:  # (second invocation: the device copy of the volume released first)
    del data; torch.cuda.empty_cache()
for chunk in (65536, 131072, 262144):
    for src, name in ((host, "pinned"), (hostp, "pageable")):
        out = None; ts = []
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = motor.fit_host_pipeline(plan, "X2", src, chunk=chunk, out=out)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(json.dumps({"chunk": chunk, "input": name, "device_copy_freed": free, "ms": [round(1e3 * t, 1) for t in ts]}), flush=True)
