"""Time the NESMA filter kernel on a full-size volume (configs[1] geometry: 128 x 128 x 64 voxels, 32 echoes)."""
import importlib
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("multicomponent-t2-toolbox_amd")
motor = importlib.import_module("multicomponent-t2-toolbox_amd.motor")

dims = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "128,128,64").split(","))
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = torch.Generator(device="cuda").manual_seed(3)
lab = torch.randint(0, 3, dims, device="cuda", generator=g)
base = torch.rand((3, nt), device="cuda", dtype=torch.float64, generator=g) + 0.5
d = base[lab] * (1.0 + 0.01 * torch.randn(dims + (nt,), device="cuda", dtype=torch.float64, generator=g))
m = torch.ones(dims, device="cuda", dtype=torch.int64)
out = motor.nesma_filter(d, m)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    out = motor.nesma_filter(d, m)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
nvox = int(np.prod(dims))
print(json.dumps({"kernel": "nesma", "dims": dims, "nt": nt, "ms": ms, "voxels_per_s": nvox / ms * 1e3,
                  "window_reads_GB": nvox * 1728 * nt * 8 / 1e9, "finite": bool(torch.isfinite(out).all())}))
