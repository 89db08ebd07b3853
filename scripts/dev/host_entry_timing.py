"""met2_fit_host on configs[1]'s volume (128x128x64, 32x60, X2/L2): 1, 2 and 4 plans sharing the one device of the test box, pinned and
pageable arrays, default and explicit block sizes.  Prints one JSON line per case (-> profiles/r04_host_entry.jsonl)."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
PKG = "multicomponent-t2-toolbox_amd"
pkg = importlib.import_module(PKG)
synth = importlib.import_module(PKG + ".synth")
host = importlib.import_module(PKG + ".host")

nte, nt2, nvox = 32, 60, 128 * 128 * 64
T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2); alphas = np.array([150.0])


def make_plans(k):
    ps = []
    for _ in range(k):
        p = pkg.Met2Plan(nte, nt2, 1)
        p.build_dictionary_epg(T2s, T1s, 10.0, alphas, 3000.0).set_penalty("L2", T2s)
        ps.append(p)
    return ps


data, _, _ = synth.make_voxels(nvox, nte=nte, seed=20260102, device="cuda")
pin = torch.empty(data.shape, dtype=torch.float64, pin_memory=True).copy_(data)
torch.cuda.synchronize()
page = np.array(pin.numpy())
pinned_out = {"fsol": torch.empty((nvox, nt2), dtype=torch.float64, pin_memory=True).numpy(),
              "sig": torch.empty((nvox, nte), dtype=torch.float64, pin_memory=True).numpy(),
              "reg": torch.empty((nvox,), dtype=torch.float64, pin_memory=True).numpy(),
              "maps": torch.empty((6, nvox), dtype=torch.float64, pin_memory=True).numpy(),
              "status": torch.empty((nvox,), dtype=torch.int32, pin_memory=True).numpy(),
              "fa_index": torch.empty((nvox,), dtype=torch.float64, pin_memory=True).numpy()}
ref = None
for k in (1, 2, 4):
    plans = make_plans(k)
    for label, src, outs in (("pinned", pin.numpy(), pinned_out), ("pageable", page, None)):
        for chunk, nosplit in ((0, False), (0, True), (65536, False), (131072, False), (262144, True)):
            os.environ.pop("MET2_HOST_NOSPLIT", None)
            if nosplit:
                os.environ["MET2_HOST_NOSPLIT"] = "1"
            best = None
            for i in range(4):
                t0 = time.perf_counter()
                outs = host.fit_host(plans, "X2", src, chunk=chunk, out=outs)
                dt = time.perf_counter() - t0
                if i:
                    best = dt if best is None else min(best, dt)
            if ref is None:
                ref = outs["fsol"].copy()
            print(json.dumps({"plans_on_one_device": k, "arrays": label, "chunk": chunk, "whole_blocks_only": nosplit, "ms": round(1e3 * best, 2), "voxels_per_s": round(nvox / best),
                              "plan_ms": [round(x, 1) for x in outs["plan_ms"]], "bit_equal": bool(np.array_equal(ref, outs["fsol"]))}), flush=True)
    for p in plans:
        p.close()
