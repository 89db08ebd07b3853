"""Development probe: step through the HIP path on a handful of voxels with progress prints."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MET2_DEBUG", "1")
import numpy as np, torch, faulthandler
faulthandler.dump_traceback_later(150, repeat=False)
import met2_amd
from met2_amd import synth
from oracle import oracle

def log(*a):
    print(*a, flush=True)

meths = sys.argv[1].split(",") if len(sys.argv) > 1 else ["NNLS"]
nvox = int(sys.argv[2]) if len(sys.argv) > 2 else 12
nte, nt2 = 32, 60
T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2); alphas = np.array([150.0])
plan = met2_amd.Met2Plan(nte, nt2, 1)
pen = os.environ.get("PEN", "L2")
plan.build_dictionary_epg(T2s, T1s, 10.0, alphas, 3000.0).set_penalty(pen, T2s)
log("plan ok", plan.launch_info())
data, _, _ = synth.make_voxels(nvox, nte=nte, seed=3, device="cpu")
log("data made on cpu")
data = data.cuda(); torch.cuda.synchronize(); log("data on gpu")
D = oracle.dictionary_fa_major(nt2, T2s, T1s, nte, 10.0, alphas, 3000.0)
L = oracle.penalty(nt2, pen, T2s)
for m in meths:
    t = time.time()
    out = plan.fit(m, data)
    torch.cuda.synchronize()
    log(m, "fit returned in %.3fs kernel %.3f ms" % (time.time() - t, plan.last_kernel_ms()))
    fs, sg, rg, st = oracle.fit_batch(m, D, L, data.cpu().numpy(), np.zeros(nvox), np.ones(nvox), lambda_reg=synth.lambda_grid(), nthreads=8)
    got = out["fsol"].cpu().numpy()
    e = np.max(np.abs(got - fs), axis=1) / np.max(np.abs(fs), axis=1)
    log(m, "rel err max %.3e median %.3e ; reg diff %.3e ; status" % (e.max(), np.median(e), np.max(np.abs(out["reg"].cpu().numpy() - rg))), np.unique(out["status"].cpu().numpy()))
    log(m, "frac > 1e-5: %.4f  > 1e-3: %.4f ; lam rel diff median %.2e" % ((e > 1e-5).mean(), (e > 1e-3).mean(), np.median(np.abs(out["reg"].cpu().numpy() - rg) / np.maximum(np.abs(rg), 1e-30))))
    mw = lambda x: x[:, T2s <= 40.0].sum(axis=1) / (x.sum(axis=1) + 1e-16)
    dm = np.abs(mw(got) - mw(fs)); log(m, "dMWF median %.2e p99 %.2e max %.2e" % (np.median(dm), np.quantile(dm, 0.99), dm.max()))
    if e.max() > 1e-5:
        bad = int(np.argmax(e))
        log("worst voxel", bad, "\n got", got[bad][:12], "\n ref", fs[bad][:12])
