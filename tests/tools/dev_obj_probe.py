import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, met2_amd
from met2_amd import synth
from oracle import oracle
nte, nt2 = 32, 60
T2s = synth.t2_grid(nt2); T1s = 1000.0 * np.ones(nt2); alphas = np.array([150.0])
pen = os.environ.get("PEN", "I")
plan = met2_amd.Met2Plan(nte, nt2, 1)
plan.build_dictionary_epg(T2s, T1s, 10.0, alphas, 3000.0).set_penalty(pen, T2s)
data, _, _ = synth.make_voxels(8, nte=nte, seed=3, device="cpu")
D = oracle.dictionary_fa_major(nt2, T2s, T1s, nte, 10.0, alphas, 3000.0)[0]
L = oracle.penalty(nt2, pen, T2s)
lams = np.array([1e-8, 1e-6, 1e-5, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 0.3, 0.5, 1.0, 1.9, 3.0, 5.0, 9.0])
np.set_printoptions(linewidth=200, precision=6)
for meth in sys.argv[1].split(","):
    got = plan.objective_grid(meth, data.cuda(), lams).cpu().numpy()
    for v in range(4):
        M = data[v].numpy() / data[v, 0].item()
        ref = oracle.objective(meth, D, M, L, lams)
        print(meth, pen, "vox", v, "\n  gpu", got[v], "\n  ref", ref, "\n  dif", got[v] - ref, flush=True)
