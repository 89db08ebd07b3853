import importlib, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
PKG = "multicomponent-t2-toolbox_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth")
T2s = np.logspace(1, np.log10(2000.0), 60); T1s = 1000.0 * np.ones(60)
for nfa in (273, 91, 15, 273, 91, 15):
    al = np.linspace(90, 180, nfa)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = pkg.Met2Plan(32, 60, nfa); t1 = time.perf_counter()
    p.build_dictionary_epg(T2s, T1s, 10.0, al, 3000.0); t2 = time.perf_counter()
    p.set_penalty("L2", T2s); t3 = time.perf_counter()
    p.close(); t4 = time.perf_counter()
    print(nfa, "create %.2f  dict %.2f  penalty(+seeds,tables) %.2f  close %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
