"""Is the host thread being throttled?  cgroup quota, throttle counters before / after a driver run with small chunks, torch's thread count."""
import importlib, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
def rd(p):
    try: return open(p).read().strip().replace("\n", " | ")
    except Exception as e: return "n/a (%s)" % type(e).__name__
print("cpu.max:", rd("/sys/fs/cgroup/cpu.max"), "| nproc", os.cpu_count(), "| affinity", len(os.sched_getaffinity(0)), "| torch threads", torch.get_num_threads(), "| OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"))
PKG = "multicomponent-t2-toolbox_amd"
motor = importlib.import_module(PKG + ".motor"); synth = importlib.import_module(PKG + ".synth")
vol, mask = synth.make_phantom((128, 128, 64), nte=32, device="cuda:0")
host = vol.cpu().numpy(); hmask = mask.cpu().numpy().astype(np.int64)
TE = 10.0 * np.arange(1, 33)
res = None
for chunk in (65536, 131072, 262144):
    motor.PIPELINE_CHUNK = chunk
    for rep in range(3):
        res = None
        s0 = rd("/sys/fs/cgroup/cpu.stat")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = motor.recon_met2_arrays(host, hmask, TE, 3000.0, "X2", "L2", "spline", 40.0)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        s1 = rd("/sys/fs/cgroup/cpu.stat")
        pick = lambda s: {kv.split()[0]: int(kv.split()[1]) for kv in s.split(" | ") if kv.split()[0] in ("nr_throttled", "throttled_usec", "usage_usec")} if "n/a" not in s else {}
        a, b = pick(s0), pick(s1)
        print("chunk %d wall %.1f ms; cpu.stat delta %s" % (chunk, dt * 1e3, {k: b[k] - a[k] for k in a}))
        res = None
