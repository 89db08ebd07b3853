#!/usr/bin/env python3
"""A/B of build-time switches with hardware counters on configs[1] (run ON THE GPU BOX):
    python3 scripts/dev/ab_pmc.py "-DMET2_REORDER=0" "-DMET2_REORDER=1"
For every set of defines: rebuild the X2-only library, one plain bench run (kernel ms) and one rocprofv3 --pmc pass (SQ instruction
and wait counters of the dominant fit_kernel dispatch).  Prints one line per build."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CTRS = os.environ["CTRS"].split() if os.environ.get("CTRS") else ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_LDS_IDX_ACTIVE"]
bench = ["python3", os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end"]
for defs in sys.argv[1:]:
    env = dict(os.environ, MET2_BUILD_DEFINES="-DMET2_ONLY=2 " + defs, TMPDIR="/tmp")
    subprocess.check_call(["python3", "-c", "import importlib; importlib.import_module('multicomponent-t2-toolbox_amd._build').build(force=True)"], cwd=ROOT, env=env)
    p = subprocess.run(bench[:2] + ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end"], capture_output=True, text=True, env=env, cwd=ROOT)
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    d = "/tmp/abpmc_%d" % abs(hash(defs))
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + CTRS + ["-d", d, "-o", "p", "--output-format", "csv", "--"] + bench, cwd="/tmp", env=env,
                   capture_output=True, text=True)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    tot = {}
    if f:
        rows = [r for r in csv.DictReader(open(f[0])) if "fit_kernel" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        # the dominant dispatch of the last step = the largest SQ_INSTS_VALU among the last two fit_kernel dispatches
        best, bv = None, -1
        for i in ids[-2:]:
            v = sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == i and r["Counter_Name"] == "SQ_INSTS_VALU")
            if v > bv: best, bv = i, v
        for r in rows:
            if int(r["Dispatch_Id"]) == best:
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    nv = line["config"]["fitted_voxels_per_gpu"]
    if os.environ.get("CTRS"):
        print("DEFS [%s] kernel_ms %.2f | " % (defs, line["roofline"]["kernel_ms"]) + " ".join("%s=%.4g" % (k, v) for k, v in sorted(tot.items())), flush=True)
        continue
    print("DEFS [%s] kernel_ms %.2f | per voxel: VALU %.0f SALU %.0f LDS %.0f | ACTIVE_VALU/WAVE_CYC %.3f WAIT_INST %.3f WAIT_ANY %.3f LDS_ACTIVE(quad-cyc/voxel) %.0f" % (
        defs, line["roofline"]["kernel_ms"], tot.get("SQ_INSTS_VALU", 0) / nv, tot.get("SQ_INSTS_SALU", 0) / nv, tot.get("SQ_INSTS_LDS", 0) / nv,
        tot.get("SQ_ACTIVE_INST_VALU", 0) / max(tot.get("SQ_WAVE_CYCLES", 1), 1), tot.get("SQ_WAIT_INST_ANY", 0) / max(tot.get("SQ_WAVE_CYCLES", 1), 1),
        tot.get("SQ_WAIT_ANY", 0) / max(tot.get("SQ_WAVE_CYCLES", 1), 1), tot.get("SQ_LDS_IDX_ACTIVE", 0) / nv), flush=True)
