#!/usr/bin/env python3
"""Counter evidence for one bench workload, run ON THE GPU BOX:

    python3 scripts/collect_pmc.py --tag r02 --key config1 -- --config 1
    python3 scripts/collect_pmc.py --tag r02 --key config4_32k -- --config 4 --dims 32,32,32

Runs `python3 bench.py <args> --steps 1 --warmup 1 --no-cpu-baseline` under rocprofv3 four times, each in its own process with
one counter group (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with a trace
domain other than --kernel-trace):
    --kernel-trace --stats                          per-kernel durations
    --pmc FETCH_SIZE / --pmc WRITE_SIZE             HBM bytes (FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B)
    --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS ... / SQ_WAVE_CYCLES SQ_WAIT_ANY ... / SQ_INSTS_VALU_FMA_F64 ... /
          SQ_LDS_BANK_CONFLICT ... / SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES ...        instruction mix, waits, LDS, matrix cores
and writes gpurun_out/<tag>_<key>_{kernel_stats,pmc}.csv plus an entry of gpurun_out/pmc_counters.json
(copy both into profiles/ to have bench.py quote them; the entry carries the digest of the kernel sources it was taken on).
The numbers describe the LAST timed launch of the dominant kernel (the one with the largest total time).
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "multicomponent-t2-toolbox_amd"


def source_sha(files=("met2_hip.hip", "fit_kernel.hpp", "nnls_wave.hpp", "nnls_big.hpp", "objectives.hpp", "wave_ops.hpp")):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, PKG, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def run_prof(outdir, name, prof_args, bench_args, timeout):
    d = os.path.join(outdir, name)
    cmd = ["rocprofv3"] + prof_args + ["-d", d, "-o", "p", "--output-format", "csv", "--", "python3", os.path.join(ROOT, "bench.py")] + bench_args
    env = dict(os.environ, TMPDIR="/tmp")
    p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
    open(os.path.join(outdir, name + ".log"), "w").write(p.stdout[-20000:] + "\n=====\n" + p.stderr[-20000:])
    if p.returncode != 0:
        raise SystemExit("rocprofv3 pass %s failed (rc %d), see %s.log" % (name, p.returncode, os.path.join(outdir, name)))
    return d


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit("no %s under %s" % (suffix, d))
    return hits[0]


def counters(d, kernel_substr, steps=2, sum_step=False):
    """{counter: value} of the dominant dispatch of the last step of the kernel whose name contains kernel_substr.  A method
    with a clean-up pass launches the same kernel symbol twice per step (first pass, then the few voxels that hit the capacity):
    of the last step's dispatches the one with the largest first counter is reported."""
    rows = [r for r in csv.DictReader(open(find(d, "counter_collection.csv"))) if kernel_substr in r["Kernel_Name"]]
    if not rows:
        return {}, None
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    per_step = max(1, len(ids) // steps)
    last_step = ids[-per_step:]
    first_counter = rows[0]["Counter_Name"]
    best, best_val = last_step[-1], -1.0
    for i in last_step:
        v = sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == i and r["Counter_Name"] == first_counter)
        if v > best_val:
            best, best_val = i, v
    out = {}
    for r in rows:
        if (int(r["Dispatch_Id"]) in last_step) if sum_step else (int(r["Dispatch_Id"]) == best):
            out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out, rows[0]["Kernel_Name"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="rXX")
    ap.add_argument("--key", required=True, help="entry name in pmc_counters.json: config<N> for a BASELINE config at full size")
    ap.add_argument("--kernel", default="fit_kernel", help="substring of the kernel to report (default: the fit kernel)")
    ap.add_argument("--timeout", type=int, default=900)
    ap.add_argument("--sha-files", default="", help="comma list of csrc files whose digest the entry carries (default: the fit kernel's sources)")
    ap.add_argument("--sum", action="store_true", help="sum the counters over ALL dispatches of the kernel in the last step (a kernel launched once per pass of voxels)")
    ap.add_argument("bench", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    bench_args = [x for x in a.bench if x != "--"] + ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end"]
    out = os.path.join(ROOT, "gpurun_out")
    work = os.path.join(out, "%s_%s_prof" % (a.tag, a.key))
    os.makedirs(work, exist_ok=True)
    d = run_prof(work, "stats", ["--kernel-trace", "--stats"], bench_args, a.timeout)
    stats = list(csv.DictReader(open(find(d, "kernel_stats.csv"))))
    with open(os.path.join(out, "%s_%s_kernel_stats.csv" % (a.tag, a.key)), "w") as f:
        f.write(open(find(d, "kernel_stats.csv")).read())
    krows = [r for r in stats if a.kernel in r["Name"]]
    krows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    dominant = krows[0]["Name"] if krows else a.kernel
    sub = dominant.split("(")[0].replace("void ", "")
    groups = [("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
              ("sq1", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]),
              ("sq2", ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES"]),
              ("sq3", ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32",
                       "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"]),
              ("lds", ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS_LOAD", "SQ_INSTS_LDS_STORE", "SQ_WAIT_INST_LDS"]),
              ("mfma", ["SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_COEXEC_CYCLES"])]
    allc = {}
    for name, ctrs in groups:
        try:
            dd = run_prof(work, name, ["--kernel-trace", "--pmc"] + ctrs, bench_args, a.timeout)
            c, kn = counters(dd, sub, sum_step=a.sum)
            allc.update(c)
        except SystemExit as e:
            print("pass %s skipped: %s" % (name, e), file=sys.stderr)
    with open(os.path.join(out, "%s_%s_pmc.csv" % (a.tag, a.key)), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
        for k in sorted(allc):
            w.writerow([dominant, k, allc[k]])
    # voxels of the launch: from a plain bench run's JSON line
    p = subprocess.run(["python3", os.path.join(ROOT, "bench.py")] + bench_args, capture_output=True, text=True, timeout=a.timeout)
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    ent = {"tag": a.tag, "src_sha": source_sha(tuple(a.sha_files.split(","))) if a.sha_files else source_sha(), "kernel": dominant, "voxels": line["config"]["voxels_per_gpu"],
           "files": "%s_%s_pmc.csv, %s_%s_kernel_stats.csv" % (a.tag, a.key, a.tag, a.key),
           "kernel_avg_ms_rocprof": float(krows[0]["AverageNs"]) / 1e6 if krows else None,
           "kernel_max_ms_rocprof": float(krows[0]["MaxNs"]) / 1e6 if krows and "MaxNs" in krows[0] else None,
           "kernel_ms_hip_events": line["roofline"].get("fa_kernel_ms") if "fa_kernel" in a.kernel else line["roofline"].get("kernel_ms", line["roofline"].get("iteration_phase_ms")),
           "bench_args": " ".join(bench_args)}
    if "FETCH_SIZE" in allc and "WRITE_SIZE" in allc:
        ent["fetch_kb_raw"] = allc["FETCH_SIZE"]; ent["write_kb_raw"] = allc["WRITE_SIZE"]
        ent["hbm_bytes_per_launch"] = 1024.0 * (2.0 * allc["FETCH_SIZE"] + allc["WRITE_SIZE"])
    for src, dst in (("SQ_INSTS_VALU", "valu_insts_per_launch"), ("SQ_INSTS_SALU", "salu_insts_per_launch"), ("SQ_INSTS_LDS", "lds_insts_per_launch"),
                     ("SQ_INSTS_MFMA", "mfma_insts_per_launch"), ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_cycles"), ("SQ_WAIT_ANY", "sq_wait_any"),
                     ("SQ_INSTS_VALU_FMA_F64", "valu_fma_f64"), ("SQ_INSTS_VALU_MUL_F64", "valu_mul_f64"), ("SQ_INSTS_VALU_ADD_F64", "valu_add_f64"),
                     ("SQ_INSTS_VALU_TRANS_F64", "valu_trans_f64"), ("SQ_INSTS_VALU_INT32", "valu_int32"), ("SQ_INSTS_VALU_INT64", "valu_int64"),
                     ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles"), ("SQ_LDS_IDX_ACTIVE", "lds_active_cycles"),
                     ("SQ_WAVE_CYCLES", "sq_wave_cycles"), ("SQ_ACTIVE_INST_VALU", "sq_active_inst_valu"), ("SQ_WAIT_INST_ANY", "sq_wait_inst_any")):
        if src in allc:
            ent[dst] = allc[src]
    jf = os.path.join(out, "pmc_counters.json")
    cur = json.load(open(jf)) if os.path.exists(jf) else {}
    if not cur and os.path.exists(os.path.join(ROOT, "profiles", "pmc_counters.json")):
        cur = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json")))
    cur[a.key] = ent
    json.dump(cur, open(jf, "w"), indent=1)
    import shutil
    shutil.rmtree(work, ignore_errors=True)          # the raw traces are tens of MB per pass; the summaries above are what is kept
    print(json.dumps(ent, indent=1))


if __name__ == "__main__":
    main()
