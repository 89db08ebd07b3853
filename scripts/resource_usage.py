#!/usr/bin/env python3
"""Static per-kernel summaries of the current sources (runs here, no GPU):
    python3 scripts/resource_usage.py [TAG]  -> profiles/TAG_kernel_resource_usage.csv, profiles/TAG_mfma_disassembly.txt   (TAG defaults to r04)
Registers, spills and occupancy from hipcc -Rpass-analysis=kernel-resource-usage; v_mfma_f64_16x16x4 counts from the -S output."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "multicomponent-t2-toolbox_amd", "csrc", f) for f in ("met2_hip.hip", "met2_fit_x2_nb1.hip", "met2_fit_x2_nb2.hip", "met2_fit_x2_second.hip", "met2_fit_nnls_lcurve.hip",
                                                                                  "met2_fit_gcv.hip", "met2_fit_bayes.hip", "met2_tv.hip")]
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"


def source_sha(files=("met2_hip.hip", "fit_kernel.hpp", "nnls_wave.hpp", "nnls_big.hpp", "objectives.hpp", "wave_ops.hpp")):
    """the digest bench.py gates the file on (same list as bench.py / collect_pmc.py)"""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "multicomponent-t2-toolbox_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()


def main():
    tmp = tempfile.mkdtemp()
    blocks, s = [], ""
    for i, src in enumerate(SRCS):
        asm = os.path.join(tmp, "k%d.s" % i)
        p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-disable-machine-licm", "-DMET2_SPLIT_TU", "-S", "--cuda-device-only",
                            "-Rpass-analysis=kernel-resource-usage", "-o", asm, src], capture_output=True, text=True)
        blocks += re.split(r"remark: Function Name: ", p.stderr)[1:]
        s += open(asm).read()
    with open(os.path.join(ROOT, "profiles", TAG + "_kernel_resource_usage.csv"), "w") as f:
        f.write("# hipcc --offload-arch=gfx950 -O3 -mllvm -disable-machine-licm -Rpass-analysis=kernel-resource-usage (scripts/resource_usage.py), sources of the current evidence\n")
        f.write("# src_sha: %s\n" % source_sha())
        f.write("# kernel, VGPRs, AGPRs, SGPRs, SGPR spills, VGPR spills, scratch B/lane, occupancy waves/SIMD\n")
        for b in blocks:
            name = b.split()[0]
            def g(k):
                m = re.search(k + r": (\d+)", b)
                return int(m.group(1)) if m else 0
            f.write('"%s", %d, %d, %d, %d, %d, %d, %d\n' % (demangle(name), g("    VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("SGPRs Spill"), g("VGPRs Spill"),
                                                           g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
    rows, sample = [], None
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        hits = re.findall(r"v_mfma_f64_16x16x4[^\n]*", m.group(2))
        if hits:
            rows.append((demangle(m.group(1)), len(hits)))
            sample = sample or hits[:4]
    with open(os.path.join(ROOT, "profiles", TAG + "_mfma_disassembly.txt"), "w") as f:
        f.write("# hipcc --offload-arch=gfx950 -O3 -S: static count of v_mfma_f64_16x16x4_f64 instructions per kernel (scripts/resource_usage.py)\n")
        f.write("# GCV: the Gram contraction M = E E^T of objectives.hpp:gcv_trace_direct; BayesReg: the trailing updates of objectives.hpp:chol_full;\n")
        f.write("# every method at two bins per lane: the trailing updates of nnls_wave.hpp:refactor_blocked (four per inlined copy)\n")
        for r in sorted(rows):
            f.write("%-62s %d\n" % r)
        f.write("# e.g.\n")
        for h in sample or []:
            f.write("#   %s\n" % h.strip())
    print(len(blocks), "kernels,", len(rows), "with MFMA")


if __name__ == "__main__":
    main()
