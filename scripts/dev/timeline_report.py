"""Digest of a rocprofv3 kernel + memory-copy trace: the last `span_ms` of activity, busy time per class and the idle gaps."""
import csv, glob, sys
d = sys.argv[1]; span = float(sys.argv[2]) if len(sys.argv) > 2 else 140.0
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M " + r.get("Direction", r.get("Name", "copy"))))
ev.sort()
end = max(e[1] for e in ev); lo = end - int(span * 1e6)
ev = [e for e in ev if e[0] >= lo]
t0 = ev[0][0]
print("window %.1f ms, %d events" % ((end - t0) / 1e6, len(ev)))
# union busy time of kernels, and of everything
def union(evs):
    tot = 0; cur_s = cur_e = None
    for s, e, _ in evs:
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot
K = [e for e in ev if e[2][0] == "K"]; M = [e for e in ev if e[2][0] == "M"]
print("kernel busy %.1f ms, copy busy %.1f ms, any busy %.1f ms" % (union(K) / 1e6, union(M) / 1e6, union(ev) / 1e6))
# kernel idle gaps > 0.3 ms
prev_e = None
for s, e, n in K:
    if prev_e is not None and s - prev_e > 300000:
        print("  kernel gap %.2f ms at +%.1f ms before %s" % ((s - prev_e) / 1e6, (s - t0) / 1e6, n))
    prev_e = e if prev_e is None else max(prev_e, e)
agg = {}
for s, e, n in ev:
    a = agg.setdefault(n, [0, 0]); a[0] += e - s; a[1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %8.2f ms %5d x %s" % (t / 1e6, c, n))
print("first event +0.0: %s; first kernel at +%.1f ms; last kernel ends +%.1f ms; last event ends +%.1f ms" % (ev[0][2], (K[0][0] - t0) / 1e6, (max(k[1] for k in K) - t0) / 1e6, (end - t0) / 1e6))
