import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:44]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M " + r.get("Direction", "copy")))
ev.sort()
end = max(e[1] for e in ev); lo = end - int(170e6)
ev = [e for e in ev if e[0] >= lo and (e[1] - e[0] > 100000 or "fit_kernel" in e[2])]
t0 = ev[0][0]
for s, e, n in ev:
    print("%8.2f -> %8.2f  (%7.2f ms)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
