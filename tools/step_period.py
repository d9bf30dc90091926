import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "grx_fetch_step_kernel<GrxShape<22" in r["Kernel_Name"]]
per, dur, other = [], [], []
for a, b in zip(idx[-9:-1], idx[-8:]):
    s0, e0, s1 = int(rows[a]["Start_Timestamp"]), int(rows[a]["End_Timestamp"]), int(rows[b]["Start_Timestamp"])
    per.append((s1 - s0) / 1000); dur.append((e0 - s0) / 1000)
    other.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a + 1: b]) / 1000)
print(sys.argv[1], "period %.1f us  step kernel %.1f us  between: %.1f us of which kernels %.1f us (%d launches)" % (sum(per)/len(per), sum(dur)/len(dur), (sum(per)-sum(dur))/len(per), sum(other)/len(other), idx[-1]-idx[-2]-1))
