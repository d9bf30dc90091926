"""Per-dispatch timeline of the last steps of a rocprofv3 --kernel-trace (csv) run of bench.py: name, stream/queue, start relative to the step kernel, duration.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o b -- python bench.py ... ; python tools/step_timeline.py DIR [kernel substring] [min step-kernel us]"""
import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else "grx_fetch_step_kernel<GrxShape<22"
mind = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0      # only launches of at least this many microseconds count as step launches
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"] and (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 >= mind]
i0, i1 = idx[-3], idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0: i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-46s q %-3s start %9.1f us  dur %8.1f us" % (r["Kernel_Name"][:46], r.get("Queue_Id", "?"), (s - t0) / 1000, (e - s) / 1000))
