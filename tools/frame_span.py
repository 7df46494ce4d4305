"""Per-frame wall span of the wavefront pipeline from a rocprofv3 kernel trace (kernels of a frame overlap on four streams, so
the per-kernel sums of --stats exceed the frame's time).  usage: frame_span.py <kernel_trace.csv> [out.json]"""
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "trace_kernel" in r["Kernel_Name"] or "shade_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frames, cur = [], []
for r in rows:
    # a frame starts with a shade kernel after a gap in which init_groups ran: detect by counting launches per frame
    cur.append(r)
# frames are delimited by the init_groups kernel
marks = sorted(int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if "init_groups" in r["Kernel_Name"])
starts = []
for t in marks:  # one frame = the init_groups launches closer than 1 ms to each other
    if not starts or t - starts[-1][-1] > 1_000_000: starts.append([t])
    else: starts[-1].append(t)
bounds = [s[0] for s in starts] + [1 << 62]
out = []
for i in range(len(starts)):
    fr = [r for r in rows if bounds[i] <= int(r["Start_Timestamp"]) < bounds[i + 1]]
    if not fr: continue
    span = (max(int(r["End_Timestamp"]) for r in fr) - bounds[i]) / 1e6
    per = {}
    for r in fr:
        k = r["Kernel_Name"].replace("void ", "").replace("(WfArgs)", "")
        per.setdefault(k, [0, 0.0]); per[k][0] += 1; per[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out.append({"frame": i, "span_ms": span, "launches": len(fr), "kernel_sum_ms": {k: round(v[1], 3) for k, v in per.items()},
                "kernel_calls": {k: v[0] for k, v in per.items()}})
for o in out: print(json.dumps(o))
if len(sys.argv) > 2: json.dump(out, open(sys.argv[2], "w"), indent=1)
