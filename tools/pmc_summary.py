"""Summarise the FETCH_SIZE / WRITE_SIZE passes of one C3 frame (tools/pmc_frame.py under rocprofv3 --pmc) into the
`traffic` figure of the bench line.  Corrections (MI355X_MICROARCH.md "HBM"; tools/fetch_calib.hip on this box):
  * both counters are in KiB;
  * FETCH_SIZE reports 1/2 of the bytes of coalesced 16-B-per-lane streaming reads (calibration: 1 048 580 KiB for 2 GiB) -> x2 for
    shade_kernel, whose reads are the streamed slot state;
  * it reports 64-byte-record gathers exactly (calibration: 2 154 840 KiB for 2 GiB of 64-B records read as 4 x dwordx4) -> x1 for
    the trace kernels;
  * WRITE_SIZE is exact for 16-B-per-lane stores.
The counters sit on the L2's fabric side: Infinity-Cache hits are included, so this is an upper bound on DRAM bytes.
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv, json, sys
def per_kernel(path):
    d = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("void ", "").replace("(WfArgs)", "")
        d[k] = d.get(k, 0.0) + float(r["Counter_Value"])
    return d
f, w = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
KiB = 1024.0
fetch = sum(v * (2.0 if "shade" in k else 1.0) for k, v in f.items()) * KiB
write = sum(w.values()) * KiB
out = {"workload": "c3_sponza_standin", "frame": "1920x1080, 64 spp, depth 8, one GPU",
       "FETCH_SIZE_KiB_raw": {k: round(v, 1) for k, v in f.items()}, "WRITE_SIZE_KiB": {k: round(v, 1) for k, v in w.items()},
       "fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes_per_frame": fetch + write}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("fetch_bytes_corrected", "write_bytes", "traffic_bytes_per_frame")}))
