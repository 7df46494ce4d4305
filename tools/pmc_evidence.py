"""Summarise the rocprofv3 --pmc passes of one frame (tools/pmc_frame.py: C3, C4 or one rank's share of C5) (tools/pmc_pass.sh, one counter block per pass) into ONE tracked JSON:
what bounds the frame kernel, with every figure DESIGN.md quotes.  Conventions (MI355X_MICROARCH.md):
  * SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, summed over waves; SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU =
    lanes active per vector instruction;
  * FETCH_SIZE / WRITE_SIZE are KiB at the L2's fabric side (Infinity-Cache hits included).  FETCH_SIZE reads 64-byte-record
    gathers exactly (profiles/r01_fetch_size_calibration.csv) and 16-byte-per-lane streaming reads at 1/2: the frame kernel's
    reads are > 95 % record gathers (node, triangle and surface records; the streamed slot state is 0.2 KB per ray), so the raw
    figure is used and `fetch_bytes_upper` doubles the state share as the upper bound;
  * GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 / kernel time = effective clock; TA_BUSY_avr = busy cycles of the average
    texture-address unit.
usage: pmc_evidence.py <gpurun_out prefix, e.g. gpurun_out/r2_p> <out.json>"""
import csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prefix, out = sys.argv[1], sys.argv[2]
C, ms = {}, []
for suffix in ("sq", "sq2", "tcp", "tcc", "fetch", "write", "ta", "ta2", "grbm"):
    path = f"{prefix}_{suffix}/run_counter_collection.csv"
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        if "frame_kernel" in r["Kernel_Name"]:
            C[r["Counter_Name"]] = C.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    m = re.search(r"frame kernel ms ([0-9.]+) rays (\d+)(?: workload (\S+) (\S+))?", open(f"{prefix}_{suffix}/log.txt").read())
    if m:
        ms.append(float(m.group(1)))
        rays = int(m.group(2))
        workload, frame = (m.group(3), m.group(4)) if m.group(3) else ("c3_sponza_standin", "1920x1080,64spp,depth8")
kernel_ms = sorted(ms)[len(ms) // 2]
stamps = set(re.findall(r"library source_sha16 (\S+)", "".join(open(f"{prefix}_{x}/log.txt").read() for x in ("sq", "tcc", "fetch") if os.path.exists(f"{prefix}_{x}/log.txt"))))
import prt_amd as _p
assert len(stamps) <= 1, f"the passes ran libraries built from different sources: {stamps}"
stamp = stamps.pop() if stamps else _p.source_sha16()  # what the passes RAN is what the summary is stamped with
g = lambda k: C.get(k, float("nan"))
clock = g("GRBM_GUI_ACTIVE") / 8 / (kernel_ms * 1e-3)
fetch, write = g("FETCH_SIZE") * 1024, g("WRITE_SIZE") * 1024
import prt_amd
d = {
    "workload": workload, "frame": frame + ", one GPU", "kernel": "frame_kernel<false, false>",
    # (tools/pmc_frame.py prints the loaded library's stamp into each pass's log)
    "source_sha16": stamp, "kernel_ms_under_profiler_median": kernel_ms, "rays_per_frame": rays,
    "counters_raw": {k: C[k] for k in sorted(C)},
    "derived": {
        "lanes_active_per_valu_instruction": g("SQ_THREAD_CYCLES_VALU") / g("SQ_INSTS_VALU"),
        "wave_time_waiting_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
        "wave_time_issue_stall_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
        "wave_time_issuing_frac (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
        "valu_instructions_per_ray": g("SQ_INSTS_VALU") / rays,
        "salu_instructions_per_ray": g("SQ_INSTS_SALU") / rays,
        # one wave64 vector instruction occupies a SIMD for 3.07 cycles with 8 waves resident (profiles/r02_valu_rate.txt); 256 CUs x 4 SIMDs
        "valu_busy_frac (SQ_INSTS_VALU x 3.07 cycles / 1024 SIMDs / frame cycles)": g("SQ_INSTS_VALU") * 3.07 / 1024.0 / (clock * kernel_ms * 1e-3),
        "vmem_read_instructions_per_ray": g("SQ_INSTS_VMEM_RD") / rays,
        "effective_clock_GHz": clock / 1e9,
        "ta_busy_frac (TA_BUSY_avr / cycles of the frame)": g("TA_BUSY_avr") / (g("GRBM_GUI_ACTIVE") / 8),
        "ta_stalled_by_cache_frac_of_busy": g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") / g("TA_TA_BUSY_sum"),
        "l1_accesses_per_ray (TCP_TOTAL_CACHE_ACCESSES)": g("TCP_TOTAL_CACHE_ACCESSES_sum") / rays,
        "l1_hit_rate": 1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum"),
        "l2_requests_per_ray": g("TCC_REQ_sum") / rays,
        "l2_hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")),
        "l2_misses_per_ray": g("TCC_MISS_sum") / rays,
        # the roofs that can bind this kernel, each a measured quantity <= 1:
        #   useful share of the vector ALU = busy fraction x lanes that do work per instruction / 64
        "valu_useful_frac (valu_busy_frac x lanes_active / 64)": g("SQ_INSTS_VALU") * 3.07 / 1024.0 / (clock * kernel_ms * 1e-3) * g("SQ_THREAD_CYCLES_VALU") / g("SQ_INSTS_VALU") / 64.0,
        #   L2 misses per second against the measured ceiling of dependent 64-byte gathers from beyond the caches at this kernel's
        #   occupancy (profiles/r03_rec_gather.txt: 52.9 G/s from a 2 GB table, 55.9 G/s from 256 MB)
        "l2_miss_rate_Gps": g("TCC_MISS_sum") / (kernel_ms * 1e-3) / 1e9,
        "gather_roof_frac (l2_miss_rate / 52.9 G per second)": g("TCC_MISS_sum") / (kernel_ms * 1e-3) / 52.9e9,
    },
    "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes_per_frame": fetch + write,
    "traffic_GBps": (fetch + write) / (kernel_ms * 1e-3) / 1e9, "hbm_frac_of_8TBps": (fetch + write) / (kernel_ms * 1e-3) / 8e12,
}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d["derived"], indent=1))
print("traffic", d["traffic_GBps"], "GB/s =", d["hbm_frac_of_8TBps"], "of peak")
