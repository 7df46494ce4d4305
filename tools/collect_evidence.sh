#!/bin/bash
# tools/collect_evidence.sh PREFIX -- the measurements profiles/ holds besides the counter passes (tools/collect_counters.sh), in one
# gpurun call: a kernel trace of bench.py (rocprofv3 --kernel-trace --stats), the bench lines of C1-C4, the rank-share projection.
# Steps are joined with &&: a step that is killed ends the call.  Output under gpurun_out/PREFIX_*.
set -e
P=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/${P}_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_trace -o run -- python3 $R/bench.py --steps 5 --warmup 2 > $R/gpurun_out/${P}_trace/bench.json 2> $R/gpurun_out/${P}_trace/err.txt &&
cd $R &&
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > gpurun_out/${P}_bench_c3.json 2> gpurun_out/${P}_bench_c3.err &&
timeout -k 10 200 python bench.py --workload c2_bunny_standin --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${P}_bench_c2.json 2> gpurun_out/${P}_bench_c2.err &&
timeout -k 10 200 python bench.py --workload c1_cornell_teapot --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${P}_bench_c1.json 2> gpurun_out/${P}_bench_c1.err &&
timeout -k 10 300 python bench.py --workload c4_sanmiguel_standin --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${P}_bench_c4.json 2> gpurun_out/${P}_bench_c4.err &&
timeout -k 10 300 python tools/rank_share_experiment.py > gpurun_out/${P}_rank_share.txt 2>&1 &&
cat gpurun_out/${P}_rank_share.txt &&
python - <<PY
import json
for w in ("c3", "c2", "c1", "c4"):
    d = json.load(open("gpurun_out/${P}_bench_%s.json" % w))
    print(w, round(d["value"], 1), "Mray/s", round(d["ms_per_step"], 2), "ms", "frac", d["roofline"]["frac"], "of", (d["roofline"]["frac_of"] or "-")[:24], "hbm_frac", d["roofline"]["hbm_frac"], "cpu", d.get("cpu_baseline", {}).get("value"))
PY
