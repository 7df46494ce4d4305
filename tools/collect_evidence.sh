#!/bin/bash
# tools/collect_evidence.sh PREFIX -- every measurement profiles/ holds for one version of the kernel sources, in one gpurun call:
# the counter passes of one C3 frame (one counter block per pass; two TA counters at most), a kernel trace of bench.py, the bench
# lines of C1-C3, the role profile (needs prt_amd/lib/var/libprt_hip_prof.so: tools/build_variants.py prof:-DPRT_PROFILE=1) and the
# rank-share projection.  Output under gpurun_out/PREFIX_*; tools/pmc_evidence.py turns the passes into the tracked JSON.
set -e
P=$1
WHAT=${2:-all} # passes | rest | all (`rest` after tools/pmc_evidence.py has written the JSON bench.py reads its traffic from)
R=$GRAFT_REPO_ROOT
cd $R
if [ $WHAT != rest ]; then
bash tools/pmc_pass.sh ${P}_sq "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
bash tools/pmc_pass.sh ${P}_sq2 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
bash tools/pmc_pass.sh ${P}_tcp "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
bash tools/pmc_pass.sh ${P}_tcc "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum"
bash tools/pmc_pass.sh ${P}_fetch "FETCH_SIZE"
bash tools/pmc_pass.sh ${P}_write "WRITE_SIZE"
bash tools/pmc_pass.sh ${P}_ta "TA_BUSY_avr TA_TA_BUSY_sum"
bash tools/pmc_pass.sh ${P}_ta2 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
bash tools/pmc_pass.sh ${P}_grbm "GRBM_COUNT GRBM_GUI_ACTIVE"
bash tools/pmc_pass.sh ${P}_ic "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH"
fi
if [ $WHAT = passes ]; then exit 0; fi
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/${P}_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_trace -o run -- python3 $R/bench.py --steps 3 --warmup 1 > $R/gpurun_out/${P}_trace/bench.json 2> $R/gpurun_out/${P}_trace/err.txt
cd $R
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/${P}_bench_c3.json 2> gpurun_out/${P}_bench_c3.err
timeout -k 10 200 python bench.py --workload c2_bunny_standin --steps 10 --warmup 3 > gpurun_out/${P}_bench_c2.json 2> gpurun_out/${P}_bench_c2.err
timeout -k 10 200 python bench.py --workload c1_cornell_teapot --steps 20 --warmup 5 > gpurun_out/${P}_bench_c1.json 2> gpurun_out/${P}_bench_c1.err
if [ -f prt_amd/lib/var/libprt_hip_prof.so ]; then timeout -k 10 200 python tools/role_profile.py prt_amd/lib/var/libprt_hip_prof.so > gpurun_out/${P}_roles.txt 2>&1; fi
timeout -k 10 300 python tools/rank_share_experiment.py > gpurun_out/${P}_rank_share.txt 2>&1
echo evidence collected
