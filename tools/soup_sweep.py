"""Extended seeded sweep of tests/test_gpu_parity.py::test_random_soups_match_oracle (GPU against the oracle, bit for bit, event
counters included) over seeds the suite does not run.  usage: soup_sweep.py FIRST LAST"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import prt_amd
import test_gpu_parity as G
import prt_testlib as T
T.oracle().orc_set_anyhit_accounting(1)  # the oracle counts occlusion queries the way the GPU visits them (and checks both visits agree)
tr = prt_amd.PathTracer(device=0)
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    try:
        G.test_random_soups_match_oracle.__wrapped__(tr, seed) if hasattr(G.test_random_soups_match_oracle, "__wrapped__") else G.test_random_soups_match_oracle(tr, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:300], flush=True)
    if seed % 10 == 0:
        print("seed", seed, "done", flush=True)
print("sweep finished,", bad, "failures", flush=True)
sys.exit(1 if bad else 0)
