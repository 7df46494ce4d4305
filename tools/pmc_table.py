"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv.  usage: pmc_table.py <csv> [...]"""
import csv, collections, sys
for path in sys.argv[1:]:
    tot = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        tot[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        n[k] += 1
    print(path)
    for (k, c), v in sorted(tot.items()):
        if v > 1e6:
            print(f"  {k:40s} {c:32s} {v:.4g}")
