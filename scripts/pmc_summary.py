"""Summarise rocprofv3 --pmc csv output: python scripts/pmc_summary.py gpurun_out/pmc_<tag>_*"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "sweep" not in k:
                continue
            print(d, k)
            for c, v in cs.items():
                print(f"   {c:36s} n={len(v)} mean={sum(v)/len(v):.6g}")
