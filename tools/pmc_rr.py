"""summarise rocprofv3 --pmc counter_collection.csv files of the register-resident kernels: per kernel name, counter means per dispatch"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"]
            if "ff_block" in n or "tblock_rr" in n:
                acc["ff_block" if "ff_block" in n else "tblock_rr"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 2:]                      # later dispatches (warm)
        print(f"   {c:28s} {sum(v) / len(v):16.0f}   (n={len(v)})")
