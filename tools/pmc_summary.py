"""Per-kernel averages of a rocprofv3 --pmc run (counter_collection.csv): python tools/pmc_summary.py DIR [name-substring]"""
import collections
import csv
import glob
import os
import re
import sys


def main(d, sub=""):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")
                if sub and sub not in name:
                    continue
                e = acc[name][r["Counter_Name"]]
                e[0] += 1
                e[1] += float(r["Counter_Value"])
    for name, cs in acc.items():
        n = max(v[0] for v in cs.values())
        print(f"{name[:120]}  dispatches={n}")
        for c, (k, s) in sorted(cs.items()):
            print(f"    {c:28s} {s / k:16.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
