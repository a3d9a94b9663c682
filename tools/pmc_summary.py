#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter_collection.csv values per kernel: tools/pmc_summary.py DIR..."""
import collections
import csv
import glob
import json
import sys

out = {}
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out.setdefault(k, {}).update({c: round(sum(x) / len(x), 1) for c, x in v.items()})
print(json.dumps(out, indent=1))
