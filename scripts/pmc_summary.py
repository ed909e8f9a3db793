"""Summarises rocprofv3 --pmc CSV outputs (one directory per pass) into profiles/*_pmc_summary.json.
usage: pmc_summary.py out.json dir1 [dir2 ...]   (every *counter_collection.csv below the directories is read)"""
import csv
import glob
import json
import os
import re
import sys

out_path, dirs = sys.argv[1], sys.argv[2:]
acc = {}
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = re.sub(r"\(.*$", "", row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip()
                k = acc.setdefault(name, {})
                c = k.setdefault(row["Counter_Name"], [0.0, set()])
                c[0] += float(row["Counter_Value"])
                c[1].add(row["Dispatch_Id"])
summary = {}
for name, counters in acc.items():
    summary[name] = {}
    for cname, (total, ids) in counters.items():
        summary[name][cname + "_avg_per_launch"] = total / max(1, len(ids))
        summary[name]["launches"] = len(ids)
json.dump(summary, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
