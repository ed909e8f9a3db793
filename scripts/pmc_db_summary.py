"""Summarises rocprofv3 --pmc rocpd databases (one per pass) per kernel: counter sums per launch (summed over the XCD/SE instances the
tool reports), launch count and mean duration.  usage: pmc_db_summary.py out.json db1 [db2 ...]"""
import json
import re
import sqlite3
import sys

out_path, dbs = sys.argv[1], sys.argv[2:]
summary = {}
for path in dbs:
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, dispatch_id, counter_name, sum(counter_value), max(duration) from pmc_events group by name, dispatch_id, counter_name").fetchall()
    per = {}
    for name, dispatch, counter, value, duration in rows:
        name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("void ", "")).replace(".kd", "").strip()
        k = per.setdefault(name, {})
        k.setdefault(counter, []).append(value)
        k.setdefault("_duration_ns", {})[dispatch] = duration
    for name, counters in per.items():
        s = summary.setdefault(name, {})
        durations = list(counters.pop("_duration_ns").values())
        s["launches"] = len(durations)
        s.setdefault("avg_duration_us_profiled", {})[path.split("/")[-2]] = sum(durations) / len(durations) / 1e3
        for counter, values in counters.items():
            s[counter + "_per_launch"] = sum(values) / len(values)
json.dump(summary, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
