"""rocprofv3 --pmc rocpd databases of bench.py (one per pass) -> the summary bench.py quotes (profiles/<tag>_pmc_summary.json).

Per kernel: counter sums per launch (summed over the XCD / SE instances the tool reports).  Derived, per MI355X_MICROARCH.md:
  * HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies a 128-B request as 64 B for wide coalesced reads; unit KB;
    Infinity-Cache hits are included);
  * MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs the launch may use x GRBM_GUI_ACTIVE / 8 XCDs); bench.py steps the pool as
    AGX_PROFILE_SLICES (default 4) slices on CU-masked streams, a tower launch then owns 1024 / slices SIMDs;
  * NB rocprofv3 serialises kernels while it collects counters: in these passes the slices' launches do NOT overlap, so the tower's clock
    here is that of a launch running alone on its part of an otherwise idle chip, not the clock it holds in the real, overlapped run;
  * solver issue-busy fraction = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES, parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES;
  * shader clock of a kernel = (GRBM_GUI_ACTIVE / 8 XCDs) / launch duration of the same dispatch (the tower runs power-limited well below the
    2.4 GHz the MFMA peak assumes, the solver at the full clock).
The summary carries the source hash of the build it was taken from (bench.py quotes it only when it matches the build it runs)."""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_path, dbs = sys.argv[1], sys.argv[2:]
per = {}
for path in dbs:
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, dispatch_id, counter_name, sum(counter_value), max(duration) from pmc_events group by name, dispatch_id, counter_name").fetchall()
    acc = {}
    for name, dispatch, counter, value, duration in rows:
        name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("void ", "")).replace(".kd", "").strip()
        k = acc.setdefault(name, {})
        k.setdefault(counter, []).append(value)
        if counter == "GRBM_GUI_ACTIVE" and duration:
            k.setdefault("_clock_mhz", []).append(value / 8.0 / duration * 1e3)
    for name, counters in acc.items():
        s = per.setdefault(name, {})
        clocks = counters.pop("_clock_mhz", None)
        if clocks:
            s["shader_clock_mhz"] = sum(clocks) / len(clocks)
        for counter, values in counters.items():
            s[counter + "_per_launch"] = sum(values) / len(values)
            s["launches"] = len(values)
import bench  # noqa: E402
slices = int(os.environ.get("AGX_PROFILE_SLICES", "4"))
summary = {"_comment": __doc__, "source_hash": bench.source_hash(), "command": "python3 bench.py --steps 40 --warmup 30 --age-steps 0 --no-cpu-baseline", "slices": slices,
           "per_kernel_raw": per}
tower = next((k for k in per if k.startswith("nn_tower_kernel<128, 15, 15")), None) or next((k for k in per if k.startswith("nn_tower_kernel<")), None)  # (C4: <128, 20, 20, ...>)
if tower:
    t = per[tower]
    if "FETCH_SIZE_per_launch" in t and "WRITE_SIZE_per_launch" in t:
        summary["nn_tower_bytes_per_launch_corrected"] = (2.0 * t["FETCH_SIZE_per_launch"] + t["WRITE_SIZE_per_launch"]) * 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES_per_launch" in t and "GRBM_GUI_ACTIVE_per_launch" in t:
        summary["nn_tower_mfma_busy_fraction"] = t["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] / (1024.0 / slices * t["GRBM_GUI_ACTIVE_per_launch"] / 8.0)
    if "shader_clock_mhz" in t:
        summary["nn_tower_shader_clock_mhz"] = t["shader_clock_mhz"]
    if "SQ_LDS_BANK_CONFLICT_per_launch" in t and t.get("SQ_LDS_IDX_ACTIVE_per_launch"):
        summary["nn_tower_lds_bank_conflict_fraction"] = t["SQ_LDS_BANK_CONFLICT_per_launch"] / t["SQ_LDS_IDX_ACTIVE_per_launch"]
solve = (next((k for k in per if k.startswith("k_search_spec<false, 15")), None) or next((k for k in per if k.startswith("k_solve<false, 15")), None)
         or next((k for k in per if k.startswith("k_search_spec<")), None) or next((k for k in per if k.startswith("k_solve<")), None))
summary["search_kernel"] = solve
if solve and "SQ_WAVE_CYCLES_per_launch" in per[solve]:
    t = per[solve]
    summary["k_solve_issue_busy_fraction"] = t["SQ_ACTIVE_INST_ANY_per_launch"] / t["SQ_WAVE_CYCLES_per_launch"]
    summary["k_solve_parked_fraction"] = t["SQ_WAIT_ANY_per_launch"] / t["SQ_WAVE_CYCLES_per_launch"]
    if "shader_clock_mhz" in t:
        summary["k_solve_shader_clock_mhz"] = t["shader_clock_mhz"]
    summary["k_solve_issue_stall_fraction"] = t["SQ_WAIT_INST_ANY_per_launch"] / t["SQ_WAVE_CYCLES_per_launch"]
json.dump(summary, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in summary.items() if k not in ("per_kernel_raw", "_comment")}, indent=1))
