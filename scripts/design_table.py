"""The rows of DESIGN.md section 5's round table, from the committed bench lines (so the numbers are copied by a program, not by hand).
usage: python scripts/design_table.py [tag]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
ROWS = [("", "C2 default, 6000 steps"), ("_driver_window_20_steps", "driver window"), ("_driver_window_slices_in_phase", "window, --stagger 0"),
        ("_serial_solver", "serial solver"), ("_trained_like_policy", "trained-like"), ("_c3_standard_10x128_800", "C3"), ("_c4_caro5_20x20", "C4"),
        ("_c5_renju_1600", "C5"), ("_soak_30000_steps", "soak"), ("_2_ranks_on_one_gpu", "2 ranks / 1 GPU"), ("_8_ranks_on_one_gpu", "8 ranks / 1 GPU"),
        ("_profiled", "profiled"), ("_pmc_workload", "pmc workload")]
for suffix, name in ROWS:
    path = os.path.join(ROOT, "profiles", "%s_bench_line%s.json" % (tag, suffix))
    if not os.path.exists(path):
        continue
    d = json.loads([x for x in open(path) if x.startswith("{")][0])
    k, r = d["kernel_ms_per_step"], d["roofline"]
    search = k.get("k_search_spec") or k.get("k_solve", 0.0)
    print("| %s | %.1f k | %.1f | %.0f | %.2f | %.2f | %.2f | %.2f | %.0f = %.2f | %.3f | host %.2f CPUs | %s" % (
        name, d["value"] / 1e3, d.get("games_per_sec", 0), d.get("moves_per_sec", 0), d["ms_per_step"], search, k.get("nn_tower", 0), k.get("k_expand", 0) + k.get("k_advance", 0),
        r["achieved"], r["frac"], r.get("time_averaged_whole_chip_frac") or 0, d["ranks"][0]["host_cpu_utilisation"],
        "pmc: busy %s clock %s traffic %s" % (r.get("mfma_busy_fraction_pmc"), r.get("shader_clock_mhz_pmc"), r.get("traffic"))))
m = os.path.join(ROOT, "profiles", "%s_match_bench_line.json" % tag)
if os.path.exists(m):
    print("match:", json.load(open(m))["simulations_per_sec"])
c = json.loads([x for x in open(os.path.join(ROOT, "profiles", "%s_bench_line.json" % tag)) if x.startswith("{")][0])
print("cpu baseline:", round(c["cpu_baseline"]["value"]), c["cpu_baseline"]["cores"], round(c["cpu_baseline"]["per_thread"]), c["cpu_baseline"]["cpu_utilisation"])
print("threads:", c["ranks"][0].get("host_threads_cpu_seconds"), "spec:", c["speculative_solver"], c["source_hash"])
