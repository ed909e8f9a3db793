"""Per-phase picture of the search launch (k_search_spec<false, 15>) -> profiles/<tag>_search_phases.json.

Three sources, none of them a guess:
  * in-kernel stamps of a profile build (scripts/solver_profile.sh: -DAGX_SOLVER_PROFILE -DAGX_SPEC_PROFILE; shader cycles / 100 MHz ticks per solve,
    summed over the run) -> where the TIME of a solve goes;
  * the kernel's ISA with a line table (hipcc -gline-tables-only --save-temps on an AGX_QUICK build, made here) -> how many VALU / SALU / LDS /
    VMEM instructions every source function CONTRIBUTES to the kernel (static; everything is inlined, the line table keeps the callee's lines);
  * rocprofv3 --pmc totals of the same kernel in the bench run (scripts/profile_bench.sh) -> instructions per launch and per solver node (dynamic,
    whole kernel).
  * optionally scripts/place_pmc.py's counts of ONE place / undo (the debug kernel on 2048 positions, by difference of two launches).
The bench line should be an un-profiled run of the PMC passes' own command (bench.py --steps 40 --warmup 30 --age-steps 0): its solver nodes per
launch turn the per-launch counters into per-node ones.
usage: python scripts/search_phases.py OUT.json PROFILE_STDERR [PMC_SUMMARY.json BENCH_LINE.json [PLACE_PMC.json]]"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "alphagomoku_amd", "csrc")
KERNEL = "_ZN12_GLOBAL__N_113k_search_specILb0ELi15EEEvN3agx9EngineDevEi"

PHASES = collections.OrderedDict([
    ("select (Search::select on the device tree)", ["select_batch", "select_edge", "cache_seek", "cache_seek_head", "has_leak", "correct_information_leak", "node_head",
                                                    "cancel_virtual_loss", "make_root_noise", "use_game_arenas"]),
    ("set_board + feature encode + hash", ["solver_set_board", "solver_encode_features", "solver_encode_forbidden", "solve_task"]),
    ("place / undo (PatternCalculator::addMove, undoMove)", ["solver_place", "solver_update_around", "pattern_prefetch", "list_get", "list_set", "line_of",
                                                             "extended_pattern", "normal_pattern", "narrow", "threat_lookup", "threat_index"]),
    ("move generation (MoveGenerator::generate)", ["generate", "try_win_in_1", "try_draw_in_1", "defend_loss_in_2", "try_win_in_3", "defend_loss_in_4", "finish_defend_loss_in_4",
                                                   "defend_loss_in_4_renju", "try_win_in_5", "defend_loss_in_6", "add_own_half_open_fours", "add_own_4x3_forks",
                                                   "try_solve_own_fork_4x3", "add_list", "add_move", "push", "defensive_mask", "defensive_moves", "get_defensive_moves",
                                                   "intersect", "intersect_init", "stencil_row", "create_remaining_moves", "mark_neighborhood", "mark_forbidden_moves",
                                                   "count", "count_of", "direction_of", "patterns", "threat_at", "item", "has_any_four", "available_fours", "is_foul", "copy_list",
                                                   "move_of", "MoveGen", "add", "at", "contains", "remove", "remove_at", "SmallSet", "promotion_moves", "straight_four_at",
                                                   "renju_is_forbidden"]),
    ("alpha-beta frame machine (recursive_solve: pick, descend, return)", ["solver_run", "act_get", "act_set", "act_find_move", "frame_get", "frame_set", "s_invert_step",
                                                                             "s_invert_up", "s_invert_down", "s_negate", "s_make", "s_pv", "s_eval", "s_proven", "s_unproven",
                                                                             "s_win", "s_loss", "s_infinite", "s_distance", "zobrist_word", "solver_evaluate",
                                                                             "wave_reduce_umax", "wave_reduce_add", "wave_reduce_xor64", "wave_scan32_add"]),
    ("transposition table + overlay (SharedHashTable)", ["tt_seek", "tt_insert", "tt_pack", "ov_find", "ov_create"]),
    ("queue, commit, scheduleToNN", ["k_search_spec", "spec_commit_game", "spec_flush", "schedule_to_nn", "symmetry_source", "shuffle_feature_directions", "symmetry_mix",
                                     "__launch_bounds__"]),
])


def isa_with_line_table(workdir):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-DAGX_QUICK",
           "-gline-tables-only", "-I" + os.path.join(ROOT, "include"), "--save-temps", "-c", os.path.join(CSRC, "engine.hip"), "-o", os.path.join(workdir, "engine_quick.o")]
    subprocess.check_call(cmd, cwd=workdir, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(workdir, "engine-hip-amdgcn-amd-amdhsa-gfx950.s")


def function_of_line(name):
    out, cur = {}, "?"
    for i, line in enumerate(open(os.path.join(CSRC, name)).read().split("\n"), 1):
        m = re.search(r"__(?:device|global)__[^;(]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", line)
        if m and not line.strip().startswith("//") and ";" not in line.split("(")[0]:
            cur = m.group(1)
        out[i] = cur
    return out


def static_mix(asm_path):
    lines = open(asm_path).read().split("\n")
    files = {}
    for line in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    start = next(i for i, x in enumerate(lines) if x.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    fmap = {n: function_of_line(n) for n in ("dev_solver.hpp", "engine.hip", "dev_mcts.hpp", "root_noise.hpp", "symmetry.hpp")}
    per = collections.defaultdict(collections.Counter)
    cur = None
    for line in lines[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        t = line.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        f, ln = cur if cur else ("?", 0)
        fn = fmap.get(f, {}).get(ln, "(no source line)" if ln == 0 else f)
        op = t.split()[0]
        kind = ("scratch" if op.startswith("scratch_") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
                else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other")
        per[fn][kind] += 1
        per[fn]["all"] += 1
    return per


def parse_profile(path):
    text = open(path).read()
    out = {}

    def last(pattern):
        found = re.findall(pattern, text)
        return found[-1] if found else None
    m = last(r"\[solver profile, 100 MHz ticks\] solves (\d+): set_board\+encode ([\d.]+) us, frame machine ([\d.]+) us, place/remove ([\d.]+) us \(([\d.]+) per solve, ([\d.]+) us each\), total ([\d.]+) us per solve")
    if m:
        out["solves"], out["set_board_encode_us"], out["frame_machine_us"], out["place_undo_us"], out["places_per_solve"], out["us_per_place"], out["total_us_per_solve"] = \
            int(m[0]), *[float(x) for x in m[1:]]
    m = last(r"\[frame machine split, us per solve\] table seek ([\d.]+), move generation ([\d.]+), ordering ([\d.]+), evaluate\+insert ([\d.]+)")
    if m:
        out["table_seek_us"], out["move_generation_us"], out["ordering_us"], out["evaluate_insert_us"] = [float(x) for x in m]
    m = last(r"\[generate\(\), shader cycles per solve; ([\d.]+) calls\] win1 (\d+), loss2 (\d+), win3 (\d+), loss4 (\d+), win5 (\d+), loss6 (\d+), half4 (\d+), rest (\d+)")
    if m:
        out["generate_calls_per_solve"] = float(m[0])
        out["generate_stage_cycles"] = dict(zip(["win_in_1", "defend_loss_in_2", "win_in_3", "defend_loss_in_4", "win_in_5", "defend_loss_in_6", "own_half_open_fours", "rest"],
                                                [int(x) for x in m[1:]]))
    m = last(r"\[frame machine, shader cycles per solve; ([\d.]+) loop turns, ([\d.]+) picks\] resume (\d+), pick (\d+), descend (\d+), child-returned (\d+)")
    if m:
        out["machine_turns_per_solve"], out["picks_per_solve"] = float(m[0]), float(m[1])
        out["machine_cycles"] = dict(zip(["resume", "pick", "descend", "child_returned"], [int(x) for x in m[2:]]))
    m = last(r"\[update_around, shader cycles per solve; ([\d.]+) calls, ([\d.]+) list changes per call\] centre (\d+), gather (\d+), lists (\d+)")
    if m:
        out["update_around_calls_per_solve"], out["list_changes_per_call"] = float(m[0]), float(m[1])
        out["update_around_cycles"] = dict(zip(["gather", "list_edits"], [int(m[3]), int(m[4])]))
    m = last(r"\[renju, per solve\] forbidden bits of the network input ([\d.]+) us, mark_forbidden_moves (\d+) shader cycles, generator foul tests (\d+) shader cycles \(([\d.]+) tests\)")
    if m and float(m[3]) > 0:
        out["renju"] = dict(forbidden_bits_us=float(m[0]), mark_forbidden_moves_cycles=int(m[1]), foul_test_cycles=int(m[2]), foul_tests_per_solve=float(m[3]))
    m = last(r"\[k_search_spec profile, wave-ms summed\] select phase ([\d.]+) \(slowest wave, max over launches ([\d.]+) ms\), waiting for items ([\d.]+), speculative solves ([\d.]+) "
             r"\((\d+), ([\d.]+) us each\), commits \+ re-runs ([\d.]+) \(longest ([\d.]+) ms\), longest launch ([\d.]+) ms")
    if m and float(m[3]) > 0:
        out["launch_wave_ms"] = dict(select=float(m[0]), waiting_for_items=float(m[2]), speculative_solves=float(m[3]), commits_and_reruns=float(m[6]))
        out["speculative_solves"], out["us_per_speculative_solve"] = int(m[4]), float(m[5])
    return out


def main():
    out_path, profile = sys.argv[1], sys.argv[2]
    pmc = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
    bench = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else None
    place = json.load(open(sys.argv[5])) if len(sys.argv) > 5 else None
    if os.environ.get("AGX_PHASES_NO_ISA"):   # (stamps only: e.g. the renju kernel, which an AGX_QUICK build does not instantiate)
        per = {}
    else:
        with tempfile.TemporaryDirectory() as tmp:
            per = static_mix(isa_with_line_table(tmp))
    known = {fn: phase for phase, fns in PHASES.items() for fn in fns}
    static = collections.OrderedDict((p, collections.Counter()) for p in PHASES)
    static["control flow without a source line (exec-mask save / restore of branches on uniform values held in VGPRs, spill code)"] = collections.Counter()
    static["other inlined helpers"] = collections.Counter()
    unassigned = collections.Counter()
    for fn, c in per.items():
        if fn in known:
            static[known[fn]] += c
        elif fn == "(no source line)":
            static[list(static.keys())[-2]] += c
        else:
            static["other inlined helpers"] += c
            unassigned[fn] += c["all"]
    stamps = parse_profile(profile)
    result = {"_comment": __doc__, "kernel": os.environ.get("AGX_PHASES_KERNEL", "k_search_spec<false, 15>"), "stamps_per_solve": stamps,
              "static_instructions_by_phase": {p: dict(c) for p, c in static.items()},
              "static_instructions_total": sum(c["all"] for c in per.values()),
              "largest_unassigned_helpers": dict(unassigned.most_common(8))}
    if stamps.get("total_us_per_solve"):
        t = stamps["total_us_per_solve"]
        result["time_share_of_a_solve"] = {
            "place / undo": stamps["place_undo_us"] / t, "move generation": stamps["move_generation_us"] / t, "ordering (pick + descend bookkeeping)": stamps["ordering_us"] / t,
            "table seek": stamps["table_seek_us"] / t, "evaluate + table insert": stamps["evaluate_insert_us"] / t, "set_board + encode": stamps["set_board_encode_us"] / t,
            "rest of the frame machine (resume, child returned, yields)":
                (stamps["frame_machine_us"] - stamps["move_generation_us"] - stamps["ordering_us"] - stamps["table_seek_us"] - stamps["evaluate_insert_us"]) / t}
    if pmc and bench:
        k = next((v for n, v in pmc.get("per_kernel_raw", {}).items() if n.startswith("k_search_spec<false, 15")), None)
        if k:
            launches_per_step = bench["slices"]["count"]
            nodes_per_launch = bench["roofline_solver"]["solver_nodes_per_sec"] * bench["ms_per_step"] * 1e-3 / launches_per_step if "roofline_solver" in bench else None
            dyn = {c.replace("_per_launch", ""): v for c, v in k.items() if c.startswith("SQ_INSTS") or c.startswith("SQ_WAVE") or c.startswith("SQ_ACTIVE") or c.startswith("SQ_WAIT")}
            result["pmc_per_launch"] = dyn
            if nodes_per_launch:
                result["solver_nodes_per_launch"] = nodes_per_launch
                result["wave_instructions_per_solver_node"] = {c: v / nodes_per_launch for c, v in dyn.items() if c.startswith("SQ_INSTS")}
    if place:
        result["one_place_or_undo_dynamic"] = {k: v for k, v in place.items() if k != "_comment"}
        if "wave_instructions_per_solver_node" in result and stamps.get("places_per_solve") and stamps.get("generate_calls_per_solve"):
            per_node = stamps["places_per_solve"] / stamps["generate_calls_per_solve"]   # (one generate() call per visited node)
            result["place_undo_share_of_a_node's_instructions"] = {
                c: place["per_place_or_undo"][c] * per_node / result["wave_instructions_per_solver_node"][c]
                for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS") if c in place["per_place_or_undo"]}
            result["places_and_undos_per_solver_node"] = per_node
    json.dump(result, open(out_path, "w"), indent=1)
    print(json.dumps({k: v for k, v in result.items() if k not in ("_comment",)}, indent=1)[:6000])


if __name__ == "__main__":
    main()
