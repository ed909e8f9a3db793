#!/usr/bin/env python3
"""bench.py — MCTS simulations/s of the device-resident self-play engine (BASELINE.json metric).

A "step" is one pass of the hot path over the whole game pool of this rank: select (<= max_batch_size PUCT descents per game)
-> threat solver + feature encode -> policy/value network on the scheduled positions -> edge generation, expand, backup ->
move decision / tree compaction.  Workload at N = 1: BASELINE.json configs[1] — freestyle 15x15, 6-block / 128-filter net,
400 playouts per move, 1024 parallel games on one MI355X, synthetic random openings, synthetic He-init weights.

Multi-GPU: games are independent, so each rank runs its own pool of 1024 games (weak scaling); there is no data-path
collective, torch.distributed is only used for the barrier and the max-over-ranks timing the contract asks for.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1: one rank per GPU.  Started by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK /
WORLD_SIZE in the environment) the ranks run as they are; started plainly (`python bench.py --gpus N`) the script launches that
command itself BEFORE anything touches a GPU and exits with its return code.  AGX_FORCE_DEVICE=d puts every rank on device d (the
N > 1 flow on a 1-GPU box).
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def nn_flops_per_position(desc):
    F, C, HW, D, blocks = desc["filters"], desc["in_channels"], desc["rows"] * desc["cols"], desc["value_hidden"], desc["blocks"]
    flops = 2 * HW * (25 * C * F + blocks * 2 * 9 * F * F + 9 * F * F + F + 4 * F) + 2 * 4 * HW * D + 6 * D
    if desc.get("action_values", 0):
        flops += 2 * HW * (9 * F * F + 3 * F)
    return flops


def source_hash():
    """sha256 over the sources libagx.so is built from (alphagomoku_amd/build.py): the library reports the hash it was compiled from
    (agx_build_hash) and committed PMC summaries carry the hash of the build they were taken from"""
    from alphagomoku_amd import build
    return build.source_hash()


def thread_cpu_seconds():
    """{thread id: (name, user + system CPU seconds)} of this process, from /proc/self/task"""
    out = {}
    ticks = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                stat = open("/proc/self/task/%s/stat" % tid).read()
            except OSError:
                continue
            name = stat[stat.index("(") + 1:stat.rindex(")")]
            fields = stat[stat.rindex(")") + 2:].split()
            out[int(tid)] = ("main thread" if int(tid) == os.getpid() else name + " (runtime thread)", (int(fields[11]) + int(fields[12])) / float(ticks))
    except OSError:
        pass
    return out


def host_cpu_limits():
    """what bounds a CPU leg on this box: the CPUs this process may run on (affinity), the cgroup CPU quota (cpu.max, cgroup v2; cfs quota, v1)
    and the CPU model"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota_cpus, quota_text = None, None
    try:
        quota_text = open("/sys/fs/cgroup/cpu.max").read().strip()
        q, period = quota_text.split()
        if q != "max":
            quota_cpus = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota_text = "%d %d" % (q, period)
            if q > 0:
                quota_cpus = q / period
        except (OSError, ValueError):
            pass
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cores, quota_cpus, quota_text, model


def cpu_baseline(args, max_seconds):
    """Times the CPU oracle (a scalar restatement of the reference search, oracle/) on this box's host cores with a stand-in
    evaluator (network cost = 0).  This is the ONLY place where bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    lib = ol.load()
    # the CPUs this process may run on (affinity view) and what the cgroup lets them consume (quota), not the machine's total
    cores, quota_cpus, quota_text, model = host_cpu_limits()
    usable = max(1, min(cores, int(quota_cpus)) if quota_cpus else cores)
    cfg = ol.default_search_config(max_batch_size=args.batch, max_simulations=args.sims, table_entries=4 * 1024 * 1024)

    def leg(threads, seconds):
        nodes, games, moves = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        took = ctypes.c_double()
        stats = (ctypes.c_uint64 * 9)()
        cpu = (ctypes.c_double * threads)()
        lib.ago_cpu_baseline_ex.restype = None
        lib.ago_cpu_baseline_ex(args.rules, args.board, args.board, ctypes.byref(cfg), threads, 1000, ctypes.c_double(seconds),
                                ctypes.byref(nodes), ctypes.byref(games), ctypes.byref(moves), ctypes.byref(took), stats, cpu)
        cpu_total = sum(cpu)
        return dict(threads=threads, simulations_per_sec=nodes.value / took.value, per_thread=nodes.value / took.value / threads, seconds=took.value,
                    simulations=nodes.value, moves=moves.value,
                    # user + system CPU seconds the worker threads consumed inside the timed region (CLOCK_THREAD_CPUTIME_ID): utilisation well below
                    # 1 = the threads did not get the CPUs they were started on (quota, oversubscription), not a property of the code
                    thread_cpu_seconds_mean=cpu_total / threads, thread_cpu_seconds_min=min(cpu), cpu_utilisation=cpu_total / (threads * took.value),
                    simulations_per_cpu_second=nodes.value / max(cpu_total, 1e-9))

    # one self-play game per thread, every game with its own 64 MB solver table like the reference's Search: the best thread count is not
    # obvious (more threads than physical cores / memory channels lose throughput), so a short sweep picks it; the best leg is the value
    if args.cpu_threads > 0:
        counts = [min(usable, args.cpu_threads)]
    else:
        counts = sorted({c for c in (usable // 8, usable // 4, usable // 2, usable) if c >= 1})
    probes = [leg(c, 0.1 * max_seconds) for c in counts] if len(counts) > 1 else []
    chosen = max(probes, key=lambda x: x["simulations_per_sec"])["threads"] if probes else counts[0]
    best = leg(chosen, max_seconds - 0.1 * max_seconds * len(probes))     # the reported value: one long leg at the best thread count
    legs = probes + [best]
    worst = min(legs, key=lambda x: x["per_thread"])
    return dict(value=best["simulations_per_sec"], unit="simulations/s", cores=best["threads"], kind="port",
                sample="%d threads x 1 self-play game each (same rules/board/playouts/batch, stand-in evaluator, NN cost excluded), %.1f s wall, %d simulations, %d moves; "
                       "thread count chosen by short probes over %s threads"
                       % (best["threads"], best["seconds"], best["simulations"], best["moves"], counts),
                host_cpus=cores, host_cpus_total=os.cpu_count(), cgroup_cpu_max=quota_text, cgroup_quota_cpus=quota_cpus, cpu_model=model,
                per_thread=best["per_thread"], cpu_utilisation=best["cpu_utilisation"], thread_cpu_seconds_mean=best["thread_cpu_seconds_mean"],
                simulations_per_cpu_second=best["simulations_per_cpu_second"], sweep=legs,
                # what the sweep's shape means is read off the legs themselves: per CPU-SECOND a thread does about the same work at every count when
                # the slowdown is lost CPU time (quota / more threads than CPUs granted: cpu_utilisation falls), and less work per CPU-second when
                # it is memory (every thread owns a 64 MB solver table — the reference's size — touched at random: n threads keep n x 64 MB hot)
                scaling_note="slowest leg: %d threads, %.0f simulations/s per thread at CPU utilisation %.2f, %.0f simulations per CPU-second (best leg: %.0f)"
                             % (worst["threads"], worst["per_thread"], worst["cpu_utilisation"], worst["simulations_per_cpu_second"], best["simulations_per_cpu_second"]),
                # the REAL reference search core (compiled with AVX2 intrinsics, one thread, fake evaluator) measured 10.5 k/s for this
                # shape in the survey container; the oracle is a scalar restatement (no SSE/AVX neighbourhood code)
                reference_core_per_thread_survey=10500.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 6000 steps = about two generations of whole games (a game lasts ~3000 pool steps): simulations/s and games/s of the steady state,
    # not of the opening phase only (150 steps of fresh games give 7 % more simulations/s and a seventh of the games/s)
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--games", type=int, default=1024, help="games per GPU")
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--board", type=int, default=15)
    ap.add_argument("--rules", type=int, default=0)
    ap.add_argument("--action-values", type=int, default=0, help="1: ResnetPVQ network (extra action-values head feeding the edge Q)")
    ap.add_argument("--table-entries", type=int, default=4 * 1024 * 1024)
    ap.add_argument("--yield-fraction", type=float, default=0.5,
                    help="straggler cut-off of the search launch: once this fraction of its games is done, a game whose batch still needs a serial re-run\n"
                         "(speculative solver) or another serial solve sits this step out (0 = never).  Pacing only, the games are the same; measured on\n"
                         "one box (profiles/r05_sweep_yield_fraction.txt, 16 solver waves per compute unit): 0.3 956 k, 0.4 964 k, 0.5 964 k, 0.6 935 k, 0.7 906 k, 0.8 902 k")
    ap.add_argument("--slices", type=int, default=4,
                    help="the pool stepped as this many slices on streams that own disjoint blocks of the chip's compute units (1 = one lock-step pool)")
    ap.add_argument("--speculative", type=int, default=1,
                    help="1: select + threat solver as one persistent launch with the leaves of a batch solved in parallel (AgxEngineConfig.speculative_solver)")
    ap.add_argument("--speculative-waves", type=int, default=0, help="waves of that launch over the whole pool, 0 = as many as stay resident (16 per compute unit on 15x15 boards, 10 on 20x20)")
    ap.add_argument("--config", default="", help="BASELINE.json preset: C2 (the default), C3 (standard, 10x128, 800 playouts: with --gpus 8 = configs[2]), "
                                                 "C4 (caro5 20x20, 10x128), C5 (renju, 10x128, 1600 playouts)")
    ap.add_argument("--age-steps", type=int, default=-1,
                    help="untimed pool steps before the timed region so that a short run measures the steady state (trees of ~2 k nodes, slices out of "
                         "phase) instead of the opening phase; -1 = 3000 when --steps < 3000, else 0")
    ap.add_argument("--policy-gain", type=float, default=1.0,
                    help="scale of the synthetic network's policy logits: 1.0 = plain He-init (near-flat priors, trees ~6x wider than self-play with a "
                         "trained network), 2.5 = peaked priors pruning to ~30 edges per node like a trained network's (a second, realistic tree shape)")
    ap.add_argument("--network-cus", type=int, default=0,
                    help="> 0: the chip as two partitions shared by all slices — this many compute units run every slice's network launches, the rest "
                         "every slice's search launches (streams ordered by events); 0: every slice owns 1 / slices of the chip for all its stages")
    ap.add_argument("--tree-cus", type=int, default=0, help="with --network-cus: compute units set aside for the expand / advance launches")
    ap.add_argument("--tree-on-network", type=int, default=0,
                    help="with --network-cus: 1 = the expand / advance launches follow the tower on the network partition's stream (the search partition then "
                         "holds nothing but persistent search launches: the next slice's waves move in as the previous launch's leave)")
    ap.add_argument("--share-cus", type=int, default=1,
                    help="the co-resident pairing experiment: this many slices share one block of compute units (e.g. --slices 8 --share-cus 2 = four "
                         "blocks of 64 CUs with two half-slices each, whose search and network launches may overlap on the same units; needs a tower "
                         "that leaves room: a library built with -DAGX_NN_WAVES_PER_EU=3 and AGX_NN_SINGLE_PLANE=1)")
    ap.add_argument("--stagger", type=int, default=1,
                    help="1: the slices enter the timed region a stage apart in their cycles, as a pool that has run for a while has them (the synchronisation in "
                         "front of the timed region would otherwise start all of them on the same stage); 0: all slices start with the search launch")
    ap.add_argument("--host-pacing", type=int, default=2,
                    help="N > 0: the host stays at most N steps ahead of every slice's stream and SLEEPS on a blocking event; 0: it enqueues until the "
                         "launch queue is full and spins there (a whole CPU more per rank: the line's ranks[].host_cpu_utilisation)")
    ap.add_argument("--plan-only", action="store_true",
                    help="print every rank's plan (device index, opening seeds, device memory the pool would allocate) as one JSON line and exit: touches no GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline, 0 = every host CPU")
    args = ap.parse_args()
    presets = {"": {}, "C2": {}, "C3": dict(rules=1, blocks=10, sims=800), "C4": dict(rules=3, board=20, blocks=10), "C5": dict(rules=2, blocks=10, sims=1600)}
    if args.config.upper() not in presets:
        raise SystemExit("bench.py: unknown --config %s (C2, C3, C4, C5)" % args.config)
    for key, value in presets[args.config.upper()].items():
        setattr(args, key, value)
    if args.age_steps < 0:
        args.age_steps = 3000 if args.steps < 3000 else 0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (nothing has touched a GPU yet: no library, no torch.cuda)
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    from alphagomoku_amd import distributed
    rank, local_rank, world = distributed.env_ranks()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (launch with --nproc-per-node %d, or plainly and let bench.py spawn the ranks)"
                         % (args.gpus, world, args.gpus))
    # the ranks only exchange a barrier and two tiny reductions of host scalars -> gloo on CPU tensors keeps torch off the GPUs
    # (set AGX_DIST_BACKEND=nccl to run the same reductions over RCCL)
    dist = distributed.init(backend=os.environ.get("AGX_DIST_BACKEND", "gloo"))

    from alphagomoku_amd import build
    if rank == 0:
        build.build(verbose=False)
    if dist is not None:
        dist.barrier()
    from alphagomoku_amd import lib, check, synthetic, selfplay, _lib
    from alphagomoku_amd.networks import AGNetwork
    build_hash = _lib.require_current_build()   # refuses a stale prebuilt libagx.so (its compiled-in hash != the sources beside it)

    device_index = int(os.environ.get("AGX_FORCE_DEVICE", local_rank))  # AGX_FORCE_DEVICE: test the N > 1 flow on a 1-GPU box
    # tree arenas: every game starts with class-0 regions (8 nodes / 192 edges per playout of the budget) in pool-wide heaps and moves into larger
    # ones on demand (AgxEngineConfig.arena_reserve: the heaps hold 4 x the class-0 total); the measured peaks are on the line
    node_capacity = max(4096, 8 * args.sims)
    edge_capacity = max(65536, 192 * args.sims)
    cfg = selfplay.default_config(rules=args.rules, board_size=args.board, n_games=args.games, max_batch_size=args.batch,
                                  max_simulations=args.sims, tss_table_entries=args.table_entries, solver_yield_fraction=args.yield_fraction,
                                  action_values=args.action_values, node_capacity=node_capacity, edge_capacity=edge_capacity, arena_reserve=3.0,
                                  record_format=2, speculative_solver=args.speculative, speculative_waves=args.speculative_waves)
    n_openings = args.games * 3  # enough openings for every game that can finish during the run; seeds are disjoint across ranks
    if args.plan_only:
        # What each rank WOULD do, from its environment alone (no device call: the library's sizing pass adds up agx_engine_create's allocations):
        # one pool per device like the reference's one generator thread per device (GeneratorManager.cpp:146-152).  No scaling curve is measured here.
        need = ctypes.c_ulonglong()
        check(lib.agx_engine_estimate_device_bytes(ctypes.byref(cfg), 256, ctypes.byref(need)))
        seed0 = distributed.rank_seed_base(rank)
        mine = [rank, local_rank, device_index, seed0, seed0 + n_openings - 1, need.value, len(os.sched_getaffinity(0))]
        everyone = distributed.gather(dist, mine)
        if rank == 0:
            print(json.dumps({"plan": [dict(rank=int(r[0]), local_rank=int(r[1]), device_index=int(r[2]), opening_seeds=[int(r[3]), int(r[4])],
                                            device_bytes=int(r[5]), host_cpus=int(r[6])) for r in everyone],
                              "n_gpus": world, "hbm_bytes_per_device": 288 * 10 ** 9, "scaling": "weak",
                              "note": "plan only: nothing was run, no scaling curve measured"}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    # a rank's host side is one launch loop: keep it on the CPUs next to its GPU (only with several ranks: a single rank keeps the box's affinity)
    numa_node, pinned_cpus = distributed.pin_to_gpu_numa_node(device_index) if world > 1 and not os.environ.get("AGX_NO_PIN") else (None, 0)
    check(lib.agx_set_device(device_index))
    desc = synthetic.net_desc(rows=args.board, cols=args.board, blocks=args.blocks, filters=args.filters, action_values=args.action_values)
    blob, _ = synthetic.make_weights(desc, policy_gain=args.policy_gain)
    net = AGNetwork(desc)
    net.loadWeights(blob)
    pool = selfplay.GeneratorPool(cfg)
    openings = synthetic.make_openings(args.board, n_openings, seed0=distributed.rank_seed_base(rank), rules=args.rules)
    pool.begin(selfplay.pack_openings(openings))
    check(lib.agx_device_synchronize())

    def make_timers(n):
        out = []
        for _ in range(n):
            t = ctypes.c_void_p()
            check(lib.agx_timer_create(ctypes.byref(t)))
            out.append(t)
        return out

    # The pool as `slices` groups of games, each stepped on its own stream that owns 1 / slices of the compute units (CU mask): the slices
    # drift out of phase, so the power-limited tower launches never cover the whole chip at once (they hold a higher clock) and no slice
    # waits for another slice's stragglers.  Same games, same results (games never interact); 1 slice = the whole pool in lock step.
    slices = max(1, min(args.slices, 16))
    while slices > 1 and (args.games % slices != 0 or args.games // slices < 4):
        slices //= 2
    streams, cus_per_slice, total_cus = [None], None, None
    net_streams, tree_streams, events = None, None, None
    if slices > 1 and args.network_cus > 0:
        streams, net_streams, tree_streams, search_cus, cus_per_slice = selfplay.chip_partitions(slices, args.network_cus, args.tree_cus)
        total_cus = search_cus + cus_per_slice + args.tree_cus
        check(lib.agx_net_set_launch_width(net._net, cus_per_slice))
        events = []
        for _ in range(3 * slices):
            ev = ctypes.c_void_p()
            check(lib.agx_event_create(ctypes.byref(ev)))
            events.append(ev)
    elif slices > 1:
        try:
            if args.share_cus > 1:
                streams, cus_per_slice = selfplay.shared_chip_slices(slices, args.share_cus)
                total_cus = cus_per_slice * slices // args.share_cus
            else:
                streams, cus_per_slice = selfplay.chip_slices(slices)
                total_cus = cus_per_slice * slices
            check(lib.agx_net_set_launch_width(net._net, cus_per_slice))
        except Exception as exc:   # no CU-mask support: one lock-step pool
            print("bench.py: chip slices unavailable (%s), running one pool" % exc, file=sys.stderr)
            slices, streams = 1, [None]

    # Where in its cycle (0 search launch, 1 network, 2 expand / advance) a slice's step BEGINS.  A pool that has run for a while has its slices out
    # of phase by itself (nothing couples them), but the device-wide synchronisation in front of the timed region lines all of them up at the
    # search launch — four towers then run at once for the first cycles, which is exactly what the slicing is there to avoid, and a 20-step
    # window is over before they have drifted apart again.  With --stagger (default) slice g enters the timed region `phase[g]` stages into
    # its cycle (the stages before that are enqueued ahead of the synchronisation, untimed); inside the timed region every slice still runs
    # EXACTLY `--steps` full cycles — select + solve, network, expand + backup + advance — only rotated.
    phase = [(0, 1, 2, 1)[g % 4] if (args.stagger and slices > 1 and net_streams is None) else 0 for g in range(slices)]

    def step_slice(g, nn_timer=None, first_stage=0):
        for stage in ((0, 1, 2), (1, 2, 0), (2, 0, 1))[first_stage]:
            run_stage(g, stage, nn_timer)
        if pacing is not None:
            pacing(g)

    def run_stage(g, stage, nn_timer):
        if net_streams is None:   # (the default: a slice's three stages on its one stream)
            if stage == 0:
                pool.select_solve_group(g, slices, streams[g])
            elif stage == 1:
                if nn_timer is not None:
                    check(lib.agx_timer_start(nn_timer, streams[g]))
                pool.evaluate_group(net, g, slices, streams[g])
                if nn_timer is not None:
                    check(lib.agx_timer_stop(nn_timer, streams[g]))
            else:
                pool.expand_backup_group(g, slices, streams[g])
            return
        if stage != 0:
            return   # (chip partitions: the three stages are enqueued together, with their events, by stage 0)
        pool.select_solve_group(g, slices, streams[g])
        ns = streams[g]
        if net_streams is not None:   # the tower runs on the network partition: its stream waits for the search launch, and the search stream for it
            ns = net_streams[g]
            check(lib.agx_event_record(events[3 * g], streams[g]))
            check(lib.agx_stream_wait_event(ns, events[3 * g]))
        if nn_timer is not None:
            check(lib.agx_timer_start(nn_timer, ns))
        pool.evaluate_group(net, g, slices, ns)
        if nn_timer is not None:
            check(lib.agx_timer_stop(nn_timer, ns))
        ts = streams[g]
        if net_streams is not None and args.tree_on_network:
            ts = ns   # (behind the tower on its own stream: no event in between)
        elif net_streams is not None:
            ts = tree_streams[g] if tree_streams is not None else streams[g]
            check(lib.agx_event_record(events[3 * g + 1], ns))
            check(lib.agx_stream_wait_event(ts, events[3 * g + 1]))
        pool.expand_backup_group(g, slices, ts)
        if ts is not streams[g]:   # the slice's next search launch waits for its tree launches
            check(lib.agx_event_record(events[3 * g + 2], ts))
            check(lib.agx_stream_wait_event(streams[g], events[3 * g + 2]))

    def keep_going(i):
        # what a generator thread does every few hundred steps — hand the finished samples over (GeneratorManager.cpp:160-164) and keep the
        # opening list ahead of the games
        nonlocal n_openings
        if (i + 1) % 256 == 0:
            pool.fetch_records(drain=True)   # format-201 samples (6 bytes per visited cell) + finished games
            if pool.stats()["openings_taken"] + args.games > n_openings:
                extra = synthetic.make_openings(args.board, args.games, seed0=distributed.rank_seed_base(rank) + n_openings, rules=args.rules)
                pool.add_openings(selfplay.pack_openings(extra))
                n_openings += args.games

    pacing = None
    if args.host_pacing > 0:   # (agx.h: agx_event_create_blocking; the same pacing as agx.hpp's HostPacer and ag::GameGenerator::generate)
        ring = args.host_pacing + 1
        pace_events = []
        for g in range(slices):
            row = []
            for _ in range(ring):
                ev = ctypes.c_void_p()
                check(lib.agx_event_create_blocking(ctypes.byref(ev)))
                row.append(ev)
            pace_events.append(row)
        pace_count = [0] * slices

        def pacing(g):
            k = pace_count[g]
            check(lib.agx_event_record(pace_events[g][k % ring], streams[g]))
            pace_count[g] = k + 1
            if k >= args.host_pacing:
                check(lib.agx_event_synchronize(pace_events[g][(k - args.host_pacing) % ring]))

    # untimed: warm-up launches, then the pool is AGED so that a short timed region sees what a long run sees — games in every phase, trees of a
    # couple of thousand nodes, arenas that have grown, slices out of phase (fresh games alone are the opening phase: small trees, no game ends)
    for i in range(args.warmup + args.age_steps):
        for g in range(slices):
            step_slice(g)
        keep_going(i)
    for g in range(slices):   # the slices' phase offsets: the stages a rotated step does not begin with (see `phase`)
        for stage in range(phase[g]):
            run_stage(g, stage, None)
    check(lib.agx_device_synchronize())
    s0 = pool.stats()
    pool.kernel_timing(True)   # HIP events around every engine kernel, on the launch stream
    t_nn = [make_timers(args.steps) for _ in range(slices)]   # the network stage (tower + value head) of every slice and step

    if dist is not None:
        dist.barrier()
    check(lib.agx_device_synchronize())
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    threads0 = thread_cpu_seconds()
    for i in range(args.steps):
        for g in range(slices):
            step_slice(g, t_nn[g][i], phase[g])
        keep_going(i)
    check(lib.agx_device_synchronize())
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    host_cpu_seconds = time.process_time() - cpu0   # user + system time of this rank's process inside the timed region (all its threads)
    threads1 = thread_cpu_seconds()
    # which threads that was: (name, CPU seconds) of the busiest ones — the launch loop is this process's main thread, the rest belong to the HIP / HSA runtime
    host_threads = sorted(((name, round(sec - threads0.get(tid, (name, 0.0))[1], 2)) for tid, (name, sec) in threads1.items()), key=lambda x: -x[1])[:4]
    s1 = pool.stats()
    kernel_ms, kernel_launches = pool.kernel_timing(False)

    def total_ms(timers):
        acc = 0.0
        ms = ctypes.c_float()
        for t in timers:
            check(lib.agx_timer_elapsed_ms(t, ctypes.byref(ms)))
            acc += ms.value
        return acc

    ms_nn = sum(total_ms(t) for t in t_nn)                 # summed over slices (their launches overlap in time)
    launches = args.steps * slices                          # launches of every stage
    ms_sel, ms_exp = kernel_ms[0] + kernel_ms[1], kernel_ms[2] + kernel_ms[3]
    sims = s1["evaluated_nodes"] - s0["evaluated_nodes"]
    evals = s1["network_evaluations"] - s0["network_evaluations"]
    moves = s1["moves_played"] - s0["moves_played"]
    games_done = s1["games_finished"] - s0["games_finished"]
    levels = s1["select_levels"] - s0["select_levels"]
    edge_reads = s1["select_edge_reads"] - s0["select_edge_reads"]
    solver_nodes = s1["solver_nodes"] - s0["solver_nodes"]
    leaks = s1["information_leaks"] - s0["information_leaks"]
    local_elapsed = elapsed
    per_rank = distributed.gather(dist, [sims, local_elapsed, distributed.rank_seed_base(rank), int(os.environ.get("AGX_FORCE_DEVICE", local_rank)),
                                         s1["first_error"], host_cpu_seconds, -1 if numa_node is None else numa_node, pinned_cpus,
                                         pool.device_bytes(), len(threads1)])
    elapsed, (sims, evals, moves, games_done) = distributed.combine(dist, elapsed, [sims, evals, moves, games_done])
    # an engine error on ANY rank fails every rank, after the collectives (a rank that raised before them would leave the others waiting)
    failed = [(i, int(r[4])) for i, r in enumerate(per_rank) if int(r[4]) != 0]
    if failed:
        if dist is not None:
            dist.destroy_process_group()
        raise RuntimeError("device engine stopped: %s" % ", ".join("rank %d error code %d" % f for f in failed))

    if rank == 0:
        flops = nn_flops_per_position(desc)
        # dominant kernel: the policy/value tower (one launch per step); algorithmic FLOPs per launch = positions x FLOPs/position
        local_evals = s1["network_evaluations"] - s0["network_evaluations"]
        # (ms_nn is the sum of all network launches' durations: with slices this is the rate of ONE launch on its slice's compute units)
        nn_tflops = (local_evals * flops) / (ms_nn * 1e-3) / 1e12 if ms_nn > 0 else 0.0
        chip_share = (cus_per_slice / float(total_cus)) if slices > 1 else 1.0
        nn_peak = 2500.0 * chip_share
        depth = levels / max(1, (s1["evaluated_nodes"] - s0["evaluated_nodes"]) + leaks)
        edges_per_level = edge_reads / max(1, levels)
        RULE_NAMES = ["freestyle", "standard", "renju", "caro5", "caro6"]
        # HBM-side roof of the tree kernels (k_select + k_expand + k_advance): algorithmic bytes per simulation per SURVEY 8(d)
        # with the MEASURED mean select depth d and edges per level E (new-leaf edge count taken as E):
        #   select d*(40 + 24E) + virtual loss 4d + hash probe d*(8 + 104) + expand 176 + 24E + backup 128d + 2dE + encode 10*HW
        hw = args.board * args.board
        fused = os.environ.get("AGX_FUSE_SELECT", "1") != "0"
        select_bytes = depth * (40 + 24 * edges_per_level) + 4 * depth + depth * 112 + 10 * hw   # descents + the feature planes (written by the solve stage)
        update_bytes = 176 + 24 * edges_per_level + 128 * depth + 2 * depth * edges_per_level
        # with select fused into the solver launch its time cannot be separated from the solver's: the HBM-side figure then covers the
        # launches that are pure tree work (expand / backup / advance) with their own bytes
        tree_bytes = update_bytes if fused else select_bytes + update_bytes
        tree_ms = (0.0 if fused else kernel_ms[0]) + kernel_ms[2] + kernel_ms[3]
        local_sims = s1["evaluated_nodes"] - s0["evaluated_nodes"]
        tree_gbs = local_sims * tree_bytes / (tree_ms * 1e-3) / 1e9 * slices if tree_ms > 0 else 0.0   # per launch x slices (a slice's launches own their CUs)
        # PMC-derived figures cannot be sampled from inside this process (rocprofv3 --pmc passes, scripts/pmc_summary.py).  They are quoted only
        # when the committed summary was taken from THIS build (same source hash) and this workload; otherwise null.
        traffic = None
        mfma_busy = None
        solver_issue = None
        nn_clock = None
        pmc_path = os.path.join(ROOT, "profiles", "r06_pmc_summary.json")
        src_hash = source_hash()
        pmc_build = None
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            pmc_build = pmc.get("source_hash")
            if (pmc_build == src_hash and pmc.get("slices", 1) == slices and args.games == 1024 and args.filters == 128 and args.blocks == 6
                    and args.board == 15 and args.rules == 0):
                traffic = pmc.get("nn_tower_bytes_per_launch_corrected")
                mfma_busy = pmc.get("nn_tower_mfma_busy_fraction")
                solver_issue = pmc.get("k_solve_issue_busy_fraction")
                nn_clock = pmc.get("nn_tower_shader_clock_mhz")
        cu_total = ctypes.c_int()
        check(lib.agx_device_cu_count(ctypes.byref(cu_total)))
        spec_waves_per_launch = max(1, pool.speculative_waves() // slices)
        gpu_ms = kernel_ms[0] + kernel_ms[1] + kernel_ms[2] + kernel_ms[3] + ms_nn
        search_kernel = "k_search_spec" if args.speculative else "k_solve"   # the launch timed as the solve stage
        per_kernel = {"k_select": kernel_ms[0], search_kernel: kernel_ms[1], "nn_tower": ms_nn, "k_expand": kernel_ms[2], "k_advance": kernel_ms[3]}
        longest = max(per_kernel, key=per_kernel.get)
        result = {
            "metric": "MCTS simulations/sec (self-play, %dx%d %s)" % (args.board, args.board, RULE_NAMES[args.rules]),
            "value": sims / elapsed,
            "unit": "simulations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "aged_steps": args.age_steps,   # untimed pool steps before the timed region (steady state: see peak_tree_per_game)
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16 (network, fp32 accumulate) + int/fp32 (tree)",
            "data": "synthetic (random openings, He-init weights seed 1234%s)" % ("" if args.policy_gain == 1.0 else ", policy logits x %g: trained-like peaked priors" % args.policy_gain),
            "config": {"workload": "%s %dx%d, %d-block/%d-filter net%s, %d playouts/move, %d parallel self-play games per GPU, max_batch_size %d"
                                   % (RULE_NAMES[args.rules], args.board, args.board, args.blocks, args.filters, " (pvq)" if args.action_values else "", args.sims,
                                      args.games, args.batch),
                       "games_per_gpu": args.games, "parallelism": "independent game pools x%d (no collective)" % world},
            "moves_per_sec": moves / elapsed,
            "games_per_sec": games_done / elapsed,
            "nn_positions_per_sec": evals / elapsed,
            # how the pool is stepped: `count` slices of games_per_gpu / count games, each on a stream that owns cus_per_slice compute units;
            # the slices' launches overlap in time, so the per-launch durations below add up to more than ms_per_step
            "slices": {"count": slices, "cus_per_slice": cus_per_slice, "slices_per_cu_block": args.share_cus, "games_per_slice": args.games // slices, "first_stage_of_a_step": phase, "host_steps_ahead": args.host_pacing if args.host_pacing > 0 else None,
                       "partitions": ({"network_cus": cus_per_slice, "search_cus": total_cus - cus_per_slice - args.tree_cus, "tree_cus": args.tree_cus}
                                      if net_streams is not None else None)},
            "stage_ms_per_step": {"select_solve": ms_sel / launches, "network": ms_nn / launches, "expand_backup_advance": ms_exp / launches,
                                  "of": "one slice's launches (average)"},
            "kernel_ms_per_step": {"k_select": kernel_ms[0] / launches, search_kernel: kernel_ms[1] / launches, "nn_tower": ms_nn / launches,
                                   "k_expand": kernel_ms[2] / launches, "k_advance": kernel_ms[3] / launches},
            # default: Search::select runs inside the solver's launch (one wave per game selects, then solves: k_solve<.., FUSED>), its time is
            # part of k_solve and k_select is 0; AGX_FUSE_SELECT=0 launches them separately
            "select_fused_into_solve": os.environ.get("AGX_FUSE_SELECT", "1") != "0",
            "speculative_solver": {"enabled": bool(args.speculative), "leaves_solved": int(s1["speculative_solves"] - s0["speculative_solves"]),
                                   "rerun_serially": int(s1["speculative_reruns"] - s0["speculative_reruns"]),
                                   "batches_deferred": int(s1["speculative_deferrals"] - s0["speculative_deferrals"]),
                                   "solves_parked": int(s1["speculative_parks"] - s0["speculative_parks"])},
            "peak_tree_per_game": {"nodes": int(s1["peak_nodes"]), "edges": int(s1["peak_edges"]), "class0_node_capacity": node_capacity,
                                   "class0_edge_capacity": edge_capacity, "arena_grows": int(s1["arena_grows"]), "arena_releases": int(s1["arena_releases"]),
                                   "arena_failures": int(s1["arena_failures"]), "arena_max_class": int(s1["arena_max_class"]),
                                   "arena_heap_high_water": float(s1["arena_heap_used"])},
            "shape": {"mean_select_depth": depth, "mean_edges_per_level": edges_per_level,
                      "solver_nodes_per_simulation": solver_nodes / max(1, s1["evaluated_nodes"] - s0["evaluated_nodes"]),
                      "nn_evals_per_simulation": local_evals / max(1, s1["evaluated_nodes"] - s0["evaluated_nodes"])},
            # per rank: its own work and wall time, the host CPU it needed (CPU seconds of the rank's process / its wall seconds — one
            # launch loop per GPU must stay well below one core) and where it was pinned
            "ranks": [{"rank": i, "device": int(r[3]), "simulations": int(r[0]), "seconds": r[1], "opening_seed_base": int(r[2]),
                       "host_cpu_seconds": r[5], "host_cpu_utilisation": r[5] / max(r[1], 1e-9), "numa_node": int(r[6]), "pinned_cpus": int(r[7]),
                       # what the rank takes of its GPU's 288 GB (trees + 16 B x table entries x games of solver tables + spill areas + exchange
                       # buffers + record pools; the network's weights are a few MB on top) and how many host threads its process runs
                       # (the launch loop + the HIP / HSA runtime's) — one such rank per GPU; NO multi-GPU scaling curve has been measured
                       "device_bytes_allocated": int(r[8]), "host_threads": int(r[9]),
                       **({"host_threads_cpu_seconds": host_threads} if i == 0 else {})}
                      for i, r in enumerate(per_rank)],
            "longest_kernel": {"name": longest, "share_of_kernel_time": per_kernel[longest] / gpu_ms if gpu_ms > 0 else None},
            "source_hash": src_hash, "library_build_hash": build_hash,
            # the roofline object is the MFMA-bound network kernel (the only kernel of the step with a compute roof); the threat solver
            # (roofline_solver) has neither an HBM nor an MFMA roof, see DESIGN.md
            "roofline": {"bound": "mfma", "kernel": "nn_tower_kernel<%d,%d,%d>" % (args.filters, args.board, args.board),
                         # a launch runs on its slice's compute units only (CU-masked stream): achieved and peak are per launch, i.e. per
                         # cus_per_slice / total of the chip; `count` such launches overlap
                         "achieved": nn_tflops, "peak": nn_peak, "unit": "TFLOP/s", "frac": nn_tflops / nn_peak, "traffic": traffic,
                         "chip_share_of_a_launch": chip_share, "whole_chip_equivalent": nn_tflops / chip_share,
                         "peak_note": ("peak = 2500 TFLOP/s (dense fp16 MFMA, whole chip) x %d/%d compute units: the launch runs on a CU-masked stream, "
                                       "%d such launches (one per slice) overlap; achieved x %d = whole_chip_equivalent against 2500"
                                       % (cus_per_slice, total_cus, slices, slices)) if slices > 1 else "whole chip: 2500 TFLOP/s dense fp16 MFMA",
                         "flops_per_position": flops, "positions_per_launch": local_evals / launches,
                         "avg_launch_ms": ms_nn / launches,
                         # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of the committed PMC passes of this command
                         "mfma_busy_fraction_pmc": mfma_busy,
                         # GRBM_GUI_ACTIVE / 8 / launch duration: the tower runs power-limited below the 2.4 GHz the 2.5 PFLOP/s peak assumes
                         # (frac keeps the nominal peak; this is the same rate against the MFMA issue rate at the clock actually held)
                         # the clock of a launch of the (serialising) PMC passes, i.e. of a tower alone on an idle chip — not of the overlapped run
                         "shader_clock_mhz_serialised_pmc_pass": nn_clock,
                         # what the matrix cores deliver over the whole step, all slices: evaluated positions/s x FLOPs / 2.5 PFLOP/s (the tower
                         # only runs part of each slice's cycle; the search kernels use no MFMA)
                         "time_averaged_whole_chip_frac": (evals / elapsed) * flops / 2.5e15 / world,
                         "pmc_summary_build": pmc_build},
            "roofline_solver": {"bound": ("none (instruction issue of the solver waves: %.1f per SIMD, the leaves of a batch in parallel)" % (pool.speculative_waves() / (4.0 * max(1, cu_total.value))) if args.speculative
                                          else "none (instruction issue / dependent-chain latency: one wave per game, tasks of a game strictly ordered)"),
                                "kernel": ("k_search_spec (select + speculative threat solver + commit, one persistent launch)" if args.speculative else
                                           ("k_solve (select + threat solver of a game in one wave)" if os.environ.get("AGX_FUSE_SELECT", "1") != "0" else "k_solve")),
                                "ms_per_step": kernel_ms[1] / launches,
                                "share_of_kernel_time": kernel_ms[1] / gpu_ms if gpu_ms > 0 else None,
                                "solver_nodes_per_sec": solver_nodes / elapsed,
                                # wave-time per solver node: launch time x the waves of the launch / nodes (an upper bound on the work per node:
                                # it counts the waves' idle tail of the launch as well)
                                "us_per_solver_node_per_wave": (kernel_ms[1] * 1e3 * (spec_waves_per_launch if args.speculative else args.games // slices) / solver_nodes)
                                if solver_nodes else None,
                                "waves_per_launch": spec_waves_per_launch if args.speculative else args.games // slices,
                                "issue_busy_fraction_pmc": solver_issue},
            # second roof ("two kernels, two roofs"): the tree kernels are gathers/scans over the flat node/edge arrays
            "roofline_tree": {"bound": "hbm", "kernels": "k_expand + k_advance" if fused else "k_select + k_expand + k_advance", "achieved": tree_gbs, "peak": 8000.0, "unit": "GB/s",
                              "frac": tree_gbs / 8000.0, "bytes_per_simulation": tree_bytes, "ms_per_step": tree_ms / launches,
                              "note": "latency-bound by one wave per game, not by bandwidth; k_solve (threat solver) has no HBM/MFMA roof"},
        }
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            result["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
