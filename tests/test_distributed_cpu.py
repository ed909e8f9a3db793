"""The N > 1 path of bench.py on CPU: two processes, gloo backend — rendezvous, per-rank seeds, MAX of the wall time and SUM of
the work counters (the only things ranks exchange: there is no collective on the data path)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    from alphagomoku_amd import distributed
    rank, local_rank, world = distributed.env_ranks()
    dist = distributed.init(backend="gloo")
    dist.barrier()
    elapsed, counters = distributed.combine(dist, 1.0 + rank, [100 * (rank + 1), 7, rank])
    everyone = distributed.gather(dist, [rank, 10 * rank])
    dist.barrier()
    print(json.dumps(dict(rank=rank, world=world, elapsed=elapsed, counters=counters, seed=distributed.rank_seed_base(rank), everyone=everyone)))
    dist.destroy_process_group()
''')


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_gloo(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err
        outs.append(json.loads(out.strip().splitlines()[-1]))
    for o in outs:
        assert o["world"] == 2
        assert o["elapsed"] == 2.0                       # MAX over ranks
        assert o["counters"] == [300.0, 14.0, 1.0]      # SUM over ranks
        assert o["everyone"] == [[0.0, 0.0], [1.0, 10.0]]  # per-rank values, by rank
    assert outs[0]["seed"] != outs[1]["seed"]


def test_single_process_is_identity():
    from alphagomoku_amd import distributed
    assert distributed.combine(None, 1.5, [3, 4]) == (1.5, [3.0, 4.0])


def test_bench_refuses_a_world_that_does_not_match_gpus():
    """`bench.py --gpus N` under a launcher with another world size must fail loudly (before anything touches a GPU) instead of silently
    running one pool and reporting n_gpus: 1"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE is 1" in (p.stderr + p.stdout)


def test_bench_spawns_its_ranks(tmp_path):
    """plain `python bench.py --gpus 2` becomes the launcher: the ranks it starts see WORLD_SIZE=2 (checked with a stand-in for
    torch.distributed.run's target: the spawned command line is bench.py itself, so here only the command construction is exercised by
    making the child fail fast on a missing GPU library call — the rank processes must exist and report their world size)"""
    probe = tmp_path / "sitecustomize.py"
    probe.write_text("import os, sys\n"
                     "if os.environ.get('WORLD_SIZE') and os.environ.get('AGX_PROBE_DIR'):\n"
                     "    open(os.path.join(os.environ['AGX_PROBE_DIR'], 'rank%s_of_%s' % (os.environ['RANK'], os.environ['WORLD_SIZE'])), 'w').close()\n"
                     "    if 'bench.py' in ' '.join(sys.argv): os._exit(0)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["PYTHONPATH"] = str(tmp_path) + os.pathsep + env.get("PYTHONPATH", "")
    env["AGX_PROBE_DIR"] = str(tmp_path)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("rank")) == ["rank0_of_2", "rank1_of_2"]


def _fake_sysfs(root, gpus, cpulists):
    """gpus: [(kfd node, domain, bus, device, function, numa node)] in any order; cpulists: {numa node: "a-b,c"}; KFD node 0 is a CPU node"""
    nodes = root / "sys/class/kfd/kfd/topology/nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for kfd, dom, bus, dev, fn, numa in gpus:
        (nodes / str(kfd)).mkdir(parents=True)
        (nodes / str(kfd) / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\ndomain %d\nlocation_id %d\n" % (dom, (bus << 8) | (dev << 3) | fn))
        pci = root / "sys/bus/pci/devices" / ("%04x:%02x:%02x.%x" % (dom, bus, dev, fn))
        pci.mkdir(parents=True)
        (pci / "numa_node").write_text("%d\n" % numa)
        (pci / "vendor").write_text("0x1002\n")
        (pci / "class").write_text("0x120000\n")
    for numa, cpus in cpulists.items():
        d = root / ("sys/devices/system/node/node%d" % numa)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")


def test_numa_pinning_follows_the_runtime_device_order(tmp_path, monkeypatch):
    """distributed.pin_to_gpu_numa_node: device k is the k-th GPU in KFD topology order (NOT in PCI address order), HIP_VISIBLE_DEVICES renumbers,
    the CPUs are the node's cpulist within the current affinity; an unreadable topology pins nothing"""
    import os
    from alphagomoku_amd import distributed
    mine = sorted(os.sched_getaffinity(0))
    lo, hi = mine[0], mine[-1]
    # KFD order: node 1 = bus 0xc1 on NUMA 1, node 2 = bus 0x05 on NUMA 0 — the opposite of the PCI address order
    _fake_sysfs(tmp_path, [(1, 0, 0xC1, 0, 0, 1), (2, 0, 0x05, 0, 0, 0)], {0: "%d" % lo, 1: "%d" % hi})
    pinned = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: pinned.append(set(cpus)))
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    assert distributed.pin_to_gpu_numa_node(0, str(tmp_path)) == (1, 1) and pinned[-1] == {hi}
    assert distributed.pin_to_gpu_numa_node(1, str(tmp_path)) == (0, 1) and pinned[-1] == {lo}
    assert distributed.pin_to_gpu_numa_node(2, str(tmp_path)) == (None, 0)            # no such device
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")                                     # device 0 is now the topology's second GPU
    assert distributed.pin_to_gpu_numa_node(0, str(tmp_path)) == (0, 1) and pinned[-1] == {lo}
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert distributed.pin_to_gpu_numa_node(0, str(tmp_path / "nothing-here")) == (None, 0)
    assert len(pinned) == 3
    # the filters compose: ROCR selects from the topology, HIP / CUDA from what ROCR left
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1,0")
    assert distributed._visible_index(0) == 1 and distributed._visible_index(1) == 0 and distributed._visible_index(2) is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert distributed._visible_index(0) == 0                                          # ROCR's second entry = the topology's first GPU
    assert distributed.pin_to_gpu_numa_node(0, str(tmp_path)) == (1, 1) and pinned[-1] == {hi}
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "1")                                    # HIP honours it as well
    assert distributed._visible_index(0) == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")                                     # two different inner lists: not resolved, the rank stays unpinned
    assert distributed._visible_index(0) is None and distributed.pin_to_gpu_numa_node(0, str(tmp_path)) == (None, 0)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-abcdef")                            # UUID form
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES")
    assert distributed._visible_index(0) is None


def test_eight_ranks_plan_disjoint_devices_and_seeds():
    """`bench.py --gpus 8 --plan-only` (no GPU here, none touched): the launcher starts eight ranks, each derives its device index from LOCAL_RANK alone
    (no AGX_FORCE_DEVICE), a seed range no other rank shares, and the device memory its pool would allocate (the library's sizing pass over
    agx_engine_create's allocations).  One pool per device is the reference's split (GeneratorManager.cpp:146-152: one generator thread per
    device); a C3 rank must fit a 288-GB device with room for the network.  NO scaling curve is measured by this: it checks the plan only."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "AGX_FORCE_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plan-only", "--config", "C3"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
    plan = line["plan"]
    assert line["n_gpus"] == 8 and len(plan) == 8 and sorted(r["rank"] for r in plan) == list(range(8))
    assert sorted(r["device_index"] for r in plan) == list(range(8))          # eight distinct devices, local rank = device
    assert all(r["device_index"] == r["local_rank"] for r in plan)
    ranges = sorted(tuple(r["opening_seeds"]) for r in plan)
    assert all(a[1] < b[0] for a, b in zip(ranges, ranges[1:]))                # disjoint opening seeds: the ranks play different games
    # a C3 pool (1024 games: 4 Mi-entry solver tables = 64 GiB, tree heaps for 800 playouts per move, speculative overlays, spill areas) on one
    # 288-GB device, the 10x128 network beside it; tests/test_bench_gpu.py checks the sizing pass against what an engine really allocates
    assert all(64 * 2 ** 30 < r["device_bytes"] < 130e9 for r in plan), [r["device_bytes"] for r in plan]
    assert len(set(r["device_bytes"] for r in plan)) == 1
    assert all(r["device_bytes"] < 0.5 * line["hbm_bytes_per_device"] for r in plan)
    assert "no scaling curve" in line["note"]
