"""Host-side mirror of the reference's self-play driver objects for the device engine.

`GeneratorPool` plays the role of one GeneratorThread with its games_per_thread GameGenerators
(src/selfplay/GeneratorManager.cpp:124-141, src/selfplay/GameGenerator.cpp:46-121): it owns one device engine handle and
drives it with select_solve -> evaluate -> expand_backup steps.  All search arithmetic happens in libagx.so."""
import ctypes

import numpy as np

from ._lib import (lib, check, AgxEngineConfig, AgxEngineBuffers, AgxEngineStats, AgxGameInfo, AgxEdgeView, AgxMoveRecord, AgxGameEnd, AgxSavedGame,
                   AgxRecordCounts, AgxGameBufferStats)

OPENING_CAP = 32


def default_config(**overrides):
    cfg = AgxEngineConfig()
    check(lib.agx_engine_default_config(ctypes.byref(cfg)))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise KeyError(k)
        setattr(cfg, k, v)
    return cfg


class HostPacer:
    """Keeps a launch loop at most `ahead` steps in front of a stream: call step(stream) behind every step's launches.  Without it the host
    enqueues until the launch queue is full and spins there — a whole CPU per device thread; with it the thread sleeps on a blocking event
    (agx.h: agx_event_create_blocking; agx.hpp: HostPacer).  Pacing only."""

    def __init__(self, ahead=2):
        self.ahead = ahead
        self.count = 0
        self.events = []
        for _ in range(ahead + 1 if ahead > 0 else 0):
            ev = ctypes.c_void_p()
            check(lib.agx_event_create_blocking(ctypes.byref(ev)))
            self.events.append(ev)

    def step(self, stream=None):
        if self.ahead <= 0:
            return
        ring = len(self.events)
        check(lib.agx_event_record(self.events[self.count % ring], stream))
        if self.count >= self.ahead:
            check(lib.agx_event_synchronize(self.events[(self.count - self.ahead) % ring]))
        self.count += 1


def chip_slices(n_slices):
    """n_slices streams that own disjoint, equal blocks of the device's compute units (agx_stream_create_with_cu_mask), for stepping a pool as
    n_slices groups: returns (streams, compute units per slice).  The streams are never destroyed (agx.h)."""
    cus = ctypes.c_int()
    check(lib.agx_device_cu_count(ctypes.byref(cus)))
    per = cus.value // n_slices
    if per < 1:
        raise ValueError("%d slices on %d compute units" % (n_slices, cus.value))
    words = (cus.value + 31) // 32
    streams = []
    for k in range(n_slices):
        mask = [0] * words
        for c in range(k * per, (k + 1) * per):
            mask[c // 32] |= 1 << (c % 32)
        s = ctypes.c_void_p()
        check(lib.agx_stream_create_with_cu_mask(ctypes.byref(s), (ctypes.c_uint32 * words)(*mask), words))
        streams.append(s)
    return streams, per


def shared_chip_slices(n_slices, sharing):
    """n_slices streams in groups of `sharing`: the streams of a group own the SAME block of compute units (n_slices / sharing disjoint blocks),
    so the kernels of a group's slices may be co-resident on those units — the co-resident pairing experiment (bench.py --share-cus).  Returns
    (streams, compute units per block)."""
    if sharing < 1 or n_slices % sharing != 0:
        raise ValueError("%d slices cannot share compute-unit blocks %d by %d" % (n_slices, sharing, sharing))
    cus = ctypes.c_int()
    check(lib.agx_device_cu_count(ctypes.byref(cus)))
    blocks = n_slices // sharing
    per = cus.value // blocks
    return [cu_mask_stream((k // sharing) * per, per, instance=k % sharing) for k in range(n_slices)], per


def cu_mask_stream(first_cu, n_cus, instance=0):
    """a stream whose kernels run on compute units [first_cu, first_cu + n_cus); `instance` distinguishes several streams on one mask"""
    total = ctypes.c_int()
    check(lib.agx_device_cu_count(ctypes.byref(total)))
    if first_cu < 0 or n_cus < 1 or first_cu + n_cus > total.value:
        raise ValueError("compute units [%d, %d) of %d" % (first_cu, first_cu + n_cus, total.value))
    words = (total.value + 31) // 32
    mask = [0] * words
    for c in range(first_cu, first_cu + n_cus):
        mask[c // 32] |= 1 << (c % 32)
    s = ctypes.c_void_p()
    check(lib.agx_stream_create_with_cu_mask_instance(ctypes.byref(s), (ctypes.c_uint32 * words)(*mask), words, instance))
    return s


def chip_partitions(n_slices, network_cus, tree_cus=0):
    """The chip as partitions shared by all slices of a pool: the first `network_cus` compute units run every slice's network launches, the
    last `tree_cus` (if any) every slice's expand / advance launches (short kernels with large workgroups, which would otherwise queue behind
    the persistent search waves), the rest every slice's search launches — one stream per slice on each partition, ordered by events.
    Slices in different phases fill each other's gaps on the search partition, and the matrix cores of the network partition always have a
    slice to work for.  Returns (search_streams, network_streams, tree_streams or None, search_cus, network_cus)."""
    total = ctypes.c_int()
    check(lib.agx_device_cu_count(ctypes.byref(total)))
    search_cus = total.value - network_cus - tree_cus
    if network_cus <= 0 or tree_cus < 0 or search_cus <= 0:
        raise ValueError("partitions of %d + %d compute units on a device with %d" % (network_cus, tree_cus, total.value))
    network = [cu_mask_stream(0, network_cus, instance=k) for k in range(n_slices)]
    search = [cu_mask_stream(network_cus, search_cus, instance=k) for k in range(n_slices)]
    tree = [cu_mask_stream(network_cus + search_cus, tree_cus, instance=k) for k in range(n_slices)] if tree_cus > 0 else None
    return search, network, tree, search_cus, network_cus


def pack_openings(openings):
    """list of lists of Move::toShort -> uint16 [n][OPENING_CAP]"""
    out = np.zeros((len(openings), OPENING_CAP), dtype=np.uint16)
    for i, o in enumerate(openings):
        if len(o) >= OPENING_CAP:
            raise ValueError("opening too long")
        out[i, 0] = len(o)
        out[i, 1:1 + len(o)] = o
    return out


class GeneratorPool:
    def __init__(self, cfg):
        self.cfg = cfg
        self._h = ctypes.c_void_p()
        check(lib.agx_engine_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        self.buffers = AgxEngineBuffers()
        check(lib.agx_engine_buffers(self._h, ctypes.byref(self.buffers)))
        self.cells = self.buffers.cells
        self.slots = self.buffers.slots

    def begin(self, openings, stream=None):
        arr = np.ascontiguousarray(openings, dtype=np.uint16)
        assert arr.ndim == 2 and arr.shape[1] == OPENING_CAP
        check(lib.agx_engine_begin(self._h, arr.ctypes.data_as(ctypes.c_void_p), arr.shape[0], stream))

    def select_solve(self, stream=None):
        check(lib.agx_engine_select_solve(self._h, stream))

    def evaluate(self, net, stream=None):
        check(lib.agx_engine_evaluate(self._h, net._net, stream))

    def expand_backup(self, stream=None):
        check(lib.agx_engine_expand_backup(self._h, stream))

    def step(self, net, stream=None):
        check(lib.agx_engine_step(self._h, net._net, stream))

    def step_group(self, net, group, n_groups, stream=None):
        check(lib.agx_engine_step_group(self._h, net._net, group, n_groups, stream))

    def expand_only(self, stream=None):
        """Search::generateEdges + expand + backup without the engine's own move rule (a game driven from outside: set_board)"""
        check(lib.agx_engine_expand_group(self._h, 0, 1, stream))

    def set_board(self, game, board, sign_to_move, stream=None):
        """Search::cleanup + Tree::setBoard(board, signToMove) + Search::setBoard for one game (evaluation/Player.cpp:100-110)"""
        b = np.ascontiguousarray(board, dtype=np.uint8).reshape(-1)
        check(lib.agx_engine_set_board(self._h, game, b.ctypes.data_as(ctypes.c_void_p), int(sign_to_move), stream))

    def cancel_pending(self, stream=None):
        """Search::cleanup: leaves selected but not expanded give their virtual losses back, every task buffer is emptied"""
        check(lib.agx_engine_cancel_pending(self._h, stream))

    def set_max_simulations(self, n):
        check(lib.agx_engine_set_max_simulations(self._h, int(n)))

    def set_batch_size(self, n):
        """Search::setBatchSize: leaves the select stage takes per game from now on (1 .. max_batch_size)"""
        check(lib.agx_engine_set_batch_size(self._h, int(n)))

    def select_group(self, group, n_groups, stream=None):
        check(lib.agx_engine_select_group(self._h, group, n_groups, stream))

    def solve_timed_group(self, group, n_groups, max_nodes, seconds, stream=None):
        """Search::solve(endTime >= 0): node limit max_nodes, `seconds` of wall clock from the launch's start shared out over the leaves"""
        check(lib.agx_engine_solve_timed_group(self._h, group, n_groups, int(max_nodes), float(seconds), stream))

    def save_games(self):
        """the games in flight (GeneratorThread::saveGames): list of dicts with game_slot, game_index, opening_id, sign_to_move, nn_queued, moves"""
        n = ctypes.c_int()
        check(lib.agx_engine_save_games(self._h, None, 0, ctypes.byref(n)))
        arr = (AgxSavedGame * max(n.value, 1))()
        check(lib.agx_engine_save_games(self._h, ctypes.cast(arr, ctypes.c_void_p), max(n.value, 1), ctypes.byref(n)))
        return [dict(game_slot=g.game_slot, game_index=g.game_index, opening_id=g.opening_id, sign_to_move=g.sign_to_move, nn_queued=g.nn_queued,
                     moves=[int(m) for m in g.moves[:g.n_moves]]) for g in arr[:n.value]]

    def restore_game(self, saved, slot=None, stream=None):
        """GameGenerator::load for one pool slot: the saved game continues there with an empty tree and solver table"""
        g = AgxSavedGame()
        g.game_slot = saved["game_slot"] if slot is None else slot
        g.game_index, g.opening_id, g.sign_to_move, g.nn_queued = saved["game_index"], saved["opening_id"], saved["sign_to_move"], saved["nn_queued"]
        g.n_moves = len(saved["moves"])
        for i, m in enumerate(saved["moves"]):
            g.moves[i] = m
        check(lib.agx_engine_restore_game(self._h, ctypes.byref(g), stream))

    def select_solve_group(self, group, n_groups, stream=None):
        check(lib.agx_engine_select_solve_group(self._h, group, n_groups, stream))

    def evaluate_group(self, net, group, n_groups, stream=None):
        check(lib.agx_engine_evaluate_group(self._h, net._net, group, n_groups, stream))

    def expand_backup_group(self, group, n_groups, stream=None):
        check(lib.agx_engine_expand_backup_group(self._h, group, n_groups, stream))

    # ---- external evaluation plumbing (tests) ----
    def scheduled(self):
        """returns (slot list, features uint32 [n][cells]) of the positions awaiting evaluation"""
        check(lib.agx_device_synchronize())
        count = np.zeros(1, np.int32)
        check(lib.agx_memcpy_d2h(count.ctypes.data_as(ctypes.c_void_p), self.buffers.d_nn_count, 4))
        n = int(count[0])
        slots = np.zeros(max(n, 1), np.int32)
        check(lib.agx_memcpy_d2h(slots.ctypes.data_as(ctypes.c_void_p), self.buffers.d_nn_list, 4 * max(n, 1)))
        feats = np.zeros((self.slots, self.cells), np.uint32)
        check(lib.agx_memcpy_d2h(feats.ctypes.data_as(ctypes.c_void_p), self.buffers.d_nn_features, feats.nbytes))
        slots = slots[:n]
        return slots, feats[slots]

    def scheduled_group(self, group, n_groups):
        """scheduled() for one group of a pool stepped in groups (slot numbers are pool-wide)"""
        check(lib.agx_device_synchronize())
        per = (self.cfg.n_games + n_groups - 1) // n_groups
        count = np.zeros(1, np.int32)
        check(lib.agx_memcpy_d2h(count.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(self.buffers.d_nn_count + 4 * group), 4))
        n = int(count[0])
        slots = np.zeros(max(n, 1), np.int32)
        first = group * per * self.cfg.max_batch_size
        check(lib.agx_memcpy_d2h(slots.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(self.buffers.d_nn_list + 4 * first), 4 * max(n, 1)))
        feats = np.zeros((self.slots, self.cells), np.uint32)
        check(lib.agx_memcpy_d2h(feats.ctypes.data_as(ctypes.c_void_p), self.buffers.d_nn_features, feats.nbytes))
        slots = slots[:n]
        return slots, feats[slots]

    def provide(self, slots, policy, value3, action_values=None):
        """writes policy [n][cells] and value (win, draw, loss) [n][3] of the given slots (and, for a pool configured with
        action_values, q (win, draw) [n][cells][2])"""
        # the other slots keep what they held: a game whose batch waits for larger arenas reads its outputs one step later
        if not hasattr(self, "_pol"):
            self._pol = np.zeros((self.slots, self.cells), np.float32)
            self._val = np.zeros((self.slots, 3), np.float32)
            self._q = np.zeros((self.slots, self.cells, 2), np.float32)
        self._pol[slots] = policy
        self._val[slots] = value3
        check(lib.agx_memcpy_h2d(self.buffers.d_nn_policy, self._pol.ctypes.data_as(ctypes.c_void_p), self._pol.nbytes))
        check(lib.agx_memcpy_h2d(self.buffers.d_nn_value, self._val.ctypes.data_as(ctypes.c_void_p), self._val.nbytes))
        if action_values is not None:
            self._q[slots] = action_values
            check(lib.agx_memcpy_h2d(self.buffers.d_nn_action_values, self._q.ctypes.data_as(ctypes.c_void_p), self._q.nbytes))

    def generate_openings(self, net, count, seed=0):
        """OpeningGenerator::generate: `count` solver-unproven, network-balanced openings (before begin()); returns
        (list of Move::toShort lists, stats dict)"""
        out = np.zeros((count, OPENING_CAP), dtype=np.uint16)
        st = (ctypes.c_int * 4)()
        check(lib.agx_engine_generate_openings(self._h, net._net, count, seed, out.ctypes.data_as(ctypes.c_void_p), st))
        names = ["candidates", "proven_by_solver", "unbalanced", "network_evaluations"]
        return [[int(x) for x in row[1:1 + int(row[0])]] for row in out], dict(zip(names, list(st)))

    def kernel_timing(self, enable):
        """(ms[4], launches[4]) of k_select / k_solve / k_expand / k_advance since the previous call; then recording on/off"""
        ms = (ctypes.c_double * 4)()
        n = (ctypes.c_longlong * 4)()
        check(lib.agx_engine_kernel_timing(self._h, 1 if enable else 0, ms, n))
        return list(ms), list(n)

    def device_bytes(self):
        """device memory the pool's engine holds (agx_engine_device_bytes)"""
        n = ctypes.c_ulonglong()
        check(lib.agx_engine_device_bytes(self._h, ctypes.byref(n)))
        return int(n.value)

    def speculative_waves(self):
        """waves of the speculative search launch over the whole pool (agx_engine_speculative_waves); 0 = serial solver"""
        n = ctypes.c_int()
        check(lib.agx_engine_speculative_waves(self._h, ctypes.byref(n)))
        return int(n.value)

    def stats(self):
        check(lib.agx_device_synchronize())
        s = AgxEngineStats()
        check(lib.agx_engine_stats(self._h, ctypes.byref(s)))
        return {name: getattr(s, name) for name, _ in s._fields_}

    def game_info(self, game, with_edges=True):
        info = AgxGameInfo()
        board = np.zeros(self.cells, np.uint8)
        edges = (AgxEdgeView * 400)()
        check(lib.agx_engine_game_info(self._h, game, ctypes.byref(info), board.ctypes.data_as(ctypes.c_void_p),
                                       ctypes.cast(edges, ctypes.c_void_p) if with_edges else None, 400))
        out = {name: getattr(info, name) for name, _ in info._fields_}
        out["board"] = board
        if with_edges:
            out["edges"] = [dict(move=e.move, visits=e.visits, prior=e.prior, win=e.win, draw=e.draw, score=e.score,
                                 flag_vl=e.flag_and_virtual_loss) for e in edges[:info.root_edges]]
        return out

    def records(self, drain=False):
        """(move records, their root-edge snapshots) produced so far; drain=True also empties the device-side pools"""
        nr, ne = ctypes.c_int(), ctypes.c_int()
        check(lib.agx_engine_records(self._h, None, 0, None, 0, ctypes.byref(nr), ctypes.byref(ne)))
        recs = (AgxMoveRecord * max(nr.value, 1))()
        edges = (AgxEdgeView * max(ne.value, 1))()
        fn = lib.agx_engine_drain_records if drain else lib.agx_engine_records
        check(fn(self._h, ctypes.cast(recs, ctypes.c_void_p), max(nr.value, 1), ctypes.cast(edges, ctypes.c_void_p), max(ne.value, 1),
                 ctypes.byref(nr), ctypes.byref(ne)))
        return recs[:nr.value], edges[:ne.value]

    def fetch_records(self, drain=False):
        """everything the record pools hold: (move records, root-edge snapshots, format-201 sample bytes as uint8 array, finished games);
        a record's sample is samples[r.sample_offset : r.sample_offset + r.sample_bytes]"""
        counts = AgxRecordCounts()
        check(lib.agx_engine_fetch_records(self._h, None, 0, None, 0, None, 0, None, 0, ctypes.byref(counts), 0))
        recs = (AgxMoveRecord * max(counts.records, 1))()
        edges = (AgxEdgeView * max(counts.edges, 1))()
        samples = np.zeros(max(counts.sample_bytes, 4), np.uint8)
        ends = (AgxGameEnd * max(counts.game_ends, 1))()
        check(lib.agx_engine_fetch_records(self._h, ctypes.cast(recs, ctypes.c_void_p), len(recs), ctypes.cast(edges, ctypes.c_void_p), len(edges),
                                           samples.ctypes.data_as(ctypes.c_void_p), samples.size, ctypes.cast(ends, ctypes.c_void_p), len(ends),
                                           ctypes.byref(counts), 1 if drain else 0))
        return recs[:counts.records], edges[:counts.edges], samples[:counts.sample_bytes], ends[:counts.game_ends]

    def step_match(self, first_net, second_net, stream=None):
        """one step of a match_mode pool: every stage one launch over both players' trees, the network stage per player"""
        check(lib.agx_engine_step_match(self._h, first_net._net, second_net._net, stream))

    def step_match_groups(self, first_net, second_net, stream=None):
        """the same as two group steps: the first players' trees with their network, then the second players' (one stream)"""
        check(lib.agx_engine_step_group(self._h, first_net._net, 0, 2, stream))
        check(lib.agx_engine_step_group(self._h, second_net._net, 1, 2, stream))

    def select_solve_match(self, stream=None):
        check(lib.agx_engine_select_solve_match(self._h, stream))

    def expand_backup_match(self, stream=None):
        check(lib.agx_engine_expand_backup_match(self._h, stream))

    def match_results(self):
        """int [pairs][4]: games won / drawn / lost by the first player of each pair, games finished by the pair"""
        out = np.zeros((self.cfg.n_games // 2, 4), np.int32)
        check(lib.agx_engine_match_results(self._h, out.ctypes.data_as(ctypes.c_void_p), out.shape[0]))
        return out

    def add_openings(self, packed_openings):
        """appends openings (pack_openings layout) for the games that finish from now on"""
        a = np.ascontiguousarray(packed_openings, dtype=np.uint16)
        check(lib.agx_engine_add_openings(self._h, a.ctypes.data_as(ctypes.c_void_p), a.shape[0]))

    def zobrist(self):
        keys = np.zeros(4 * self.cells, np.uint64)
        check(lib.agx_engine_zobrist(self._h, keys.ctypes.data_as(ctypes.c_void_p), keys.size))
        return keys

    def debug_solve(self, boards, signs):
        boards = np.ascontiguousarray(boards, dtype=np.uint8).reshape(-1, self.cells)
        signs = np.ascontiguousarray(signs, dtype=np.int32)
        n = boards.shape[0]
        feats = np.zeros((n, self.cells), np.uint32)
        moves = np.zeros((n, self.cells), np.uint16)
        scores = np.zeros((n, self.cells), np.uint16)
        counts = np.zeros(n, np.int32)
        flags = np.zeros(n, np.uint32)
        results = np.zeros(n, np.uint16)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        check(lib.agx_debug_solve(self._h, p(boards), p(signs), n, p(feats), p(moves), p(scores), p(counts), p(flags), p(results)))
        nodes = np.zeros(n, np.uint64)
        check(lib.agx_debug_solve_nodes(self._h, n, p(nodes)))
        return dict(features=feats, moves=moves, scores=scores, counts=counts, flags=flags, results=results, nodes=nodes)

    def debug_pattern_state(self, boards, signs, moves):
        boards = np.ascontiguousarray(boards, dtype=np.uint8).reshape(-1, self.cells)
        signs = np.ascontiguousarray(signs, dtype=np.int32)
        moves = np.ascontiguousarray(moves, dtype=np.uint16)
        n = boards.shape[0]
        n_moves = moves.shape[1] if moves.size else 0
        stride = 2 * self.cells * 2 + 32
        pt = np.zeros((n, self.cells, 8), np.uint8)
        th = np.zeros((n, self.cells, 2), np.uint8)
        lists = np.zeros((n, stride), np.int16)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        check(lib.agx_debug_pattern_state(self._h, p(boards), p(signs), p(moves), n, n_moves, p(pt), p(th), p(lists), stride))
        return pt, th, lists

    def close(self):
        if self._h:
            lib.agx_engine_destroy(self._h)
            self._h = ctypes.c_void_p()


class GameBuffer:
    """GameDataBuffer + GeneratorManager::addToBuffer (src/dataset/GameDataBuffer.cpp, src/selfplay/GeneratorManager.cpp:160-164): finished games in
    the reference's dataset format 201; the samples are quantised on the device (record_format bit 1)"""

    def __init__(self, rules, rows, cols, draw_after=0):
        self._h = ctypes.c_void_p()
        check(lib.agx_game_buffer_create(rules, rows, cols, draw_after, ctypes.byref(self._h)))

    def collect(self, pool):
        """drains the pool's records and appends the games that have finished; returns how many were added"""
        added = ctypes.c_int()
        check(lib.agx_game_buffer_collect(self._h, pool._h, ctypes.byref(added)))
        return added.value

    def stats(self):
        s = AgxGameBufferStats()
        check(lib.agx_game_buffer_stats(self._h, ctypes.byref(s)))
        return {name: getattr(s, name) for name, _ in s._fields_}

    def game(self, index):
        """GameDataStorage::serialize bytes of one game"""
        size = ctypes.c_size_t()
        check(lib.agx_game_buffer_game(self._h, index, None, 0, ctypes.byref(size)))
        out = np.zeros(size.value, np.uint8)
        check(lib.agx_game_buffer_game(self._h, index, out.ctypes.data_as(ctypes.c_void_p), out.size, ctypes.byref(size)))
        return out

    def save(self, path, compress=True):
        check(lib.agx_game_buffer_save(self._h, str(path).encode(), 1 if compress else 0))

    def close(self):
        if self._h:
            lib.agx_game_buffer_destroy(self._h)
            self._h = ctypes.c_void_p()
