"""ctypes binding of include/agx.h.  Fails loudly when the HIP library is missing: there is no CPU fallback."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AGX_LIB_PATH") or os.path.join(_HERE, "libagx.so")   # (AGX_LIB_PATH: developer builds, e.g. scripts/sanitize_cpu.sh)


class AgxError(RuntimeError):
    pass


class AgxNetDesc(ctypes.Structure):
    _fields_ = [("rows", ctypes.c_int), ("cols", ctypes.c_int), ("blocks", ctypes.c_int),
                ("filters", ctypes.c_int), ("in_channels", ctypes.c_int), ("value_hidden", ctypes.c_int),
                ("action_values", ctypes.c_int)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise AgxError(
            "HIP extension %s is missing — build it with `python -m alphagomoku_amd.build` "
            "(there is no CPU fallback for the product path)" % LIB_PATH)
    return ctypes.CDLL(LIB_PATH)


class _Lazy:
    """Loads libagx.so on first attribute access so that importing the package never touches the GPU."""
    _cdll = None

    def _get(self):
        if _Lazy._cdll is None:
            cdll = _load()
            _declare(cdll)
            _Lazy._cdll = cdll
        return _Lazy._cdll

    def __getattr__(self, name):
        return getattr(self._get(), name)


def _declare(c):
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    c.agx_last_error.restype = ctypes.c_char_p
    c.agx_last_error.argtypes = []
    c.agx_version.restype = ci
    c.agx_build_hash.restype = ctypes.c_char_p
    c.agx_build_hash.argtypes = []
    c.agx_set_device.argtypes = [ci]
    c.agx_device_cu_count.argtypes = [ctypes.POINTER(ci)]
    c.agx_net_blob_floats.restype = sz
    c.agx_net_blob_floats.argtypes = [ctypes.POINTER(AgxNetDesc)]
    c.agx_net_create.argtypes = [ctypes.POINTER(AgxNetDesc), ctypes.POINTER(vp)]
    c.agx_net_load_weights.argtypes = [vp, vp, sz]
    c.agx_nn_forward.argtypes = [vp, vp, ci, vp, vp, vp]
    c.agx_net_description.argtypes = [vp, vp]
    c.agx_net_set_launch_width.argtypes = [vp, ci]
    c.agx_nn_forward_pvq.argtypes = [vp, vp, ci, vp, vp, vp, vp]
    c.agx_nn_forward_indirect_pvq.argtypes = [vp, vp, vp, vp, ci, vp, vp, vp, vp]
    c.agx_net_destroy.argtypes = [vp]
    c.agx_malloc.argtypes = [ctypes.POINTER(vp), sz]
    c.agx_free.argtypes = [vp]
    c.agx_memcpy_h2d.argtypes = [vp, vp, sz]
    c.agx_memcpy_d2h.argtypes = [vp, vp, sz]
    c.agx_memset.argtypes = [vp, ci, sz]
    c.agx_device_synchronize.argtypes = []
    c.agx_timer_create.argtypes = [ctypes.POINTER(vp)]
    c.agx_timer_start.argtypes = [vp, vp]
    c.agx_timer_stop.argtypes = [vp, vp]
    c.agx_timer_elapsed_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    c.agx_timer_poll_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
    c.agx_timer_destroy.argtypes = [vp]


lib = _Lazy()


def require_current_build():
    """Raises unless libagx.so was compiled from the sources beside it (its agx_build_hash() == build.source_hash())."""
    from . import build
    have, want = lib.agx_build_hash().decode(), build.source_hash()
    if have != want and not os.environ.get("AGX_NO_BUILD"):   # (AGX_NO_BUILD: developer A/B runs with swapped-in variants)
        raise AgxError("libagx.so was built from other sources (library %s, sources %s): run `python -m alphagomoku_amd.build`" % (have, want))
    return have


def check(status):
    if status != 0:
        raise AgxError("agx error %d: %s" % (status, lib.agx_last_error().decode()))


class AgxEngineConfig(ctypes.Structure):
    _fields_ = [("rules", ctypes.c_int), ("board_size", ctypes.c_int), ("draw_after", ctypes.c_int), ("n_games", ctypes.c_int),
                ("max_batch_size", ctypes.c_int), ("max_simulations", ctypes.c_int), ("exploration_constant", ctypes.c_float),
                ("exploration_scaling", ctypes.c_float), ("init_to", ctypes.c_int), ("information_leak_threshold", ctypes.c_float),
                ("policy_expansion_threshold", ctypes.c_float), ("tss_max_positions", ctypes.c_int),
                ("tss_table_entries", ctypes.c_uint64), ("zobrist_seed", ctypes.c_uint64), ("node_capacity", ctypes.c_int),
                ("edge_capacity", ctypes.c_int), ("record_capacity", ctypes.c_int), ("record_edge_capacity", ctypes.c_int),
                ("solver_yield_fraction", ctypes.c_float), ("final_selector", ctypes.c_int), ("use_symmetries", ctypes.c_int),
                ("symmetry_seed", ctypes.c_uint64), ("max_children", ctypes.c_int), ("noise_type", ctypes.c_int), ("noise_weight", ctypes.c_float),
                ("noise_seed", ctypes.c_uint64), ("action_values", ctypes.c_int), ("match_mode", ctypes.c_int), ("policy_temperature", ctypes.c_float),
                ("arena_reserve", ctypes.c_float), ("search_threads", ctypes.c_int), ("record_format", ctypes.c_int), ("record_sample_capacity", ctypes.c_int), ("game_end_capacity", ctypes.c_int),
                ("speculative_solver", ctypes.c_int), ("speculative_waves", ctypes.c_int), ("force_expand_root", ctypes.c_int), ("search_buffers", ctypes.c_int)]


class AgxEngineBuffers(ctypes.Structure):
    _fields_ = [("d_nn_features", ctypes.c_void_p), ("d_nn_policy", ctypes.c_void_p), ("d_nn_value", ctypes.c_void_p),
                ("d_nn_action_values", ctypes.c_void_p),
                ("d_nn_list", ctypes.c_void_p), ("d_nn_count", ctypes.c_void_p), ("slots", ctypes.c_int), ("cells", ctypes.c_int)]


class AgxEngineStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_ulonglong) for n in
                ["evaluated_nodes", "network_evaluations", "information_leaks", "proven_edge_visits", "wasted_expansions",
                 "duplicate_selections", "solver_nodes", "select_levels", "select_edge_reads", "moves_played", "peak_nodes",
                 "peak_edges"]] + [(n, ctypes.c_int) for n in
                                   ["games_finished", "openings_taken", "active_games", "records_used", "record_edges_used",
                                    "first_error", "arena_grows", "arena_releases", "arena_failures", "arena_max_class"]] + \
               [("arena_heap_used", ctypes.c_float), ("speculative_parks", ctypes.c_int), ("speculative_solves", ctypes.c_ulonglong),
                ("speculative_reruns", ctypes.c_ulonglong), ("speculative_deferrals", ctypes.c_ulonglong)]


class AgxEdgeView(ctypes.Structure):
    _fields_ = [("prior", ctypes.c_float), ("win", ctypes.c_float), ("draw", ctypes.c_float), ("visits", ctypes.c_int32),
                ("move", ctypes.c_uint16), ("score", ctypes.c_uint16), ("flag_and_virtual_loss", ctypes.c_uint16),
                ("reserved", ctypes.c_uint16)]


class AgxGameInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ["active", "sign_to_move", "n_moves", "outcome", "error", "opening_id", "games_done",
                                            "n_nodes", "n_edges", "root_visits"]] + \
               [("root_win", ctypes.c_float), ("root_draw", ctypes.c_float), ("root_score", ctypes.c_int), ("root_edges", ctypes.c_int),
                ("grow_pending", ctypes.c_int), ("arena_class", ctypes.c_int), ("root_moves_left", ctypes.c_float), ("max_depth", ctypes.c_int)]


class AgxMoveRecord(ctypes.Structure):
    _fields_ = [("game_serial", ctypes.c_int), ("move_number", ctypes.c_int), ("move", ctypes.c_uint16), ("root_score", ctypes.c_uint16),
                ("root_visits", ctypes.c_int), ("root_win", ctypes.c_float), ("root_draw", ctypes.c_float), ("n_edges", ctypes.c_int),
                ("edge_offset", ctypes.c_int), ("root_flags", ctypes.c_int), ("game_slot", ctypes.c_int), ("game_index", ctypes.c_int),
                ("sample_offset", ctypes.c_int), ("sample_bytes", ctypes.c_int), ("outcome", ctypes.c_int)]


class AgxGameEnd(ctypes.Structure):
    _fields_ = [("game_serial", ctypes.c_int), ("game_slot", ctypes.c_int), ("game_index", ctypes.c_int), ("outcome", ctypes.c_int),
                ("n_moves", ctypes.c_int), ("moves", ctypes.c_uint16 * 400)]


class AgxSavedGame(ctypes.Structure):
    _fields_ = [("game_slot", ctypes.c_int), ("game_index", ctypes.c_int), ("opening_id", ctypes.c_int), ("sign_to_move", ctypes.c_int),
                ("nn_queued", ctypes.c_int), ("n_moves", ctypes.c_int), ("moves", ctypes.c_uint16 * 400)]


class AgxRecordCounts(ctypes.Structure):
    _fields_ = [("records", ctypes.c_int), ("edges", ctypes.c_int), ("sample_bytes", ctypes.c_int), ("game_ends", ctypes.c_int)]


class AgxGameBufferStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ["games", "samples", "cross_win", "draws", "circle_win", "game_length"]]


_declare_nn = _declare


def _declare(c):  # noqa: F811
    _declare_nn(c)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    c.agx_nn_forward_indirect.argtypes = [vp, vp, vp, vp, ci, vp, vp, vp]
    c.agx_engine_default_config.argtypes = [ctypes.POINTER(AgxEngineConfig)]
    c.agx_engine_create.argtypes = [ctypes.POINTER(AgxEngineConfig), ctypes.POINTER(vp)]
    c.agx_engine_destroy.argtypes = [vp]
    c.agx_engine_begin.argtypes = [vp, vp, ci, vp]
    for name in ["agx_engine_select_solve", "agx_engine_expand_backup"]:
        getattr(c, name).argtypes = [vp, vp]
    for name in ["agx_engine_evaluate", "agx_engine_step"]:
        getattr(c, name).argtypes = [vp, vp, vp]
    for name in ["agx_engine_select_solve_group", "agx_engine_expand_backup_group", "agx_engine_expand_group", "agx_engine_advance_group"]:
        getattr(c, name).argtypes = [vp, ci, ci, vp]
    for name in ["agx_engine_evaluate_group", "agx_engine_step_group"]:
        getattr(c, name).argtypes = [vp, vp, ci, ci, vp]
    c.agx_stream_create.argtypes = [ctypes.POINTER(vp)]
    c.agx_stream_create_with_cu_mask.argtypes = [ctypes.POINTER(vp), vp, ci]
    c.agx_stream_create_with_cu_mask_instance.argtypes = [ctypes.POINTER(vp), vp, ci, ci]
    c.agx_event_create.argtypes = [ctypes.POINTER(vp)]
    c.agx_event_create_blocking.argtypes = [ctypes.POINTER(vp)]
    c.agx_event_synchronize.argtypes = [vp]
    c.agx_event_record.argtypes = [vp, vp]
    c.agx_stream_wait_event.argtypes = [vp, vp]
    c.agx_event_destroy.argtypes = [vp]
    c.agx_stream_masked_count.argtypes = []
    c.agx_stream_destroy.argtypes = [vp]
    c.agx_stream_synchronize.argtypes = [vp]
    c.agx_engine_set_board.argtypes = [vp, ci, vp, ci, vp]
    c.agx_engine_set_max_simulations.argtypes = [vp, ci]
    c.agx_engine_set_batch_size.argtypes = [vp, ci]
    c.agx_engine_solve_timed_group.argtypes = [vp, ci, ci, ci, ctypes.c_double, vp]
    c.agx_engine_select_group.argtypes = [vp, ci, ci, vp]
    c.agx_engine_save_games.argtypes = [vp, vp, ci, ctypes.POINTER(ci)]
    c.agx_engine_restore_game.argtypes = [vp, ctypes.POINTER(AgxSavedGame), vp]
    c.agx_engine_set_force_expand_root.argtypes = [vp, ci]
    c.agx_engine_cancel_pending.argtypes = [vp, vp]
    c.agx_engine_root_summary.argtypes = [vp, ci, vp, ctypes.POINTER(ctypes.c_int)]
    c.agx_engine_buffers.argtypes = [vp, ctypes.POINTER(AgxEngineBuffers)]
    c.agx_engine_stats.argtypes = [vp, ctypes.POINTER(AgxEngineStats)]
    c.agx_engine_device_bytes.argtypes = [vp, ctypes.POINTER(ctypes.c_ulonglong)]
    c.agx_engine_estimate_device_bytes.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_ulonglong)]
    c.agx_engine_speculative_waves.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
    c.agx_engine_kernel_timing.argtypes = [vp, ci, vp, vp]
    c.agx_engine_add_openings.argtypes = [vp, vp, ci]
    c.agx_engine_match_results.argtypes = [vp, vp, ci]
    c.agx_engine_step_match.argtypes = [vp, vp, vp, vp]
    c.agx_engine_select_solve_match.argtypes = [vp, vp]
    c.agx_engine_expand_backup_match.argtypes = [vp, vp]
    c.agx_engine_drain_records.argtypes = [vp, vp, ci, vp, ci, vp, vp]
    c.agx_engine_generate_openings.argtypes = [vp, vp, ci, ctypes.c_uint32, vp, vp]
    c.agx_engine_game_info.argtypes = [vp, ci, ctypes.POINTER(AgxGameInfo), vp, vp, ci]
    c.agx_engine_records.argtypes = [vp, vp, ci, vp, ci, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    c.agx_engine_zobrist.argtypes = [vp, vp, sz]
    c.agx_engine_fetch_records.argtypes = [vp, vp, ci, vp, ci, vp, ci, vp, ci, ctypes.POINTER(AgxRecordCounts), ci]
    c.agx_game_buffer_create.argtypes = [ci, ci, ci, ci, ctypes.POINTER(vp)]
    c.agx_game_buffer_destroy.argtypes = [vp]
    c.agx_game_buffer_clear.argtypes = [vp]
    c.agx_game_buffer_collect.argtypes = [vp, vp, ctypes.POINTER(ci)]
    c.agx_game_buffer_stats.argtypes = [vp, ctypes.POINTER(AgxGameBufferStats)]
    c.agx_game_buffer_game.argtypes = [vp, ci, vp, sz, ctypes.POINTER(sz)]
    c.agx_game_buffer_save.argtypes = [vp, ctypes.c_char_p, ci]
    c.agx_game_buffer_load.argtypes = [vp, ctypes.c_char_p]
    c.agx_game_buffer_take_pending.argtypes = [vp, vp, ci, ci, vp, sz, ctypes.POINTER(sz)]
    c.agx_game_buffer_restore_pending.argtypes = [vp, vp, ci, ci, vp, sz]
    c.agx_game_buffer_forget_engine.argtypes = [vp, vp]
    c.agx_sample_v201_unpack.argtypes = [vp, sz, ci, ci, vp, vp, vp, vp, vp, vp, ctypes.POINTER(sz)]
    c.agx_debug_solve.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp]
    c.agx_debug_new_generation.argtypes = [vp]
    c.agx_debug_solve_nodes.argtypes = [vp, ci, vp]
    c.agx_debug_pattern_state.argtypes = [vp, vp, vp, vp, ci, ci, vp, vp, vp, ci]
    c.agx_host_tables.argtypes = [ci, vp, vp, vp, vp]
    c.agx_make_opening.argtypes = [ci, ci, ctypes.c_uint32, vp]
    c.agx_get_outcome.argtypes = [ci, ci, vp, ci, ci, ci, ci, vp]
