"""ctypes loader of the CPU oracle (tests / smoke / bench cpu_baseline only)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RULES = {"FREESTYLE": 0, "STANDARD": 1, "RENJU": 2, "CARO5": 3, "CARO6": 4}
SIGNS = {"NONE": 0, "CROSS": 1, "CIRCLE": 2}
MODES = {"BASIC": 0, "THREATS": 1, "OPTIMAL": 2, "REDUCED": 3, "LEGAL": 4}
OUTCOMES = {"UNKNOWN": 0, "DRAW": 1, "CROSS_WIN": 2, "CIRCLE_WIN": 3}


class AgoSearchConfig(ctypes.Structure):
    _fields_ = [("max_batch_size", ctypes.c_int), ("exploration_constant", ctypes.c_float),
                ("exploration_scaling", ctypes.c_float), ("init_to", ctypes.c_int), ("max_children", ctypes.c_int),
                ("policy_expansion_threshold", ctypes.c_float), ("information_leak_threshold", ctypes.c_float),
                ("tss_max_positions", ctypes.c_int), ("tss_table_entries", ctypes.c_uint64),
                ("max_simulations", ctypes.c_int), ("zobrist_seed", ctypes.c_uint64),
                ("final_selector", ctypes.c_int), ("use_symmetries", ctypes.c_int), ("symmetry_seed", ctypes.c_uint64),
                ("noise_type", ctypes.c_int), ("noise_weight", ctypes.c_float), ("noise_seed", ctypes.c_uint64)]


def default_search_config(max_batch_size=8, max_simulations=400, table_entries=1 << 16, final_selector=0, use_symmetries=0, noise_type=0,
                          noise_weight=0.0):
    return AgoSearchConfig(max_batch_size, 1.25, 0.0, 0, 2 ** 31 - 1, 1.0e-4, 0.01, 100, table_entries,
                           max_simulations, 0x9E3779B97F4A7C15, final_selector, use_symmetries, 0x5DEECE66D, noise_type, noise_weight,
                           0x2545F4914F6CDD1D)


_lib = None


def load():
    global _lib
    if _lib is None:
        path = os.environ.get("AGO_LIB_PATH") or os.path.join(ROOT, "oracle", "libagoracle.so")   # (AGO_LIB_PATH: scripts/sanitize_cpu.sh)
        if "AGO_LIB_PATH" not in os.environ and os.path.exists(os.path.join(ROOT, "oracle", "Makefile")):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libagoracle.so"])
        lib = ctypes.CDLL(path)
        for name in ["ago_defensive_moves", "ago_open_three_promotion_moves", "ago_score_op", "ago_score_make", "ago_move_to_short"]:
            getattr(lib, name).restype = ctypes.c_uint16
        lib.ago_game_create_ex.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(AgoSearchConfig)]
        for name in ["ago_solver_create", "ago_game_create", "ago_game_create_ex"]:
            getattr(lib, name).restype = ctypes.c_void_p
        lib.ago_solver_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
        lib.ago_game_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(AgoSearchConfig)]
        for name in ["ago_solver_destroy", "ago_solver_new_generation", "ago_game_destroy"]:
            getattr(lib, name).argtypes = [ctypes.c_void_p]
        lib.ago_solver_zobrist.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.ago_solver_solve.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6
        lib.ago_game_begin.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.ago_game_set_serial.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.ago_game_set_search_threads.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.ago_apply_symmetry.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.ago_game_set_force_expand_root.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.ago_game_record_flags.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.ago_game_set_policy_temperature.argtypes = [ctypes.c_void_p, ctypes.c_float]
        lib.ago_game_match_begin.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.ago_game_take_turn.argtypes = [ctypes.c_void_p]
        lib.ago_game_external_move.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.ago_game_last_move.argtypes = [ctypes.c_void_p]
        lib.ago_game_sign_to_move.argtypes = [ctypes.c_void_p]
        lib.ago_game_step_select.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.ago_game_step_expand.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.ago_game_async_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.ago_game_async_provide.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.ago_game_async_provide.restype = None
        lib.ago_game_step_expand_q.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.ago_game_outcome.argtypes = [ctypes.c_void_p]
        lib.ago_game_num_records.argtypes = [ctypes.c_void_p]
        lib.ago_game_record.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int]
        lib.ago_game_root.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 9 + [ctypes.c_int]
        lib.ago_game_tree_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.ago_game_tree_info.restype = None
        lib.ago_game_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.ago_cpu_baseline.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(AgoSearchConfig), ctypes.c_int, ctypes.c_int,
                                         ctypes.c_double] + [ctypes.c_void_p] * 5
        vp, ci = ctypes.c_void_p, ctypes.c_int
        lib.ago_sample_v201_pack.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp, ctypes.c_uint16, ci, vp, ci]
        lib.ago_sample_v201_unpack.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, vp]
        lib.ago_game_record_v201.argtypes = [vp, ci, vp, ci]
        lib.ago_game_storage_v201.argtypes = [vp, vp, ci]
        lib.ago_score_to_int8.argtypes = [ctypes.c_uint16]
        lib.ago_int8_to_score.argtypes = [ci]
        lib.ago_int8_to_score.restype = ctypes.c_uint16
        lib.ago_prepare_opening.argtypes = [ci, ci, ci, ctypes.c_uint32, vp]
        lib.ago_fake_eval.argtypes = [ci, ci, vp, vp, vp]
        _lib = lib
    return _lib


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def board_array(board):
    return np.ascontiguousarray(np.array(board, dtype=np.uint8))


def move_short(m):
    return m["sign"] | (m["row"] << 2) | (m["col"] << 9)


def short_to_move(s):
    return dict(sign=s & 3, row=(s >> 2) & 127, col=(s >> 9) & 127)


def movegen(lib, rules, board, sign, mode, draw_after=-1):
    b = board_array(board)
    rows, cols = b.shape
    moves = np.zeros(1024, np.uint16)
    scores = np.zeros(1024, np.uint16)
    flags = ctypes.c_int()
    result = ctypes.c_uint16()
    n = lib.ago_movegen(rules, rows, cols, ptr(b), sign, mode, draw_after, ptr(moves), ptr(scores), ctypes.byref(flags), ctypes.byref(result))
    return dict(moves=[int(x) for x in moves[:n]], scores=[int(x) for x in scores[:n]], must_defend=bool(flags.value & 1),
                has_initiative=bool(flags.value & 2), fully_expanded=bool(flags.value & 4), score=result.value)


def encode_features(lib, rules, board, sign):
    b = board_array(board)
    rows, cols = b.shape
    out = np.zeros(rows * cols, np.uint32)
    lib.ago_encode_features(rules, rows, cols, ptr(b), sign, ptr(out))
    return out.reshape(rows, cols)
