import sys, os, ctypes
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_lib as ol
import test_engine_gpu as teg
olib = ol.load()
kw = dict(solver_yield_fraction=float(os.environ.get("YIELD", "0.9")), speculative_solver=int(os.environ.get("SPEC", "1")), speculative_waves=int(os.environ.get("WAVES", "96")))
compared, stats = teg._play_and_compare(olib, int(os.environ.get("RULES", "0")), games=int(os.environ.get("GAMES", "16")), batch=8, sims=100, max_steps=int(os.environ.get("STEPS", "400")),
                                        evaluator=teg._stand_in_evaluator(olib, 225), table_entries=1 << int(os.environ.get("TBITS", "16")), **kw)
print("compared", compared, {k: stats[k] for k in ("first_error", "speculative_solves", "speculative_reruns", "speculative_deferrals", "solver_nodes", "evaluated_nodes")})
