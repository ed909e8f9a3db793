"""ag::setupSignalHandler / ag::hasCapturedSignal of the reference-named boundary (include/alphagomoku_agx/selfplay.hpp; utils/os_utils.hpp:46-63 in the reference):
the flag GeneratorManager::generate polls.  Host code only — runs in a child process (the handler replaces the interpreter's own SIGINT handler)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, os, signal, sys
lib = ctypes.CDLL(os.path.join(sys.argv[1], "alphagomoku_amd", "libagx_ag.so"))
setup = lib._ZN2ag18setupSignalHandlerENS_10SignalTypeENS_17SignalHandlerModeE
setup.argtypes = [ctypes.c_int, ctypes.c_int]
setup.restype = None
captured = lib._ZN2ag17hasCapturedSignalENS_10SignalTypeE
captured.argtypes = [ctypes.c_int]
captured.restype = ctypes.c_bool
INT, TERM = 0, 5                      # SignalType
DEFAULT, IGNORE, CUSTOM = 0, 1, 2     # SignalHandlerMode
assert not captured(INT) and not captured(TERM)
setup(INT, CUSTOM)
os.kill(os.getpid(), signal.SIGINT)   # with the custom handler installed the process survives and the flag is set
assert captured(INT) and not captured(TERM)
assert captured(INT)                  # a captured signal stays captured, as in the reference
setup(TERM, IGNORE)
os.kill(os.getpid(), signal.SIGTERM)  # ignored: no flag, no death
assert not captured(TERM)
print("ok")
'''


def test_custom_sigint_handler_sets_the_flag_generate_polls():
    lib = os.path.join(ROOT, "alphagomoku_amd", "libagx_ag.so")
    assert os.path.exists(lib), "libagx_ag.so is built by alphagomoku_amd/build.py"
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stderr[-2000:]
