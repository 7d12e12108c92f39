import sys, os, ctypes, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
L = importlib.import_module("efficientlo-net_amd._lib")
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libelo_timed.so")
L._lib = None
import subprocess
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sys.argv = ["x", "--kernel", "cv1", "--batch", str(batch), "--reps", "5"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "roofline_micro.py")).read())
lib = L.lib()
buf = (ctypes.c_longlong * (16 * 64))()
lib.elo_debug_phases.argtypes = [ctypes.c_void_p]; lib.elo_debug_phases.restype = ctypes.c_int
assert lib.elo_debug_phases(buf) == 0
t = np.array(buf).reshape(64, 16)[:, :10]
d = np.diff(t, axis=1)       # wall_clock64 ticks at 100 MHz => 10 ns
names = ["group", "gather", "barrier", "cv0", "cv1", "cv2", "cv_xyz", "sum_cv0", "sum_cv1", "pool"]
print("median phase durations over 64 blocks (us):")
for i, n in enumerate(names[:9]):
    print("  %-8s -> %-8s %6.2f" % (n, names[i + 1] if i + 1 < len(names) else "", np.median(d[:, i]) / 100.0))
print("  total %.2f" % (np.median(t[:, 9] - t[:, 0]) / 100.0))
T = np.array(buf).reshape(64, 16)
print("pool detail: start->max %.2f, max->exp %.2f, exp->store %.2f, store->end %.2f" % tuple(np.median(x)/100.0 for x in (T[:,10]-T[:,8], T[:,11]-T[:,10], T[:,12]-T[:,11], T[:,9]-T[:,12])))
