"""create / run / destroy many solvers: device memory must come back"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg2png_amd as j
from jpeg2png_amd import synth
import resource
hip = j.hip_runtime()
hip.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
def free_mem():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value
planes3 = synth.make_planes(1024, 768, "420", 10, seed=1)
planes1 = synth.make_planes(1024, 1024, "444", 10, seed=2, y_only=True)
for p in planes3 + planes1:
    p.fdata = j.decode_plane(p)
import copy
j.compute(copy.deepcopy(planes3), 0.3, [0.001] * 3, 2)
base = free_mem(); rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for i in range(150):
    j.compute(copy.deepcopy(planes3), 0.3, [0.001] * 3, 3, log=(i % 2 == 0))
    with j.Solver(planes1, 0.3, [0.001], 3, band=(256, 768)) as s:
        s.set_logging(True); s.phase_gradient(); s.phase_project()
    j.decode_plane(planes1[0])
after = free_mem(); rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print(f"device free before {base >> 20} MiB, after {after >> 20} MiB, delta {(base - after) >> 20} MiB; host max RSS {rss0 >> 10} -> {rss1 >> 10} MiB")
sys.exit(0 if base - after < (64 << 20) else 1)
