import sys, os, copy
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'tests'))
import numpy as np
import jpeg2png_amd as j
from conftest import make_case
from oracle import bindings as ob
planes = make_case(64, 48, "444", 10, seed=1234+7, y_only=True)
for its in (1,2):
    for w in (0.3, 0.0):
        want,_ = ob.oracle_compute(planes, w, [0.001], its)
        got = copy.deepcopy(planes); j.compute(got, w, [0.001], its)
        g = got[0].fdata
        print("its",its,"w",w,"nan count", np.isnan(g).sum(), "maxdiff", np.nanmax(np.abs(g-want[0])), "neq", (g!=want[0]).sum())
        if np.isnan(g).any():
            ys,xs = np.where(np.isnan(g)); print(" nan rows", np.unique(ys)[:20], "cols", np.unique(xs)[:20])
        else:
            ys,xs = np.where(g!=want[0]); print(" diff rows", np.unique(ys)[:20], "cols", np.unique(xs)[:40])
