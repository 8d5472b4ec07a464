import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from paml_amd import hostlib
g = helpers.load_golden("lyso_bsa")
a = hostlib.Analysis(os.path.join(ROOT, "tests/golden/ctl/lyso_bsa.ctl"), "codeml")
xb = np.array(g["x"])
x0 = xb.copy(); x0[a.ntime:] *= 1.1
lo, hi = a.bounds()
r = a.optimize(np.clip(x0, lo, hi))
xa = r["x"]
ts = np.linspace(-0.2, 1.2, 29)
xs = np.array([np.clip(xa + t * (xb - xa), lo, hi) for t in ts])
l = a.eval_batch_gpu(xs)
for t, v in zip(ts, l):
    print("%.2f %.7f" % (t, v))
# gradient at xa by central differences in double step sizes
for h in (1e-6, 1e-5, 1e-4):
    gr = []
    for i in range(a.ntime, a.np):
        e = np.zeros(a.np); e[i] = h * (abs(xa[i]) + 1)
        xp, xm = np.clip(xa + e, lo, hi), np.clip(xa - e, lo, hi)
        lp, lm = a.eval_batch_gpu(np.array([xp, xm]))
        gr.append((lp - lm) / (xp[i] - xm[i]))
    print(h, np.array2string(np.array(gr), precision=5))
