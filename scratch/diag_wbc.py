import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import oracle_py as O
import gpu_helpers as G
pkg = load_pkg()
ctx = pkg.Context(0, 4096, 16)
G.setup_a1(ctx, pkg, 10)
b = pkg.make_batch(256, 10, "a1", seed=51)
got = G.run_fb_debug(ctx, pkg, b)
md = pkg.model_desc("a1")
errs = {k: 0.0 for k in ("H","G","C","Jc","Jcdqd","pGC","vGC")}
Herr = np.zeros((18,18))
for i in range(64):
    r = O.fb_compute(md, b["fb_state"][i].astype(np.float64), np.float64)
    for k in errs: errs[k] = max(errs[k], np.abs(got[k][i]-r[k]).max())
    Herr = np.maximum(Herr, np.abs(got["H"][i]-r["H"]))
print("fb max abs errs (gpu fp32-rounded output vs oracle f64):", {k: "%.2e"%v for k,v in errs.items()})
np.set_printoptions(linewidth=250, precision=1)
print("H err pattern (x1e-9):"); print(Herr*1e9)
out = G.run_wbc(ctx, pkg, b)
tau64 = np.zeros((256,12)); 
for i in range(256):
    r = O.wbc_run(md, b["fb_state"][i].astype(np.float64), b["wbc_cmd"][i].astype(np.float64), b["prev_ori_vel"][i].astype(np.float64), dtype=np.float64)
    tau64[i] = r["tau"]
e = np.abs(out["tau"]-tau64)
worst = np.argsort(e.max(axis=1))[::-1][:8]
np.set_printoptions(linewidth=250, precision=3, suppress=False)
for i in worst:
    print(i, "contact", b["wbc_cmd"][i,63:67], "err", e[i], "tau", tau64[i])
print("median err", np.median(e), "frac > 1e-6", (e>1e-6).mean())
