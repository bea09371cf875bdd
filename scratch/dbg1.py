import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, sparkfm_amd as F
from helpers import random_problem
a = random_problem(9, 500, 64, 32, 1, 1)
ds = F.DataSet(a["row_ptr"], a["col"], a["val"], a["y"]).cache()
fm = F.FMModel(63, 32); fm.w0, fm.w, fm.v = a["w0"], a["w"], a["v"]
yh = fm.predict(ds).astype(np.float32)
w32, x32 = a["w"].astype(np.float32), a["val"].astype(np.float32)
lin = np.float32(a["w0"]) + w32[a["col"]] * x32
bad = np.nonzero(yh != lin)[0]
for r in bad[:6]:
    c = a["col"][r]; print(r, c, x32[r], w32[c], yh[r], lin[r], float(np.float32(a["w0"])), np.float32(np.float64(np.float32(a["w0"])) + np.float64(w32[c])*np.float64(x32[r])))
