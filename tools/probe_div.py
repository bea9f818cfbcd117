import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from canny_edge_amd import capi
from test_gpu_numerics import _weights_for
divs=set()
for s in np.arange(0.2,2.67,0.05): divs.add(_weights_for(float(np.float32(s)))[-1])
for s in (0.5,1.0,1.4,2.0): divs.update(_weights_for(s))
divs.update([1.0,0.5,0.33333334,0.7865707, 0.2, 0.9999999])
worst_all=0.0
with capi.Context(0) as c:
    for d in sorted(divs):
        n,w=c.selftest_div(d)
        worst_all=max(worst_all,w)
        if n: print(f"{d:.9g} mism={n} worst={w:.6e} log2={np.log2(w) if w>0 else 0:.2f}")
print("WORST", worst_all, np.log2(worst_all) if worst_all>0 else None)
