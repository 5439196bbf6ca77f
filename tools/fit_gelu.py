"""Fit of the one-transcendental GELU of the 16-bit modes (btsbot_amd/csrc/common.h, gelu_poly):
    gelu(x) = max(x, 0) - |x| 2^q(|x|),   q ~ log2 Phi(-a)
minimising the largest |error| of gelu itself over [-12, 12].  Prints the coefficients per degree."""
import numpy as np
from scipy.special import erf, log_ndtr
from scipy.optimize import least_squares
def gelu(x): return 0.5*x*(1+erf(x/np.sqrt(2)))
xs = np.linspace(-12, 12, 240001)
g = gelu(xs)
def model(c, x):
    a = np.abs(x)
    t = np.full_like(x, c[-1])
    for ck in c[-2::-1]:
        t = t*a + ck
    E = np.exp2(t)
    return np.maximum(x, 0) - a*E
for m in (2,3,4,5,6):
    ai = np.linspace(0, 6, 6001)
    y = log_ndtr(-ai)/np.log(2)
    A = np.stack([ai**j for j in range(m+1)],1)
    # weight: error in GELU = a*E*ln2*dq
    w = ai*np.exp2(y)+1e-3
    c0 = np.linalg.lstsq(A*w[:,None], y*w, rcond=None)[0]
    f = lambda c: (model(c,xs)-g)
    best=(np.abs(f(c0)).max(), c0)
    for p in (2,4,8):
        r = least_squares(lambda c: np.sign(f(c))*np.abs(f(c))**p*10.0**(3*p), best[1], method='lm', max_nfev=8000)
        e = np.abs(f(r.x)).max()
        if e<best[0]: best=(e,r.x)
    print(m, "max abs err", best[0], "coef", [float(v) for v in best[1]])
