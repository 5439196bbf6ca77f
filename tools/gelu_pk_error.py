"""CPU emulation (numpy float16) of the packed-f16 GELU candidate for the f16 mode: degree-3
minimax exponent on v_pk_fma_f16, 2^q on v_exp_f16, max(x, 0) - |x| 2^q packed -- against what the f16 mode ships
(degree-5 exponent in fp32, ONE rounding to f16 on the way to the fc2 operand).  Error of the hidden activation that
reaches the matrix pipe, |x| <= 4, against the exact erf GELU."""
import math
import numpy as np
from scipy.special import erf

x = np.linspace(-8, 8, 400001).astype(np.float32)
ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / math.sqrt(2)))
h = np.float16
xh = x.astype(h)
a = np.abs(xh)
c = [h(v) for v in (-0.024772998623334343, -0.49926576060257244, -1.1287482669759885, -1.0036805164077327)]
t = (a.astype(np.float32) * np.float32(c[0]) + np.float32(c[1])).astype(h)       # v_pk_fma_f16: one rounding per fma
t = (a.astype(np.float32) * t.astype(np.float32) + np.float32(c[2])).astype(h)
t = (a.astype(np.float32) * t.astype(np.float32) + np.float32(c[3])).astype(h)
e = np.exp2(t.astype(np.float32)).astype(h)                                       # v_exp_f16
g = (np.maximum(xh, h(0)).astype(np.float32) - a.astype(np.float32) * e.astype(np.float32)).astype(h)
err_pk = np.abs(g.astype(np.float64) - ref)


def q5(v):
    t5 = v * np.float32(-0.0004726569791655389) + np.float32(0.007079169600613161)
    for k in (-0.05181158088529847, -0.46001256950698816, -1.1507770495088248, -1.000039487932206):
        t5 = v * t5 + np.float32(k)
    return t5


a32 = np.abs(x)
g5 = (np.maximum(x, 0) - a32 * np.exp2(q5(a32))).astype(h)
err5 = np.abs(g5.astype(np.float64) - ref)
m = np.abs(x) <= 4
print("packed-f16 degree 3     : max abs err %.2e, rms %.2e" % (err_pk[m].max(), math.sqrt((err_pk[m] ** 2).mean())))
print("shipped (fp32 degree 5) : max abs err %.2e, rms %.2e  (= the f16 rounding of the operand)" % (err5[m].max(), math.sqrt((err5[m] ** 2).mean())))
