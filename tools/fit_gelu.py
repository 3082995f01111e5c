import numpy as np
from scipy.special import erfc, erf
from scipy.optimize import least_squares
SQ = np.sqrt(0.5)
def g(a):  # -log2(erfc(a/sqrt2))/a
    a = np.asarray(a, dtype=np.float64)
    return -np.log2(erfc(a * SQ)) / a
A = 7.0
a = np.linspace(1e-4, A, 20001)
e = erfc(a * SQ)
w = 0.5 * np.log(2) * a * a * e          # d gelu / d g
def gelu_ref(u):
    return 0.5 * u * (1 + erf(u * SQ))
for deg in (4, 5, 6, 7, 8):
    V = np.vander(a, deg + 1, increasing=True)
    ww = w.copy()
    c = None
    for it in range(60):   # Lawson iteration towards minimax
        c, *_ = np.linalg.lstsq(V * ww[:, None], g(a) * ww, rcond=None)
        err = np.abs((V @ c - g(a)) * w)
        ww = ww * (0.5 + err / err.max()) ; ww /= ww.max() / w.max()
    # exact error of the resulting gelu in float64
    arg = -(a * (V @ c))
    ge = 0.5 * a * np.exp2(arg)          # 0.5 a e  (gelu(u) = relu(u) - 0.5 a e)
    err64 = np.abs(ge - 0.5 * a * e)
    print(deg, "max |gelu err| f64 eval:", err64.max(), "at a =", a[err64.argmax()], "lead coeff", c[-1])
    np.save(f"/tmp/fit/c{deg}.npy", c)
