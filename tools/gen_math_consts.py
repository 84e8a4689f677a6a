#!/usr/bin/env python3
"""Derive the polynomial constants used by include/kabc_math.h.

Run:  python tools/gen_math_consts.py
Prints C hex-float literals.  Nothing here is taken from the reference; the
fits are plain Chebyshev interpolants / exact Taylor rationals computed with
mpmath at 60 digits.
"""
import mpmath as mp

mp.mp.dps = 60


def hexf(x):
    return float(x).hex()


def show(name, coeffs):
    print(f"/* {name} */")
    for i, c in enumerate(coeffs):
        print(f"  {name}[{i}] = {hexf(c)}  /* {mp.nstr(c, 20)} */")


# ---- log: log(1+f) = f - hfsq + s*(hfsq + R(z)),  s=f/(2+f), z=s^2,
#      R(z) = z*(2/3 + 2/5 z + 2/7 z^2 ...).  Fit g(z) = R(z)/z on
#      z in [0, zmax], zmax = ((sqrt2-1)/(sqrt2+1))^2.
smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1)
zmax = smax ** 2 * mp.mpf("1.0001")


def g_log(z):
    if z < mp.mpf("1e-40"):
        return mp.mpf(2) / 3
    s = mp.sqrt(z)
    return (mp.log((1 + s) / (1 - s)) / s - 2) / z


# chebyfit returns coefficients highest power first
c = mp.chebyfit(g_log, [0, zmax], 8)
show("LG", list(reversed(c)))

# ---- exp Taylor 1/k!, k = 2..13
show("EXPT", [1 / mp.factorial(k) for k in range(0, 14)])
# ---- sin / cos Taylor in x^2
show("SINT", [(-1) ** k / mp.factorial(2 * k + 1) for k in range(0, 9)])
show("COST", [(-1) ** k / mp.factorial(2 * k) for k in range(0, 9)])
# ---- constants
ln2 = mp.log(2)
# ln2_hi has its low 21 bits zero so k*ln2_hi is exact for |k| < 2^20
import struct
hi = float(ln2)
b = struct.unpack("<Q", struct.pack("<d", hi))[0] & ~((1 << 21) - 1)
hi = struct.unpack("<d", struct.pack("<Q", b))[0]
print("ln2_hi", hi.hex(), "ln2_lo", hexf(ln2 - mp.mpf(hi)))
print("inv_ln2", hexf(1 / ln2))
pio2 = mp.pi / 2
print("pio2_hi", hexf(pio2), "pio2_lo", hexf(pio2 - mp.mpf(float(pio2))))
print("half_log_2pi", hexf(mp.log(2 * mp.pi) / 2))
print("log_2pi", hexf(mp.log(2 * mp.pi)))
# Stirling coefficients B_{2k} / (2k (2k-1))
show("STIR", [mp.bernoulli(2 * k) / (2 * k * (2 * k - 1)) for k in range(1, 9)])
print("sqrt3", hexf(mp.sqrt(3)), "inv_sqrt3", hexf(1 / mp.sqrt(3)))
