"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's conversion of a measured spectrum to linear RGB
(the data/ior/<material>.{eta,k}.spd presets of the conductor models), in numpy single-precision scalars so that
every rounding happens where the reference's `Float` arithmetic rounds. Nothing outside tests/ may import this.

  src/libcore/spectrum.cpp:172-184   Spectrum::fromContinuousSpectrum, three-sample build
  src/libcore/spectrum.cpp:546-568   ContinuousSpectrum::average
  src/libcore/spectrum.cpp:575-602   InterpolatedSpectrum(path)
  src/libcore/spectrum.cpp:650-714   InterpolatedSpectrum::average, ::eval
  src/libcore/quad.cpp:305-409       GaussLobattoIntegrator::integrate / calculateAbsTolerance / adaptiveGaussLobattoStep
  src/libcore/spectrum.cpp:630-648   InterpolatedSpectrum::zeroExtend
  src/libcore/spectrum.cpp:483-495   BlackBodySpectrum::eval
  src/librender/scenehandler.cpp:534-620   <blackbody> and <spectrum> values of a scene file

Parity unpinned: the reference holds no expected RGB triple for any of its spectra (no test, no fixture, no log), so
this restatement is checked only against an independent dense Riemann sum of the same integrand.
"""
import bisect
import math
import os
import re

import numpy as np

F = np.float32
_EPS = F(np.finfo(np.float32).eps)


def cie_tables():
    """The matching-function rows the library compiles in ({lambda, x, y, z} per line)."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mitsuba_amd", "csrc", "cie1931.inc")
    rows = [[float(t.rstrip("f")) for t in re.findall(r"[-0-9.eE]+f", line)] for line in open(path) if line.startswith("{")]
    return np.asarray(rows, np.float32)


class Spd:
    def __init__(self, lam, val):
        self.lam = [F(x) for x in lam]
        self.val = [F(x) for x in val]

    @staticmethod
    def load(path):
        lam, val = [], []
        for line in open(path):
            line = line.strip()
            if not line or line[0] == "#":
                continue
            t = line.split()
            try:
                lam.append(float(t[0])), val.append(float(t[1]))
            except (ValueError, IndexError):
                break
        return Spd(lam, val)

    def eval(self, l):
        """spectrum.cpp:688-714 -- note the order in which :706 hands the two values to lerp"""
        if len(self.lam) < 2 or l < self.lam[0] or l > self.lam[-1]:
            return F(0)
        i1, i2 = bisect.bisect_left(self.lam, l), bisect.bisect_right(self.lam, l)
        if i1 != i2:
            return self.val[i1]
        a, b, fa, fb = self.lam[i1 - 1], self.lam[i1], self.val[i1 - 1], self.val[i1]
        t = (l - a) / (b - a)
        return (F(1) - t) * fb + t * fa

    def average(self, lo, hi):
        """spectrum.cpp:650-686"""
        lo, hi = F(lo), F(hi)
        start, end = max(lo, self.lam[0]), min(hi, self.lam[-1])
        if len(self.lam) < 2 or end <= start:
            return F(0)
        e = max(bisect.bisect_left(self.lam, start), 1) - 1
        result = F(0)
        while e + 1 < len(self.lam) and end >= self.lam[e]:
            a, b = self.lam[e], self.lam[e + 1]
            ca, cb, fa, fb = max(a, start), min(b, end), self.val[e], self.val[e + 1]
            inv = F(1) / (b - a)
            if cb > ca:
                ta, tb = (ca - a) * inv, (cb - a) * inv
                result = result + F(0.5) * (((F(1) - ta) * fa + ta * fb) + ((F(1) - tb) * fa + tb * fb)) * (cb - ca)
            e += 1
        return result / (hi - lo)


    def zero_extended(self):
        """spectrum.cpp:630-648"""
        assert len(self.lam) >= 2
        spacing = F(0)
        for i in range(len(self.lam) - 1):
            spacing = spacing + (self.lam[i + 1] - self.lam[i])
        spacing = spacing / F(len(self.lam) - 1)
        lam, val = list(self.lam), list(self.val)
        if val[0] != 0:
            lam.insert(0, lam[0] - spacing), val.insert(0, F(0))
        if val[-1] != 0:
            lam.append(lam[-1] + spacing), val.append(F(0))
        return Spd(lam, val)


class BlackBody:
    """spectrum.cpp:483-495: Planck's law in double precision, W m^-2 nm^-1 sr^-1, rounded to single on return"""

    def __init__(self, temperature):
        self.temperature = F(temperature)

    def eval(self, l):
        c, k, h = 299792458.0, 1.3806488e-23, 6.62606957e-34
        lam = float(F(l)) * 1e-9
        x = (h / k) * c / (lam * float(self.temperature))
        e = math.exp(x) if x < 709 else math.inf
        return F((2 * h * c * c) * lam ** -5.0 / ((e - 1.0) * 1e9))


def scene_value(spectrum, scale=1.0):
    """What SceneHandler stores for a <spectrum> / <blackbody> element (scenehandler.cpp:534-620): conversion, clamp at
    zero, scale. `spectrum` is a BlackBody or an Spd (which is zero-extended first)."""
    if isinstance(spectrum, Spd):
        spectrum = spectrum.zero_extended()
    return np.maximum(to_rgb(spectrum), F(0)) * F(scale)


_ALPHA, _BETA = F(math.sqrt(2.0 / 3.0)), F(1.0 / math.sqrt(5.0))
_X1, _X2, _X3 = F(0.94288241569547971906), F(0.64185334234578130578), F(0.23638319966214988028)


class _Lobatto:
    """quad.cpp:305-409 with the settings of spectrum.cpp:547 (10000 evaluations, both tolerances 1e-4, no convergence
    estimate)"""

    def __init__(self, f):
        self.f, self.evals = f, 0

    def tolerance(self, a, b):
        f = self.f
        m, h = (a + b) / F(2), (b - a) / F(2)
        y1, y3, y5, y7, y9, y11, y13 = f(a), f(m - _ALPHA * h), f(m - _BETA * h), f(m), f(m + _BETA * h), f(m + _ALPHA * h), f(b)
        acc = h * (F(0.0158271919734801831) * (y1 + y13)
                   + F(0.0942738402188500455) * (f(m - _X1 * h) + f(m + _X1 * h))
                   + F(0.1550719873365853963) * (y3 + y11)
                   + F(0.1888215739601824544) * (f(m - _X2 * h) + f(m + _X2 * h))
                   + F(0.1997734052268585268) * (y5 + y9)
                   + F(0.2249264653333395270) * (f(m - _X3 * h) + f(m + _X3 * h))
                   + F(0.2426110719014077338) * y7)
        self.evals += 13
        result = F(np.inf)
        if acc != 0:
            result = acc * max(F(1e-4), _EPS) / _EPS
        return min(result, F(1e-4) / _EPS)

    def step(self, a, b, fa, fb, acc):
        f = self.f
        h, m = (b - a) / F(2), (a + b) / F(2)
        mll, ml, mr, mrr = m - _ALPHA * h, m - _BETA * h, m + _BETA * h, m + _ALPHA * h
        fmll, fml, fm, fmr, fmrr = f(mll), f(ml), f(m), f(mr), f(mrr)
        i2 = (h / F(6)) * (fa + fb + F(5) * (fml + fmr))
        i1 = (h / F(1470)) * (F(77) * (fa + fb) + F(432) * (fmll + fmrr) + F(625) * (fml + fmr) + F(672) * fm)
        self.evals += 5
        if self.evals >= 10000:
            return i1
        dist = acc + (i1 - i2)
        if dist == acc or mll <= a or b <= mrr:
            return i1
        s = self.step
        return (s(a, mll, fa, fmll, acc) + s(mll, ml, fmll, fml, acc) + s(ml, m, fml, fm, acc)
                + s(m, mr, fm, fmr, acc) + s(mr, mrr, fmr, fmrr, acc) + s(mrr, b, fmrr, fb, acc))

    def integrate(self, a, b):
        if a == b:
            return F(0)
        self.evals = 0
        acc = self.tolerance(a, b)
        self.evals += 2
        return self.step(a, b, self.f(a), self.f(b), acc)


def product_average(s1, s2, lo, hi):
    """spectrum.cpp:546-568 applied to a ProductSpectrum (:503-505)"""
    lo, hi = F(lo), F(hi)
    if hi <= lo:
        return F(0)
    q = _Lobatto(lambda l: s1.eval(l) * s2.eval(l))
    n = max(1, int(math.ceil(float((hi - lo) / F(50)))))
    step, pos, integral = (hi - lo) / F(n), lo, F(0)
    for _ in range(n):
        integral = integral + q.integrate(pos, pos + step)
        pos = pos + step
    return integral / (hi - lo)


def to_rgb(spd):
    """spectrum.cpp:172-184 followed by fromXYZ (:222-227)"""
    with np.errstate(over="ignore"):
        cie = cie_tables()
        lo, hi = cie[0, 0], cie[-1, 0]
        cx, cy, cz = (Spd(cie[:, 0], cie[:, c]) for c in (1, 2, 3))
        X, Y, Z = (product_average(spd, c, lo, hi) for c in (cx, cy, cz))
        norm = F(1) / cy.average(lo, hi)
        X, Y, Z = X * norm, Y * norm, Z * norm
        return np.asarray([F(3.240479) * X + F(-1.537150) * Y + F(-0.498535) * Z,
                           F(-0.969256) * X + F(1.875991) * Y + F(0.041556) * Z,
                           F(0.055648) * X + F(-0.204043) * Y + F(1.057311) * Z], np.float32)


def to_rgb_dense(spd, samples_per_nm=200):
    """The same quantity by a midpoint Riemann sum in double precision (no adaptive quadrature): what the adaptive
    rule converges to, up to its 1e-4 tolerances."""
    cie = cie_tables().astype(np.float64)

    def reversed_lerp(xs, vs, l):
        i = np.clip(np.searchsorted(xs, l, side="left"), 1, len(xs) - 1)
        t = (l - xs[i - 1]) / (xs[i] - xs[i - 1])
        out = (1 - t) * vs[i] + t * vs[i - 1]
        return np.where((l < xs[0]) | (l > xs[-1]), 0.0, out)

    n = int((cie[-1, 0] - cie[0, 0]) * samples_per_nm)
    l = cie[0, 0] + (np.arange(n) + 0.5) / samples_per_nm
    if isinstance(spd, BlackBody):
        c, k, h = 299792458.0, 1.3806488e-23, 6.62606957e-34
        s = (2 * h * c * c) * (l * 1e-9) ** -5.0 / (np.expm1((h / k) * c / (l * 1e-9 * float(spd.temperature))) * 1e9)
    else:
        s = reversed_lerp(np.asarray(spd.lam, np.float64), np.asarray(spd.val, np.float64), l)
    xyz = [np.mean(s * reversed_lerp(cie[:, 0], cie[:, c], l)) for c in (1, 2, 3)]
    ynorm = np.trapezoid(cie[:, 2], cie[:, 0]) / (cie[-1, 0] - cie[0, 0])
    X, Y, Z = (v / ynorm for v in xyz)
    return np.asarray([3.240479 * X - 1.537150 * Y - 0.498535 * Z, -0.969256 * X + 1.875991 * Y + 0.041556 * Z,
                       0.055648 * X - 0.204043 * Y + 1.057311 * Z])
