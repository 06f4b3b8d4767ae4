#!/usr/bin/env python3
"""Fits the coefficients of bf16 mode's GELU approximation (csrc/m2t_common.h M2T_GELU_A/B/C):
    Phi(t) ~ sigma(t (a + b t^2 + c t^4)),  gelu = t Phi,  gelu' = s + t s (1 - s) (a + 3 b t^2 + 5 c t^4)
minimising max(|gelu - exact|, |gelu' - exact|) over |t| <= 9 (Nelder-Mead from the tanh-GELU constants)."""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf

t = np.linspace(-9, 9, 36001)
Phi = 0.5 * (1 + erf(t / np.sqrt(2)))
phi = np.exp(-t * t / 2) / np.sqrt(2 * np.pi)
g, gd = t * Phi, Phi + t * phi


def model(p, t):
    t2 = t * t
    u = p[0] + p[1] * t2 + p[2] * t2 * t2
    du = p[0] + 3 * p[1] * t2 + 5 * p[2] * t2 * t2
    s = 1 / (1 + np.exp(-np.clip(t * u, -80, 80)))
    return t * s, s + t * s * (1 - s) * du


def cost(p):
    a, b = model(p, t)
    return max(np.abs(a - g).max(), np.abs(b - gd).max())


r = minimize(cost, [1.5957691, 0.0713548, 0.0], method="Nelder-Mead", options=dict(xatol=1e-10, fatol=1e-12, maxiter=20000))
a, b = model(r.x, t)
print("a, b, c =", r.x, " max |gelu err| %.3e  max |gelu' err| %.3e" % (np.abs(a - g).max(), np.abs(b - gd).max()))
p0 = [1.5957691, 0.0713548, 0.0]
print("tanh-GELU for comparison: %.3e %.3e" % (np.abs(model(p0, t)[0] - g).max(), np.abs(model(p0, t)[1] - gd).max()))
