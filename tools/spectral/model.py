#!/usr/bin/env python3
"""numpy model of the spectral lower-bound filter (Z60 = Z4 x Z15 by CRT: DFT along Z15, direct along Z4).
Checks the algebra against the direct circular correlation and measures the fp16 pipeline's error."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from navtech_radar_slam_amd import synth

R, C = 20, 60
c_of = np.zeros((4, 15), dtype=int)
for c in range(60):
    c_of[c % 4, c % 15] = c
W = np.exp(2j * np.pi / 15)

def normalise(d):
    d = d.reshape(R, C).astype(np.float64)
    n = np.sqrt((d * d).sum(0))
    out = np.where(n > 0, d / np.where(n > 0, n, 1), 0.0)
    return out, (n > 0)

def direct_S(q, e):
    return np.array([(np.roll(q, -k, axis=1) * e).sum() for k in range(60)])   # S_k = sum q[:, c+k] e[:, c]

def spectra(x):
    """X[f][a][r] complex, f = 0..7 (Z15 frequencies, the rest are conjugates)"""
    g = x[:, c_of]                      # (R, 4, 15)
    b = np.arange(15)
    F = np.exp(-2j * np.pi * np.outer(np.arange(8), b) / 15)      # (8, 15)
    return np.einsum('fb,rab->far', F, g)

def spectral_S(q, e, half=False):
    Q, E = spectra(q), spectra(e)
    if half:
        rnd = lambda z: z.real.astype(np.float16).astype(np.float64) + 1j * z.imag.astype(np.float16).astype(np.float64)
        Q, E = rnd(Q), rnd(E)
    S = np.zeros(60)
    Cs = np.zeros((4, 8), dtype=complex)
    for k4 in range(4):
        Cs[k4] = np.einsum('far,far->f', np.roll(Q, -k4, axis=1), np.conj(E))
    if half:
        Cs32 = Cs.astype(np.complex64)
        re = Cs32.real.astype(np.float16).astype(np.float64); im = Cs32.imag.astype(np.float16).astype(np.float64)
        re[:, 0] = Cs32.real[:, 0]       # DC: hi + lo split, ~exact
        Cs = re + 1j * im
    for k4 in range(4):
        for k15 in range(15):
            ph = W ** (np.arange(8) * k15)
            wr, wi = ph.real * 2 / 16, ph.imag * 2 / 16
            wr[0] = 1 / 16
            if half:
                wr = wr.astype(np.float16).astype(np.float64); wi = wi.astype(np.float16).astype(np.float64)
            val = (Cs[k4].real * wr - Cs[k4].imag * wi).sum() * 16 / 15
            k = [kk for kk in range(60) if kk % 4 == k4 and kk % 15 == k15][0]
            S[k] = val
    return S

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for binary in (True, False):
        d = synth.random_descriptors(5, 40, binary=binary)
        worst = worst_h = 0.0
        for t in range(200):
            i, j = rng.integers(0, 40, 2)
            q, mq = normalise(d[i]); e, me = normalise(d[j])
            S0 = direct_S(q, e)
            S1 = spectral_S(q, e)
            S2 = spectral_S(q, e, half=True)
            worst = max(worst, np.abs(S1 - S0).max())
            worst_h = max(worst_h, np.abs(S2 - S0).max() / np.sqrt(mq.sum() * me.sum()))
        print(f"binary={binary}: algebra err {worst:.2e}; fp16 pipeline err / sqrt(nq ne) = {worst_h:.2e}")
