"""CPU check of the algebra behind the spectral filter (csrc/sc_spec.hip): the CRT split Z60 = Z4 x Z15 with a
DFT along Z15 and a direct correlation along Z4 reproduces the 60-shift circular cross-correlation exactly, and
an fp16 emulation of the two MFMA stages stays far inside the kernel's error budget (tools/spectral/model.py is
the numpy model the kernel was written from)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("spectral_model", os.path.join(ROOT, "tools", "spectral", "model.py"))
model = importlib.util.module_from_spec(spec)
spec.loader.exec_module(model)

from navtech_radar_slam_amd import synth  # noqa: E402


def test_crt_index_map_is_a_bijection():
    assert sorted(model.c_of.reshape(-1).tolist()) == list(range(60))
    for c in range(60):
        assert model.c_of[c % 4, c % 15] == c
        assert (45 * (c % 4) + 16 * (c % 15)) % 60 == c          # the closed form used by the kernels


def test_spectral_correlation_equals_direct():
    rng = np.random.default_rng(0)
    for binary in (True, False):
        d = synth.random_descriptors(5, 24, binary=binary)
        for _ in range(40):
            i, j = rng.integers(0, 24, 2)
            q, _ = model.normalise(d[i])
            e, _ = model.normalise(d[j])
            assert np.abs(model.spectral_S(q, e) - model.direct_S(q, e)).max() < 1e-12


def test_fp16_pipeline_error_is_inside_the_budget():
    rng = np.random.default_rng(1)
    worst = 0.0
    for binary in (True, False):
        d = synth.random_descriptors(6, 24, binary=binary)
        for _ in range(60):
            i, j = rng.integers(0, 24, 2)
            q, mq = model.normalise(d[i])
            e, me = model.normalise(d[j])
            err = np.abs(model.spectral_S(q, e, half=True) - model.direct_S(q, e)).max()
            worst = max(worst, err / np.sqrt(mq.sum() * me.sum()))
    assert worst < 2.05e-3 / 4       # kSpecEps is the rigorous worst case; typical errors are ~20x smaller
