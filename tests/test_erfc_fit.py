"""The permafrost curve's erfc as one polynomial range (hx_erfc_fit.inc, tools/make_erfc_fit.py,
hx_frozen_fraction_batch in hx_dev_math.h): the committed table reproduces erfc -- and with it the
reference's frozen fraction 1 - cdf(lognormal(mu, sigma), Tb), simpleNbox-runtime.cpp:1006-1034 --
to a few 1e-16 absolute over every argument the curve can take."""
import json
import os
import re

import numpy as np
from scipy.special import erfc

from conftest import ROOT


def _table():
    txt = open(os.path.join(ROOT, "hector_amd", "csrc", "hx_erfc_fit.inc")).read()
    deg = int(re.search(r"HX_ERFC_FIT_DEGREE (\d+)", txt).group(1))
    body = txt[txt.index("{", txt.index("HX_ERFC_FIT_COEFFS")) + 1:txt.rindex("}")]
    vals = [float.fromhex(x.strip()) for x in body.replace("\\", " ").replace("\n", " ").split(",") if x.strip()]
    assert len(vals) == deg + 1
    return deg, np.array(vals)


def _frozen_fraction(d, c):
    """hx_frozen_fraction_batch, operation by operation (without the fused multiply-adds)."""
    a = np.abs(d)
    t = 2.0 / (2.0 + a)
    u = (t + t) - 1.0
    p = np.full_like(d, c[0])
    for cj in c[1:]:
        p = p * u + cj
    half = 0.5 * (t * np.exp(np.maximum(p - a * a, -746.0)))
    return np.where(d > 0.0, half, 1.0 - half)


def test_table_reproduces_erfc():
    deg, c = _table()
    rep = json.load(open(os.path.join(ROOT, "profiles", "erfc_fit_report.json")))
    assert rep["degree"] == deg == 27 and rep["max_abs_error_double_vs_50_digits"] < 1e-15
    assert np.abs(c).max() < 0.7          # Horner's rule is well conditioned
    d = np.concatenate([np.linspace(-27.0, 27.0, 200001), [-1e3, -600.0, 0.0, 1e-300, -1e-300, 40.0]])
    want = 1.0 - erfc(-d) / 2.0
    got = _frozen_fraction(d, c)
    assert np.abs(got - want).max() < 1.5e-15
    assert got[np.argmin(d)] == 1.0 and got[np.argmax(d)] == 0.0   # far below 0 degC / far above


def test_frozen_fraction_of_the_default_biome():
    """mu = 1.67, sigma = 0.986 (inst/input/hector_*.ini): half of the permafrost is thawed at
    exp(mu) = 5.3 K of biome warming; a biome at or below 0 K stays frozen (the kernel's select)."""
    _, c = _table()
    mu, sigma = 1.67, 0.986
    Tb = np.array([0.05, 0.5, 1.0, 3.0, np.exp(mu), 8.0, 15.0])
    d = (np.log(Tb) - mu) / (sigma * np.sqrt(2.0))
    ff = _frozen_fraction(d, c)
    assert abs(ff[4] - 0.5) < 1e-15
    assert (np.diff(ff) < 0).all() and ff[0] > 0.9999 and ff[-1] < 0.2


def test_committed_table_is_what_the_generator_produces(tmp_path):
    import importlib.util
    import pytest
    pytest.importorskip("mpmath")
    spec = importlib.util.spec_from_file_location("make_erfc_fit", os.path.join(ROOT, "tools", "make_erfc_fit.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gen.main(str(tmp_path / "fit.inc"), str(tmp_path / "report.json"))
    assert open(tmp_path / "fit.inc").read() == open(os.path.join(ROOT, "hector_amd", "csrc", "hx_erfc_fit.inc")).read()
