"""Reference property tests re-expressed on the oracle (tests/testthat/
test_biome.R:193-283, test_parameters.R, test_atmosphere.R:46-67)."""
import numpy as np
import pytest


def test_identical_biome_split_is_identical_climate(oracle):
    """test_biome.R:193-256: splitting `global` into identical biomes must not
    change the climate (bit-identical in the reference)."""
    base, e0, _ = oracle.run()
    p = oracle.split_equal(oracle.default_params(), 4)
    split, e1, _ = oracle.run(p)
    assert e0 == 0 and e1 == 0
    for v in ("CO2_concentration", "global_tas", "veg_c", "soil_c", "permafrost_c"):
        assert np.array_equal(base[v], split[v]), v


def test_tgav_identity(oracle):
    """test_atmosphere.R:46-67: Tgav = flnd*Tland + (1-flnd)*bsi*SST."""
    o, _, _ = oracle.run()
    flnd, bsi = 0.29, 1.3
    assert np.allclose(o["global_tas"], flnd * o["land_tas"] + (1 - flnd) * bsi * o["sst"],
                       rtol=0, atol=1e-14)


@pytest.mark.parametrize("name,delta,var,sign", [
    ("S", 1.0, "global_tas", +1),           # higher ECS -> warmer
    ("q10", 0.6, "CO2_concentration", +1),  # higher Q10 -> more respiration -> more CO2
    ("beta", 0.2, "CO2_concentration", -1),  # stronger fertilisation -> less CO2
    ("diff", 1.0, "global_tas", -1),        # more ocean heat uptake -> cooler surface
    ("aero", 0.5, "global_tas", -1),        # stronger (negative) aerosol forcing -> cooler
])
def test_directional_parameter_responses(oracle, name, delta, var, sign):
    """test_parameters.R: monotone responses of 2100 values."""
    base, _, _ = oracle.run(run_to=2100)
    p = oracle.default_params()
    if name == "S": p.S += delta
    elif name == "q10": p.q10_rh[0] += delta
    elif name == "beta": p.beta[0] += delta
    elif name == "diff": p.diff += delta
    elif name == "aero": p.aero_scalar += delta
    pert, err, _ = oracle.run(p, run_to=2100)
    assert err == 0
    i = 2100 - 1745
    assert sign * (pert[var][i] - base[var][i]) > 0


def test_heterogeneous_biomes_run_clean(oracle):
    p = oracle.split_equal(oracle.default_params(), 4)
    for b in range(4):
        p.warmingfactor[b] = 1.0 * (1 + 0.5 * b)
        p.q10_rh[b] = 1.2 + 0.4 * b
    o, err, _ = oracle.run(p)
    assert err == 0
    assert 400 < o["CO2_concentration"][-1] < 1200


def test_carbonate_unit_vector(oracle):
    """pCO2 rises with DIC and falls with alkalinity; pH ~ 8 for modern surface water."""
    vol = 3.6e14 * 0.85 * 100
    a = oracle.csys(20.9, 766.0, 2300e-6, vol)
    b = oracle.csys(20.9, 780.0, 2300e-6, vol)
    c = oracle.csys(20.9, 766.0, 2350e-6, vol)
    assert 7.5 < a[1] < 8.6
    assert b[0] > a[0] > c[0]
