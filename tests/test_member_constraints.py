"""Constraints that differ between members: hx_setvar_dated_members on
CO2_constrain / NBP_constrain / tas_constrain / RF_tot_constrain / CH4_constrain, a different
series (or none: NaN) for every member, each member against the oracle reading a scenario that
carries that member's constraint -- the reference's per-run constraint
(simpleNbox-runtime.cpp:347-383,567-603; temperature_component.cpp:510-525;
forcing_component.cpp:498-505; ch4_component.cpp:156-157), for all members in one launch."""
import numpy as np
import pytest

import hector_amd
from conftest import SCENARIO, edited_pack

Y0, Y1 = 1745, 2300
REL_CO2, ABS_T = 2e-8, 2e-8


def member_constraints_vs_oracle(lib, tmp_path, run_to=2150, **kw):
    import oracle_binding
    n = 6
    S = np.array([2.5, 3.0, 3.5, 4.0, 4.5, 5.0])
    c = hector_amd.Core(SCENARIO, n, lib_path=lib, **kw)
    c.setvar("S", S, "degC")
    c.set_outputs(["CO2_concentration", "global_tas", "RF_tot", "NBP", "CH4_concentration", "timesteps"])
    c.run(run_to)
    base = {v: c.fetchvars(v, (Y0, run_to)).copy() for v in ("CO2_concentration", "global_tas", "RF_tot",
                                                              "CH4_concentration")}
    nan = np.nan
    yrs = np.arange(1900, 2051)
    k = yrs - Y0
    # member 0: no constraint at all; 1: CO2 path x 1.2; 2: NBP = 0.5 in 1950-2000; 3: tas path
    # + 0.3 K in 1980-2050; 4: RF_tot x 0.8 in 1900-2000; 5: CO2 x 0.9 AND CH4 x 1.5
    co2 = np.full((yrs.size, n), nan); co2[:, 1] = base["CO2_concentration"][k, 1] * 1.2
    co2[:, 5] = base["CO2_concentration"][k, 5] * 0.9
    nbp = np.full((yrs.size, n), nan); nbp[(yrs >= 1950) & (yrs <= 2000), 2] = 0.5
    tas = np.full((yrs.size, n), nan); m = yrs >= 1980; tas[m, 3] = base["global_tas"][k[m], 3] + 0.3
    ftot = np.full((yrs.size, n), nan); m = yrs <= 2000; ftot[m, 4] = base["RF_tot"][k[m], 4] * 0.8
    ch4 = np.full((yrs.size, n), nan); ch4[:, 5] = base["CH4_concentration"][k, 5] * 1.5
    sets = [("CO2_constrain", "simpleNbox", co2, "ppmv CO2"), ("NBP_constrain", "simpleNbox", nbp, "Pg C/yr"),
            ("tas_constrain", "temperature", tas, "degC"), ("RF_tot_constrain", "forcing", ftot, "W/m2"),
            ("CH4_constrain", "CH4", ch4, "ppbv CH4")]
    for name, _, vals, unit in sets:
        c.setvar_dated_members(name, yrs, vals, unit)
    c.reset(Y0); c.run(run_to)
    assert (c.status() == 0).all()
    got = {v: c.fetchvars(v, (Y0, run_to)) for v in ("CO2_concentration", "global_tas", "RF_tot", "NBP")}
    ts = c.fetchvars("timesteps", (Y0 + 1, run_to))
    # member 0 is untouched (the extended kernel integrates one more solver variable than the plain
    # one that produced `base`, so not bit for bit); the constrained members follow their constraints
    for v in ("CO2_concentration", "global_tas", "RF_tot"):
        assert np.abs(got[v][:, 0] - base[v][:, 0]).max() < 1e-9 * max(1.0, np.abs(base[v][:, 0]).max()), v
    assert np.allclose(got["CO2_concentration"][k, 1], co2[:, 1], rtol=1.5e-8)
    assert np.allclose(got["NBP"][1950 - Y0:2001 - Y0, 2], 0.5, atol=1e-9)
    # fetchvars returns what was set
    back = c.fetchvars("CO2_constrain", (1900, 2050))
    assert np.array_equal(np.isnan(back), np.isnan(co2)) and np.allclose(back[:, 1], co2[:, 1])
    nk = run_to - Y0 + 1
    for i in range(n):
        path, first = None, True
        for name, sec, vals, _ in sets:
            ok = ~np.isnan(vals[:, i])
            if ok.any():
                yy, vv = yrs[ok], vals[ok, i]
                if name == "RF_tot_constrain":
                    # the reference's Ftot_constrain tseries extrapolates flat before its first
                    # date (forcing_component.cpp:498 tests only the LAST date): the oracle reads
                    # dense series, so its pack carries that back-fill
                    yy = np.concatenate([np.arange(Y0, yy[0]), yy])
                    vv = np.concatenate([np.full(yy.size - vv.size, vv[0]), vv])
                path = edited_pack(tmp_path / ("m%d.hxs" % i), sec, name, yy, vv,
                                   base=None if first else path)
                first = False
        o = oracle_binding.Oracle(path or SCENARIO)
        p = o.default_params(); p.S = S[i]
        r, err, _ = o.run(p, run_to=run_to)
        assert err == 0
        ref = r["CO2_concentration"][:nk]
        assert (np.abs(got["CO2_concentration"][:, i] - ref) / ref).max() < REL_CO2, i
        assert np.abs(got["global_tas"][:, i] - r["global_tas"][:nk]).max() < ABS_T, i
        assert np.abs(got["RF_tot"][:, i] - r["RF_tot"][:nk]).max() < ABS_T, i
        assert np.array_equal(ts[:, i], r["timesteps"][1:nk]), i


def test_member_constraints_vs_oracle(emul_lib, tmp_path):
    member_constraints_vs_oracle(emul_lib, tmp_path, allow_emulation=True)


def test_member_constraint_errors(emul_lib):
    c = hector_amd.Core(SCENARIO, 2, lib_path=emul_lib, allow_emulation=True)
    with pytest.raises(hector_amd.HectorAmdError, match="[Uu]nits"):
        c.setvar_dated_members("CO2_constrain", [1900], np.ones((1, 2)), "Pg C")
    with pytest.raises(hector_amd.HectorAmdError, match="startDate"):
        c.setvar_dated_members("CH4_constrain", [1745], np.ones((1, 2)), "ppbv CH4")
    with pytest.raises(hector_amd.HectorAmdError, match="not supported"):
        c.setvar_dated_members("SV", [1900], np.ones((1, 2)))


@pytest.mark.gpu
def test_member_constraints_vs_oracle_on_gpu(hip_lib, tmp_path):
    member_constraints_vs_oracle(hip_lib, tmp_path, run_to=2300, device=0)
