"""N2O and halocarbon parameters that differ between members: the reference
perturbs them per run (n2o_component.cpp:95-130 TN2O0 / N0 / UC_N2O, halocarbon_component.cpp:
118-150 tau / rho / delta); here setvar() takes one value per member and the two components'
recurrences run per member on the device (hx_gas_kernel) ahead of the year loop.  Every member
against the oracle reading a scenario with that member's values."""
import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble
from conftest import SCENARIO, edited_pack

Y0 = 1745
GASES = ["CF4", "HFC134a", "CFC12"]


def gas_params(n, seed=3):
    u = lambda k: ensemble.uniform01(np.arange(n), 40 + k, seed=20260928 + seed)
    tn = 132.0 * (0.8 + 0.4 * u(0))          # TN2O0, INI value 132 years
    taus = {}
    return tn, u


def member_gas_params_vs_oracle(lib, tmp_path, n, run_to=2300, check=None, **kw):
    import oracle_binding
    c = hector_amd.Core(SCENARIO, n, lib_path=lib, **kw)
    tn, u = gas_params(n)
    S = 2.0 + 3.0 * u(9)
    tau0 = {g: c.getvar("tau_" + g)[0] for g in GASES}
    rho0 = c.getvar("rho_CF4")[0]
    taus = {g: tau0[g] * (0.7 + 0.6 * u(1 + k)) for k, g in enumerate(GASES)}
    rho = rho0 * (0.9 + 0.2 * u(5))
    c.setvar("S", S, "degC").setvar("TN2O0", tn, "Years").setvar("rho_CF4", rho, "W/m2/pptv")
    for g in GASES:
        c.setvar("tau_" + g, taus[g], "Years")
    c.set_outputs(["CO2_concentration", "global_tas", "RF_tot", "N2O_concentration", "RF_N2O", "timesteps"])
    c.run(run_to)
    assert (c.status() == 0).all()
    assert np.array_equal(c.getvar("TN2O0"), tn) and np.array_equal(c.getvar("tau_CF4"), taus["CF4"])
    got = {v: c.fetchvars(v, (Y0, run_to)) for v in ("CO2_concentration", "global_tas", "RF_tot",
                                                    "N2O_concentration", "RF_N2O", "RF_CF4",
                                                    "CF4_concentration", "HFC134a_concentration")}
    ts = c.fetchvars("timesteps", (Y0 + 1, run_to))
    nk = run_to - Y0 + 1
    members = range(n) if check is None else check
    for i in members:
        sc = {("N2O", "TN2O0"): tn[i], ("CF4_halocarbon", "rho_CF4"): rho[i]}
        for g in GASES:
            sc[(g + "_halocarbon", "tau")] = taus[g][i]
        o = oracle_binding.Oracle(edited_pack(tmp_path / ("g%d.hxs" % i), None, None, [], [], scalars=sc))
        p = o.default_params(); p.S = S[i]
        r, err, _ = o.run(p, run_to=run_to)
        assert err == 0
        ref = r["CO2_concentration"][:nk]
        assert (np.abs(got["CO2_concentration"][:, i] - ref) / ref).max() < 2e-8, i
        assert np.abs(got["global_tas"][:, i] - r["global_tas"][:nk]).max() < 2e-8, i
        assert np.abs(got["RF_tot"][:, i] - r["RF_tot"][:nk]).max() < 2e-8, i
        n2o = r["N2O_concentration"][:nk]
        assert (np.abs(got["N2O_concentration"][:, i] - n2o) / n2o).max() < 1e-12, i
        assert np.abs(got["RF_N2O"][:, i] - r["RF_N2O"][:nk]).max() < 1e-10, i
        assert np.array_equal(ts[:, i], r["timesteps"][1:nk]), i
    # the perturbations matter: members differ in N2O and in the CF4 forcing
    assert np.ptp(got["N2O_concentration"][-1]) > 5.0 and np.ptp(got["RF_CF4"][-1]) > 1e-4
    assert np.ptp(got["HFC134a_concentration"][2100 - Y0]) > 1.0
    # back to one value for every member: the shared host path again, same as a fresh core
    c.setvar("TN2O0", 132.0, "Years").setvar("rho_CF4", rho0, "W/m2/pptv")
    for g in GASES:
        c.setvar("tau_" + g, tau0[g], "Years")
    c.run(run_to)
    f = hector_amd.Core(SCENARIO, n, lib_path=lib, **kw)
    f.setvar("S", S, "degC"); f.set_outputs(["CO2_concentration", "global_tas", "N2O_concentration"]); f.run(run_to)
    for v in ("CO2_concentration", "global_tas", "N2O_concentration"):
        assert np.abs(c.fetchvars(v, (Y0, run_to)) - f.fetchvars(v, (Y0, run_to))).max() < 1e-9, v


def test_member_gas_params_vs_oracle(emul_lib, tmp_path):
    member_gas_params_vs_oracle(emul_lib, tmp_path, 8, run_to=2200, allow_emulation=True)


def test_shared_gas_params_one_value_only_where_not_supported(emul_lib):
    c = hector_amd.Core(SCENARIO, 3, lib_path=emul_lib, allow_emulation=True)
    with pytest.raises(hector_amd.HectorAmdError, match="member-independent"):
        c.setvar("rho_bc", [1.0, 2.0, 3.0])
    with pytest.raises(hector_amd.HectorAmdError, match="[Uu]nits"):
        c.setvar("TN2O0", [100.0, 110.0, 120.0], "ppbv N2O")


@pytest.mark.gpu
def test_member_gas_params_256_members_vs_oracle_on_gpu(hip_lib, tmp_path):
    """VERDICT r1 item 7 'Done': a 256-member ensemble perturbing TN2O0 and three halocarbon
    lifetimes (and one radiative efficiency), every member against the oracle."""
    member_gas_params_vs_oracle(hip_lib, tmp_path, 256, device=0)
