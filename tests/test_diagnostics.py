"""Every variable the reference's output stream writes (csv_outputstream_visitor.cpp:126-365):
recorded by the run kernel, derived on the device from recorded outputs, or answered on the
host from the shared gas cycles -- all against the oracle.  Host emulation here; the GPU suite
repeats it through the HIP library."""
import numpy as np
import pytest

import hector_amd
from conftest import SCENARIO

Y0, Y1 = 1745, 2300

# capability -> (first year compared, absolute tolerance scale)
KERNEL_VARS = ["NPP", "RH", "rh_det", "rh_soil", "rh_ch4", "f_frozen", "atmos_c_residual", "gmst",
               "heatflux_mixed", "heatflux_interior", "heatflux", "HL_ocean_uptake",
               "LL_ocean_uptake", "ocean_uptake", "HL_ocean_c", "LL_ocean_c", "IO_ocean_c",
               "DO_ocean_c", "HL_downwelling", "HL_PCO2", "LL_PCO2", "HL_pH", "LL_pH", "TAU_OH",
               "O3_concentration", "CH4_concentration", "ocean_timesteps"]
DERIVED_VARS = ["HL_sst", "LL_sst", "HL_DIC", "LL_DIC", "HL_CO3", "LL_CO3", "HL_OmegaAr",
                "LL_OmegaAr", "HL_OmegaCa", "LL_OmegaCa", "HL_Revelle", "LL_Revelle", "ocean_tas",
                "RF_N2O", "RF_CH4", "RF_H2O_strat", "RF_O3_trop", "slr", "sl_rc", "slr_no_ice",
                "sl_rc_no_ice"]
HOST_VARS = ["RF_BC", "RF_OC", "RF_SO2", "RF_NH3", "RF_aci", "RF_vol", "RF_albedo", "RF_misc",
             "N2O_concentration"]
ORACLE_NAME = {"ocean_timesteps": "timesteps"}
TOL = {}   # per-variable exceptions to the 2e-8 every diagnostic is held to (none needed)


def check_all_diagnostics(lib, oracle, S, q10, aero, vol, **kw):
    n = len(S)
    c = hector_amd.Core(SCENARIO, n, lib_path=lib, **kw)
    c.setvar("S", S, "degC").setvar("q10_rh", q10).setvar("aero_scalar", aero).setvar("vol_scalar", vol)
    c.set_outputs(KERNEL_VARS + DERIVED_VARS + ["RF_tot", "RF_CO2", "global_tas"])
    c.run(Y1)
    assert (c.status() == 0).all()
    got = {v: c.fetchvars(v, (Y0 + 1, Y1)) for v in KERNEL_VARS + DERIVED_VARS + HOST_VARS}
    rf_sum = c.fetchvars("RF_CO2", (Y0 + 1, Y1)).copy()
    for v in ["RF_N2O", "RF_CH4", "RF_H2O_strat", "RF_O3_trop", "RF_BC", "RF_OC", "RF_SO2",
              "RF_NH3", "RF_aci", "RF_vol", "RF_albedo", "RF_misc"]:
        rf_sum += got[v]
    halo = sorted(h for h in c.halocarbons())
    rf_halo = sum(c.fetchvars("RF_" + h, (Y0 + 1, Y1)) for h in halo)
    # the reference's RF_tot is the sum of its parts (forcing_component.cpp:489-492)
    assert np.abs(rf_sum + rf_halo - c.fetchvars("RF_tot", (Y0 + 1, Y1))).max() < 1e-12
    for i in range(n):
        p = oracle.default_params()
        p.S = S[i]; p.q10_rh[0] = q10[i]; p.aero_scalar = aero[i]; p.vol_scalar = vol[i]
        r, err, _ = oracle.run(p)
        assert err == 0
        for v in KERNEL_VARS + DERIVED_VARS + HOST_VARS:
            ref = r[ORACLE_NAME.get(v, v)][1:]
            scale = max(1.0, np.abs(ref).max())
            assert np.abs(got[v][:, i] - ref).max() < TOL.get(v, 2e-8) * scale, (v, i)
        assert np.abs(rf_halo[:, i] - r["RF_halocarbons"][1:]).max() < 1e-12
    return c


def test_all_output_stream_variables_vs_oracle(emul_lib, oracle):
    check_all_diagnostics(emul_lib, oracle, np.array([3.0, 2.1, 5.2]), np.array([2.2, 1.4, 2.8]),
                          np.array([1.0, 0.6, 1.3]), np.array([1.0, 1.2, 0.8]),
                          allow_emulation=True)


def test_diagnostics_do_not_change_the_run(emul_lib):
    a = hector_amd.Core(SCENARIO, 2, lib_path=emul_lib, allow_emulation=True)
    a.set_outputs(["CO2_concentration", "global_tas"]); a.run(2100)
    b = hector_amd.Core(SCENARIO, 2, lib_path=emul_lib, allow_emulation=True)
    b.set_outputs(KERNEL_VARS + DERIVED_VARS + ["CO2_concentration", "global_tas"]); b.run(2100)
    # two instantiations of the run kernel (in the extended one the thawed-permafrost pool is a
    # Runge-Kutta variable instead of an exact advance): the same run to the last few ulps
    for v in ("CO2_concentration", "global_tas"):
        x, y = a.fetchvars(v, (Y0, 2100)), b.fetchvars(v, (Y0, 2100))
        assert np.abs(x - y).max() <= 1e-12 * np.abs(x).max()
    with pytest.raises(hector_amd.HectorAmdError):
        a.fetchvars("HL_CO3", (2000, 2001))      # needs outputs that were not recorded
