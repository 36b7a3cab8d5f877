"""ctypes binding of the CPU oracle (oracle/libhector_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing in the product path imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root
ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(ORACLE_DIR, "libhector_oracle.so")
MAXB = 32

VARS = ["CO2_concentration", "global_tas", "RF_tot", "RF_CO2", "heatflux", "ocean_c", "HL_pH",
        "atmos_co2", "sst", "permafrost_c", "land_tas", "CH4_concentration", "N2O_concentration",
        "O3_concentration", "veg_c", "detritus_c", "soil_c", "thawedp_c", "earth_c", "timesteps",
        "max_timestep", "solver_dt", "solver_steps", "rhs_evals", "LL_pH", "HL_PCO2", "LL_PCO2",
        "ocean_uptake", "NBP", "RF_CH4", "RF_N2O",
        "NPP", "RH", "rh_det", "rh_soil", "rh_ch4", "f_frozen", "atmos_c_residual",
        "gmst", "heatflux_mixed", "heatflux_interior", "ocean_tas",
        "HL_ocean_uptake", "LL_ocean_uptake", "HL_ocean_c", "LL_ocean_c", "IO_ocean_c",
        "DO_ocean_c", "HL_DIC", "LL_DIC", "HL_downwelling",
        "HL_OmegaAr", "LL_OmegaAr", "HL_OmegaCa", "LL_OmegaCa",
        "HL_sst", "LL_sst", "HL_CO3", "LL_CO3", "HL_Revelle", "LL_Revelle", "TAU_OH",
        "RF_H2O_strat", "RF_O3_trop", "RF_BC", "RF_OC", "RF_SO2", "RF_NH3", "RF_aci",
        "RF_vol", "RF_albedo", "RF_misc", "RF_halocarbons",
        "slr", "sl_rc", "slr_no_ice", "sl_rc_no_ice"] + \
    ["b%d.%s" % (b, v) for b in range(4) for v in
     ("veg_c", "detritus_c", "soil_c", "permafrost_c", "thawedp_c", "NPP", "RH", "rh_ch4", "f_frozen",
      "detritus_tempfert", "soil_tempfert")]


class Params(ctypes.Structure):
    _fields_ = ([("S", ctypes.c_double), ("diff", ctypes.c_double), ("qco2", ctypes.c_double),
                 ("aero_scalar", ctypes.c_double), ("vol_scalar", ctypes.c_double),
                 ("C0", ctypes.c_double), ("nbiome", ctypes.c_int)] +
                [(n, ctypes.c_double * MAXB) for n in
                 "beta q10_rh warmingfactor npp_flux0 veg_c detritus_c soil_c permafrost_c "
                 "f_nppv f_nppd f_litterd rh_ch4_frac pf_mu pf_sigma fpf_static".split()] +
                [(n, ctypes.c_double) for n in
                 "tt tu twi tid preind_surface_c preind_interdeep_c lo_warming_ratio".split()])


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


class Oracle:
    def __init__(self, scenario):
        if not os.path.exists(ORACLE_LIB):
            build()
        lib = ctypes.CDLL(ORACLE_LIB)
        lib.hxo_scenario_load.restype = ctypes.c_void_p
        lib.hxo_scenario_load.argtypes = [ctypes.c_char_p]
        lib.hxo_scenario_free.argtypes = [ctypes.c_void_p]
        lib.hxo_scenario_start.argtypes = [ctypes.c_void_p]
        lib.hxo_scenario_end.argtypes = [ctypes.c_void_p]
        lib.hxo_params_default.argtypes = [ctypes.c_void_p, ctypes.POINTER(Params)]
        lib.hxo_params_split_equal.argtypes = [ctypes.POINTER(Params), ctypes.c_int]
        lib.hxo_run_member.argtypes = [ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_int,
                                       ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        lib.hxo_run_ensemble_ecs_q10.argtypes = [ctypes.c_void_p, ctypes.POINTER(Params),
                                                 ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.hxo_csys.argtypes = [ctypes.c_double] * 4 + [ctypes.c_void_p]
        lib.hxo_doeclim_kernel.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
        self.lib = lib
        self.sc = lib.hxo_scenario_load(scenario.encode())
        if not self.sc:
            raise RuntimeError("oracle: cannot load scenario " + scenario)
        self.start = lib.hxo_scenario_start(self.sc)
        self.end = lib.hxo_scenario_end(self.sc)
        self.ns = self.end - self.start + 1

    def default_params(self):
        p = Params()
        self.lib.hxo_params_default(self.sc, ctypes.byref(p))
        return p

    def set_rounding_noise(self, rel):
        """see hxo_set_rounding_noise (hector_oracle.h)"""
        self.lib.hxo_set_rounding_noise.argtypes = [ctypes.c_double]
        self.lib.hxo_set_rounding_noise(float(rel))

    def split_equal(self, p, n):
        self.lib.hxo_params_split_equal(ctypes.byref(p), n)
        return p

    def run(self, p=None, run_to=None):
        """-> (dict var -> [ns], err, spinup_steps)"""
        p = p or self.default_params()
        out = np.zeros((len(VARS), self.ns))
        steps = ctypes.c_int(0)
        err = self.lib.hxo_run_member(self.sc, ctypes.byref(p), run_to or self.end,
                                      out.ctypes.data, ctypes.byref(steps))
        return {v: out[i] for i, v in enumerate(VARS)}, err, steps.value

    def run_spinup(self, p=None):
        """The spinup alone -> (dict var -> [steps]: what the output stream sees after every
        spinup step, its spinup = 1 rows; err; steps)"""
        p = p or self.default_params()
        self.lib.hxo_scenario_max_spinup.argtypes = [ctypes.c_void_p]
        mx = self.lib.hxo_scenario_max_spinup(self.sc)
        out = np.zeros((len(VARS), mx))
        steps = ctypes.c_int(0)
        fn = self.lib.hxo_run_member_spinup
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        err = fn(self.sc, ctypes.byref(p), out.ctypes.data, ctypes.byref(steps))
        return {v: out[i, :steps.value] for i, v in enumerate(VARS)}, err, steps.value

    def run_tracking(self, p, tracking_date, run_to=None):
        """-> (values[ns, TP], fractions[ns, TP, TP], pool names, err); see hector_oracle.h"""
        p = p or self.default_params()
        run_to = run_to or self.end
        tp = 2 + 5 * p.nbiome + 4
        out = np.zeros((len(VARS), self.ns))
        f = np.zeros((self.ns, tp, tp)); v = np.zeros((self.ns, tp))
        dp = ctypes.POINTER(ctypes.c_double)
        fn = self.lib.hxo_run_member_tracking
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_int, ctypes.c_int, dp,
                       ctypes.POINTER(ctypes.c_int), dp, dp]
        err = fn(self.sc, ctypes.byref(p), int(run_to), int(tracking_date), out.ctypes.data_as(dp),
                 None, f.ctypes.data_as(dp), v.ctypes.data_as(dp))
        names = ["atmos_co2", "earth_c"]
        for b in range(p.nbiome):
            names += ["b%d.%s" % (b, k) for k in ("veg_c", "detritus_c", "soil_c", "permafrost_c",
                                                    "thawedp_c")]
        names += ["HL", "LL", "intermediate", "deep"]
        return v, f, names, err

    def run_ecs_q10(self, S, q10, run_to=None, base=None):
        """-> co2[n, ns], tgav[n, ns], err"""
        S = np.ascontiguousarray(S, dtype=np.float64)
        q10 = np.ascontiguousarray(q10, dtype=np.float64)
        n = S.size
        co2 = np.zeros((n, self.ns))
        tg = np.zeros((n, self.ns))
        base = base or self.default_params()
        err = self.lib.hxo_run_ensemble_ecs_q10(self.sc, ctypes.byref(base), n, S.ctypes.data,
                                                q10.ctypes.data, run_to or self.end,
                                                co2.ctypes.data, tg.ctypes.data)
        return co2, tg, err

    def run_ensemble(self, params, run_to=None, timesteps=True):
        """params: list of Params (one per member) -> co2[n, ns], tgav[n, ns],
        timesteps[n, ns] (uint8, stashes per year) or None, errs[n]; see hector_oracle.h"""
        n = len(params)
        arr = (Params * n)(*params)
        co2 = np.zeros((n, self.ns)); tg = np.zeros((n, self.ns))
        ts = np.zeros((n, self.ns), dtype=np.uint8) if timesteps else None
        errs = np.zeros(n, dtype=np.int32)
        fn = self.lib.hxo_run_ensemble
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        fn(self.sc, ctypes.cast(arr, ctypes.c_void_p), n, run_to or self.end, co2.ctypes.data,
           tg.ctypes.data, ts.ctypes.data if timesteps else None, errs.ctypes.data)
        return co2, tg, ts, errs

    def csys(self, Tc, carbon, alk, volume):
        out = np.zeros(6)
        self.lib.hxo_csys(Tc, carbon, alk, volume, out.ctypes.data)
        return out

    def doeclim_kernel(self, diff, ns):
        out = np.zeros(ns)
        self.lib.hxo_doeclim_kernel(diff, ns, out.ctypes.data)
        return out
