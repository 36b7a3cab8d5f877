import sys, os, tempfile, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT") or os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, R); sys.path.insert(0, R + "/oracle"); sys.path.insert(0, R + "/tests")
import hector_amd, oracle_binding
from conftest import edited_pack
kw = dict(device=0) if "--gpu" in sys.argv else dict(lib_path=R + "/tests/emul/libhector_amd_emul.so", allow_emulation=True)
SH = {"M0": "CH4", "Tsoil": "CH4", "Tstrat": "CH4", "N0": "N2O", "PO3": "ozone", "TOH0": "OH", "delta_co2": "forcing", "delta_ch4": "forcing",
      "delta_n2o": "forcing", "rho_bc": "forcing", "rho_oc": "forcing", "rho_so2": "forcing", "rho_nh3": "forcing"}
rng = np.random.default_rng(3); tmp = tempfile.mkdtemp(); worst = 0
for rd in range(10):
    name = ["ssp119", "ssp245", "ssp585", "ssp370"][rd % 4]
    path = os.path.join(R, "hector_amd", "data", name + ".hxs")
    c = hector_amd.Core(path, 2, **kw); c.setvar("S", [2.5, 4.0], "degC")
    sc = {}
    for k, sec in SH.items():
        v0 = c.getvar(k)[0]; v = v0 * rng.uniform(0.7, 1.3) if v0 != 0 else rng.uniform(-0.2, 0.2)
        c.setvar(k, [v]); sc[(sec, k)] = v
    for h in c.halocarbons()[:6]:
        for pre in ("rho_", "delta_"):
            v0 = c.getvar(pre + h)[0]; v = v0 * rng.uniform(0.7, 1.3) if v0 != 0 else rng.uniform(-0.2, 0.2)
            c.setvar(pre + h, [v]); sc[(h + "_halocarbon", pre + h)] = v
    c.set_outputs(["CO2_concentration", "global_tas", "RF_tot", "CH4_concentration", "RF_CH4", "RF_N2O"]); c.run(2300)
    assert (c.status() == 0).all()
    o = oracle_binding.Oracle(edited_pack(os.path.join(tmp, "s%d.hxs" % rd), None, None, [], [], base=path, scalars=sc))
    for i, S in enumerate((2.5, 4.0)):
        p = o.default_params(); p.S = S
        r, err, _ = o.run(p); assert err == 0
        for v in ("CO2_concentration", "global_tas", "RF_tot", "CH4_concentration", "RF_CH4", "RF_N2O"):
            d = np.abs(c.fetchvars(v, (1745, 2300))[:, i] - r[v]).max() / max(1.0, np.abs(r[v]).max()); worst = max(worst, d)
            assert d < 2e-8, (name, v, d)
print("shared-parameter fuzz ok, worst %.2e" % worst)
