"""The command-line front end (hector_amd/csrc/hx_main.cpp; reference: src/main.cpp +
src/csv_outputstream_visitor.cpp): runs a scenario file and writes outputstream_<run>.csv
in the reference's long format.  Here the test-only emulation build of the same source;
test_gpu_parity.py runs the product binary."""
import csv
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, SCENARIO

EMUL_CLI = os.path.join(ROOT, "tests", "emul", "hector-amd-emul")
GOLDEN_TO_STREAM = {"CO2_concentration": "CO2_concentration", "global_tas": "global_tas",
                    "RF_tot": "RF_tot", "RF_CO2": "RF_CO2", "heatflux": "heatflux",
                    "ocean_c": "ocean_c", "HL_pH": "HL_pH", "atmos_co2": "atmos_co2",
                    "sst": "sst", "permafrost_c": "permafrost_c"}


def read_stream(path):
    with open(path) as f:
        first = f.readline()
        assert first.startswith("# Output from")
        rows = list(csv.DictReader(f))
    assert list(rows[0].keys()) == ["year", "run_name", "spinup", "component", "variable", "value",
                                    "units"]
    return rows


def check_stream_against_golden(rows, golden, run_name):
    by = {}
    for r in rows:
        assert r["spinup"] in ("0", "1")
        if r["run_name"] == run_name and r["spinup"] == "0":
            by.setdefault(r["variable"], {})[int(r["year"])] = float(r["value"])
    for gv, sv in GOLDEN_TO_STREAM.items():
        ref = golden[gv]
        sig = 4 if sv.startswith("RF_") else 6          # digits the reference's stream prints
        first = 1750 if sv.startswith("RF_") else 1746  # forcings start at the base year
        for y in range(first, 2301, 7):
            want = float("%.*g" % (sig, ref[y - 1745]))
            assert by[sv][y] == pytest.approx(want, rel=2e-6 if sig == 6 else 2e-4, abs=1e-12), (sv, y)
    return by


def test_cli_writes_the_reference_output_stream(emul_lib, golden, tmp_path):
    assert os.path.exists(EMUL_CLI)
    r = subprocess.run([EMUL_CLI, SCENARIO, "--output-dir", str(tmp_path)], capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stderr
    rows = read_stream(tmp_path / "outputstream_ssp245.csv")
    by = check_stream_against_golden(rows, golden, "ssp245")
    comps = {(r["component"], r["variable"], r["units"]) for r in rows}
    for want in [("simpleNbox", "NBP", "Pg C/yr"), ("ocean", "HL_Revelle", "(unitless)"),
                 ("temperature", "gmst", "degC"), ("forcing", "RF_CF4", "W/m2"),
                 ("CF4_halocarbon", "CF4_concentration", "pptv"), ("slr", "slr", "cm"),
                 ("OH", "TAU_OH", "Years"), ("ozone", "O3_concentration", "DU O3"),
                 ("N2O", "N2O_concentration", "ppbv N2O")]:
        assert want in comps, want
    assert min(by["RF_tot"]) == 1750 and min(by["slr"]) == 1746 and min(by["sl_rc"]) == 1990
    # the spinup = 1 rows (csv_outputstream_visitor.cpp:86-95): the step number in the year
    # column, one set of carbon-cycle rows per spinup step, ahead of the run years
    spin = [r for r in rows if r["spinup"] == "1"]
    assert rows.index(spin[-1]) < rows.index(next(r for r in rows if r["spinup"] == "0"))
    steps = sorted({int(r["year"]) for r in spin})
    assert steps == list(range(1, 499)) and len(spin) == 498 * 25
    last = {r["variable"]: float(r["value"]) for r in spin if r["year"] == "498"}
    assert last["CO2_concentration"] == pytest.approx(277.15, rel=1e-6)     # pinned to C0
    assert last["veg_c"] == pytest.approx(golden["veg_c"][0], rel=2e-6) if "veg_c" in golden else True
    assert last["atmos_co2"] == pytest.approx(golden["atmos_co2"][0], rel=2e-6)
    assert last["ocean_c"] == pytest.approx(golden["ocean_c"][0], rel=2e-6)
    assert last["permafrost_c"] == pytest.approx(golden["permafrost_c"][0], rel=2e-6)
    assert {r["component"] for r in spin} == {"simpleNbox", "ocean"}
    r = subprocess.run([EMUL_CLI, SCENARIO, "--output-dir", str(tmp_path), "--no-spinup-rows",
                        "--run-to", "1750"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert all(x["spinup"] == "0" for x in read_stream(tmp_path / "outputstream_ssp245.csv"))


def test_cli_ensemble_members_and_errors(emul_lib, golden, tmp_path):
    r = subprocess.run([EMUL_CLI, SCENARIO, "--members", "3", "--set", "S=2.0,3.0,4.5",
                        "--run-to", "2100", "--output-dir", str(tmp_path), "--precision", "15"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = read_stream(tmp_path / "outputstream_ssp245.csv")
    assert {r["run_name"] for r in rows} == {"ssp245.0", "ssp245.1", "ssp245.2"}
    t = {r["run_name"]: float(r["value"]) for r in rows
         if r["variable"] == "global_tas" and r["year"] == "2100"}
    assert t["ssp245.0"] < t["ssp245.1"] < t["ssp245.2"]
    assert abs(t["ssp245.1"] - golden["global_tas"][2100 - 1745]) < 2e-8   # S = 3 is the default
    # src/main.cpp:47-60: missing file / no argument -> message and exit code 1
    r = subprocess.run([EMUL_CLI, str(tmp_path / "nope.ini")], capture_output=True, text=True)
    assert r.returncode == 1 and "Couldn't find input file" in r.stderr
    r = subprocess.run([EMUL_CLI], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr


def test_cli_shards_the_members_over_a_device_list(emul_lib, tmp_path):
    """hector-amd --devices i,j,...: the C++ host path over hx_newcore_devices -- the same stream
    as one core, value for value (host build: the device list [0, 0], statistics aside)."""
    args = ["--members", "5", "--set", "S=2.0,2.5,3.0,3.5,4.5", "--run-to", "1900", "--precision", "17"]
    a, b = tmp_path / "one", tmp_path / "many"
    for d, extra in ((a, []), (b, ["--devices", "0,0"])):
        r = subprocess.run([EMUL_CLI, SCENARIO] + args + extra + ["--output-dir", str(d)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    body = lambda p: [l for l in open(p / "outputstream_ssp245.csv") if not l.startswith("#")]
    assert body(a) == body(b) and len(body(a)) > 1000


def test_cli_prints_biome_rows_for_multi_biome_scenarios(emul_lib, tmp_path):
    """csv_outputstream_visitor.cpp:169-198: per-biome rows when the core has several biomes."""
    from test_biomes_ini import biome_pack
    path = biome_pack(tmp_path / "biome.hxs")
    r = subprocess.run([EMUL_CLI, path, "--run-to", "1800", "--output-dir", str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = read_stream(tmp_path / "outputstream_ssp245.csv")
    names = {r["variable"] for r in rows if r["year"] == "1800"}
    for b in ("boreal", "tropical"):
        for v in ("NPP", "RH", "rh_ch4", "veg_c", "detritus_c", "soil_c", "permafrost_c", "thawedp_c",
                  "f_frozen", "detritus_tempfert", "soil_tempfert"):
            assert b + "." + v in names
    veg = {r["variable"]: float(r["value"]) for r in rows if r["year"] == "1800" and r["variable"].endswith("veg_c")}
    assert veg["boreal.veg_c"] + veg["tropical.veg_c"] == pytest.approx(veg["veg_c"], rel=1e-5)


def test_cli_writes_tracking_csv(emul_lib, tmp_path):
    """src/main.cpp:91-105 + CSVFluxPoolVisitor: [core] trackingDate inside the run ->
    tracking_<run_name>.csv, rows year,component,pool_name,pool_value,pool_units,source_name,
    source_fraction for trackingDate..end; fractions of every (year, pool) sum to 1."""
    from conftest import edited_pack
    pack = edited_pack(tmp_path / "t.hxs", None, None, [], [], scalars={("core", "trackingDate"): 1850})
    r = subprocess.run([EMUL_CLI, pack, "--run-to", "1900", "--output-dir", str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    with open(tmp_path / "tracking_ssp245.csv") as f:
        head = f.readline().strip()
        rows = [l.strip().split(",") for l in f]
    assert head == "year,component,pool_name,pool_value,pool_units,source_name,source_fraction"
    assert sorted({int(r[0]) for r in rows}) == list(range(1850, 1901))
    tot = {}
    for r in rows:
        tot[(r[0], r[2])] = tot.get((r[0], r[2]), 0.0) + float(r[6])
    assert len(tot) == 51 * 11 and max(abs(v - 1.0) for v in tot.values()) < 1e-4
    assert {r[1] for r in rows if r[2] in ("HL", "LL", "intermediate", "deep")} == {"ocean"}
    # without a tracking date no tracking file (the reference writes an empty one)
    assert not os.path.exists(tmp_path / "tracking_ssp245.0.csv")


def test_cli_rh_ch4_row_is_what_the_reference_stream_prints(emul_lib, tmp_path):
    """csv_outputstream_visitor.cpp:151 writes final_rh under the name rh_ch4: the file does too
    (fetchvars("rh_ch4") is the CH4 respiration, tested in test_diagnostics)."""
    r = subprocess.run([EMUL_CLI, SCENARIO, "--run-to", "2050", "--output-dir", str(tmp_path),
                        "--precision", "15"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = read_stream(tmp_path / "outputstream_ssp245.csv")
    v = {(x["variable"], x["year"]): float(x["value"]) for x in rows if x["variable"] in ("RH", "rh_ch4")}
    for y in ("1800", "2000", "2050"):
        assert v[("rh_ch4", y)] == v[("RH", y)] and v[("RH", y)] > 10.0


def test_cli_output_off_silences_a_components_rows(emul_lib, tmp_path):
    """`output=0` in a component's section (Core::outputEnabled, src/core.cpp:257-262): the stream
    visitor leaves that component's rows out (every visit() of csv_outputstream_visitor.cpp);
    the run itself is unchanged."""
    from conftest import edited_pack
    p = edited_pack(tmp_path / "quiet.hxs", None, None, [], [],
                    scalars={("ocean", "output"): 0.0, ("forcing", "output"): 0.0, ("CF4_halocarbon", "output"): 0.0})
    r = subprocess.run([EMUL_CLI, str(p), "--output-dir", str(tmp_path), "--run-to", "1800"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = read_stream(next(tmp_path.glob("outputstream_*.csv")))
    comps = {r["component"] for r in rows}
    assert "ocean" not in comps and "forcing" not in comps and "CF4_halocarbon" not in comps
    assert {"simpleNbox", "temperature", "C2F6_halocarbon", "OH"} <= comps
    q = subprocess.run([EMUL_CLI, SCENARIO, "--output-dir", str(tmp_path / "full"), "--run-to", "1800"],
                       capture_output=True, text=True)
    assert q.returncode == 0, q.stderr
    full = read_stream(next((tmp_path / "full").glob("outputstream_*.csv")))
    keep = [(r["year"], r["component"], r["variable"], r["value"]) for r in full
            if r["component"] not in ("ocean", "forcing", "CF4_halocarbon")]
    assert keep == [(r["year"], r["component"], r["variable"], r["value"]) for r in rows]
