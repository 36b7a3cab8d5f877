#!/usr/bin/env python3
"""Import a Hector INI + its csv: tables into a dense "scenario pack" (.hxs).

Runs ONLY in the development container (needs the reference's *input data*,
/root/reference/inst/input/...).  The pack is pure data: every scalar key of
the INI and every time series densified on startDate..endDate, so that the
GPU box (which has no /root/reference) can build a core from it.

Semantics reproduced (reference file:line):
  * INI syntax `key=value`, `key[year]=value`, `key=csv:file`, `;` comments
    after whitespace                          src/ini.c:35-45,88-110
    src/ini_to_core_reader.cpp:100-180
  * csv tables: header row names the column, first column is the date,
    `;`-prefixed and UNITS rows skipped, blanks ignored
                                              src/csv_table_reader.cpp:115-198
  * a time series read at a date: exact hit -> the value; otherwise linear
    interpolation between neighbours, FLAT beyond either end; a one-point
    series is a constant                      inst/include/tseries.hpp:302-334
                                              src/h_interpolator.cpp:103-122

Pack format (line oriented, ASCII):
    HXS 1
    meta   <key> <value>
    scalar <section> <key> <value-as-written>
    series <section> <key> <first_year> <n> <v0> <v1> ...   (repr() doubles)
"""
import os
import sys


def parse_ini(path):
    """-> list of (section, name, value) in file order."""
    out = []
    section = ""
    with open(path, "r", encoding="utf-8", errors="replace") as f:
        for raw in f:
            line = raw.strip()
            if not line or line[0] in ";#":
                continue
            if line[0] == "[":
                section = line[1:line.index("]")].strip()
                continue
            # strip inline comment: ';' preceded by whitespace
            cut = None
            prev_ws = False
            for i, ch in enumerate(line):
                if ch == ";" and prev_ws:
                    cut = i
                    break
                prev_ws = ch in " \t"
            if cut is not None:
                line = line[:cut].rstrip()
            if "=" in line:
                k, v = line.split("=", 1)
            elif ":" in line:
                k, v = line.split(":", 1)
            else:
                continue
            out.append((section, k.strip(), v.strip()))
    return out


_csv_cache = {}


def read_csv_column(path, column):
    """-> dict date -> float for one named column."""
    if path not in _csv_cache:
        rows = []
        with open(path, "r", encoding="utf-8", errors="replace") as f:
            for raw in f:
                line = raw.rstrip("\n").rstrip("\r")
                if not line or line.lstrip().startswith(";"):
                    continue
                rows.append(line.split(","))
        _csv_cache[path] = rows
    rows = _csv_cache[path]
    header = [c.strip() for c in rows[0]]
    if column not in header[1:]:
        raise KeyError("no column %s in %s" % (column, path))
    ci = header.index(column, 1)
    data = {}
    for r in rows[1:]:
        idx = r[0].strip()
        if idx == "UNITS" or idx == "":
            continue
        cell = r[ci].strip() if ci < len(r) else ""
        if cell == "":
            continue
        data[float(idx)] = float(cell)
    return data


def tseries_get(points, t):
    """tseries<T>::get + h_interpolator::f_linear semantics."""
    if len(points) == 1:
        return next(iter(points.values()))
    if t in points:
        return points[t]
    xs = sorted(points)
    if t < xs[0]:
        return points[xs[0]]
    if t > xs[-1]:
        return points[xs[-1]]
    # neighbours
    lo = max(x for x in xs if x < t)
    hi = min(x for x in xs if x > t)
    y0, y1 = points[lo], points[hi]
    return y0 + (t - lo) * (y1 - y0) / (hi - lo)


def constraint_get(key, points, y):
    """Constraint series keep the reference's per-series rule: CO2/NBP/CH4/N2O/halocarbon
    constraints exist only at the dates given (tseries::exists); tas_constrain interpolates
    between its first and last date (temperature_component.cpp:510-511); RF_tot_constrain is
    also used, flat, before its first date (forcing_component.cpp:498).  NaN = no value."""
    if y in points:
        return points[y]
    if key not in ("tas_constrain", "RF_tot_constrain"):
        return float("nan")
    xs = sorted(points)
    if y > xs[-1]:
        return float("nan")
    if y < xs[0]:
        return points[xs[0]] if key == "RF_tot_constrain" else float("nan")
    return tseries_get(points, y)


def main(ini_path, out_path):
    items = parse_ini(ini_path)
    inidir = os.path.dirname(os.path.abspath(ini_path))
    start = end = None
    for s, k, v in items:
        if s == "core" and k == "startDate":
            start = int(float(v))
        if s == "core" and k == "endDate":
            end = int(float(v))
    assert start is not None and end is not None
    scalars = []
    series = {}
    order = []
    for s, k, v in items:
        if "[" in k:  # key[year]=value
            name = k[:k.index("[")]
            year = float(k[k.index("[") + 1:k.index("]")])
            series.setdefault((s, name), {})[year] = float(v)
            if (s, name) not in order:
                order.append((s, name))
        elif v.startswith("csv:"):
            p = v[4:]
            if not os.path.exists(p):
                p = os.path.join(inidir, p)
            series.setdefault((s, k), {}).update(read_csv_column(p, k))
            if (s, k) not in order:
                order.append((s, k))
        else:
            scalars.append((s, k, v))
    n = end - start + 1
    with open(out_path, "w") as f:
        f.write("HXS 1\n")
        f.write("meta source %s\n" % os.path.basename(ini_path))
        f.write("meta generator tools/import_scenario.py\n")
        for s, k, v in scalars:
            f.write("scalar %s %s %s\n" % (s, k, v))
        for (s, k) in order:
            pts = series[(s, k)]
            if k.endswith("_constrain"):
                vals = [constraint_get(k, pts, float(y)) for y in range(start, end + 1)]
            else:
                vals = [tseries_get(pts, float(y)) for y in range(start, end + 1)]
            f.write("series %s %s %d %d %s\n" %
                    (s, k, start, n, " ".join(repr(float(x)) for x in vals)))
    print("wrote %s: %d scalars, %d series x %d years" %
          (out_path, len(scalars), len(order), n))


if __name__ == "__main__":
    ini = sys.argv[1] if len(sys.argv) > 1 else \
        "/root/reference/inst/input/hector_ssp245.ini"
    out = sys.argv[2] if len(sys.argv) > 2 else \
        os.path.join(os.path.dirname(__file__), "..", "hector_amd", "data",
                     "ssp245.hxs")
    main(ini, out)
