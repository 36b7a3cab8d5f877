#!/usr/bin/env python3
"""Convert the reference's own golden trajectory into a compact fixture.

Source (data file held by the reference's tests, not code):
  /root/reference/tests/testthat/compdata/hector_comp.csv
  -- 10 variables x 556 years, SSP2-4.5 default member, Hector v3.5.0 @ e98dc2d,
  written by data-raw/generate-comp-data.R and asserted to 1e-10 by
  tests/testthat/test_old-new.R:11.

Output: tests/golden/hector_comp_ssp245.txt, one line per variable:
  <variable> <first_year> <n> <value strings exactly as in the CSV...>
"""
import csv, os, sys
src = sys.argv[1] if len(sys.argv) > 1 else \
    "/root/reference/tests/testthat/compdata/hector_comp.csv"
dst = os.path.join(os.path.dirname(__file__), "..", "tests", "golden",
                   "hector_comp_ssp245.txt")
rows = {}
with open(src) as f:
    for r in csv.DictReader(f):
        rows.setdefault(r["variable"], {})[int(r["year"])] = r["value"]
with open(dst, "w") as f:
    f.write("# golden trajectory: reference tests/testthat/compdata/hector_comp.csv "
            "(hector_ssp245.ini, v3.5.0, commit e98dc2d); made by tools/make_golden.py\n")
    for var in sorted(rows):
        ys = sorted(rows[var])
        assert ys == list(range(ys[0], ys[0] + len(ys)))
        f.write("%s %d %d %s\n" % (var, ys[0], len(ys), " ".join(rows[var][y] for y in ys)))
print("wrote", dst, {k: len(v) for k, v in rows.items()})
