// hx_dev_const.h -- model constants of the reference's headers (DOECLIM, ocean geometry)
// Part of the device code of hx_kernels.hip (one translation unit; see its header for the
// reference file:line map).
#pragma once

// a branch the common run does not take: its block is laid out behind the function's hot path, so
// that the hot path falls through (a taken branch costs a lone wavefront an instruction-fetch
// bubble of ~20 clocks; ~70 branches a model year, SQ_WAIT_INST_ANY = 6.6 % of the wavefront's time)
#define HX_RARE(x) __builtin_expect(!!(x), 0)

namespace {

constexpr double PGC2PPM = 1.0 / 2.13;  // carbon-cycle-model.hpp:29
constexpr double PG_C_TO_TG_CH4 = 1000.0 * 16.04 / 12.01;

// ---- DOECLIM constants  inst/include/temperature_component.hpp:77-98 -------
constexpr double D_ak = 0.31, D_bk = 1.59, D_csw = 0.13,
                 D_secs = 60.0 * 60.0 * 24.0 * 365.2422, D_rlam = 1.43,
                 D_zbot = 4000.0, D_bsi = 1.3, D_cal = 0.52, D_cas = 7.80,
                 D_flnd = 0.29, D_fso = 0.95;

// ---- ocean geometry  src/ocean_component.cpp:202-303 -----------------------
constexpr double O_part_high = 0.15, O_part_low = 1 - 0.15;
constexpr double O_spy = 60.0 * 60 * 24 * 365.25;
constexpr double O_area = 3.6e14;
constexpr double O_vLL = O_area * O_part_low * 100.0;
constexpr double O_vHL = O_area * O_part_high * 100.0;
constexpr double O_vI = O_area * 900.0;
constexpr double O_vD = O_area * (3777.0 - 900.0 - 100.0);
constexpr double O_AsHL = O_area * O_part_high, O_AsLL = O_area * O_part_low;
constexpr double O_S = 34.5, O_U = 6.7;

}  // namespace
