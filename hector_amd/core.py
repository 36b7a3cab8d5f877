"""Python face of the ensemble core, named after the reference's R API
(R/hector.R:57-189, R/messages.R:46-140, R/biome.R:61-130): newcore(), run(),
reset(), shutdown(), setvar(), fetchvars(), split_biome() -- with a member axis.
Everything here is a thin ctypes veneer over the C ABI (include/hector_amd.h);
the numerics live in the HIP kernels.
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import HectorAmdError, DEFAULT_SCENARIO

# (the R-style capability accessors -- ECS(), BETA(biome) ... -- live in hector_amd.capabilities)


class Core:
    """An N-member ensemble core bound to one GPU (device=) or sharded over a list of GPUs
    (devices=[...]: contiguous member blocks, hx_newcore_devices)."""

    def __init__(self, scenario=None, n_members=1, device=0, lib_path=None,
                 allow_emulation=False, name=None, devices=None):
        self._lib = _lib.load(lib_path, allow_emulation)
        self._h = ctypes.c_void_p()
        if devices is None:
            self._ck(self._lib.hx_newcore((scenario or DEFAULT_SCENARIO).encode(), int(n_members),
                                          int(device), ctypes.byref(self._h)))
        else:
            devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
            self._ck(self._lib.hx_newcore_devices((scenario or DEFAULT_SCENARIO).encode(),
                                                  int(n_members), devs, len(devices),
                                                  ctypes.byref(self._h)))
        self.n_members = int(n_members)
        if name is None:   # newcore(..., name =): default here the INI's run_name
            rn = ctypes.c_char_p()
            self._ck(self._lib.hx_run_name(self._h, ctypes.byref(rn)))
            name = rn.value.decode() if rn.value else "Unnamed Hector core"
        self.name = name
        s, e, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._ck(self._lib.hx_dates(self._h, ctypes.byref(s), ctypes.byref(e), ctypes.byref(c)))
        self.strtdate, self.enddate = s.value, e.value

    def _ck(self, rc):
        if rc != 0:
            raise HectorAmdError(self._lib.hx_last_error().decode())

    @property
    def backend(self):
        return self._lib.hx_backend().decode()

    @property
    def current_date(self):
        c = ctypes.c_int()
        self._ck(self._lib.hx_dates(self._h, None, None, ctypes.byref(c)))
        return c.value

    def setvar(self, var, values, unit=None):
        v = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64)))
        self._ck(self._lib.hx_setvar(self._h, var.encode(),
                                     v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), v.size,
                                     unit.encode() if unit else None))
        return self

    def getvar(self, var):
        out = np.empty(self.n_members)
        self._ck(self._lib.hx_getvar(self._h, var.encode(),
                                     out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return out

    def create_biome(self, biome):
        self._ck(self._lib.hx_create_biome(self._h, biome.encode()))
        return self

    def delete_biome(self, biome):
        self._ck(self._lib.hx_delete_biome(self._h, biome.encode()))
        return self

    def rename_biome(self, oldname, newname):
        self._ck(self._lib.hx_rename_biome(self._h, oldname.encode(), newname.encode()))
        return self

    def split_biome(self, names, fveg_c=None, fdetritus_c=None, fsoil_c=None,
                    fpermafrost_c=None, fnpp_flux0=None, old_biome=None):
        n = len(names)
        arr = (ctypes.c_char_p * n)(*[s.encode() for s in names])

        def f(x):
            if x is None:
                return None
            a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
            assert a.size == n
            keep.append(a)
            return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        keep = []
        if old_biome is None:
            self._ck(self._lib.hx_split_biome(self._h, n, arr, f(fveg_c), f(fdetritus_c),
                                              f(fsoil_c), f(fpermafrost_c), f(fnpp_flux0)))
        else:
            self._ck(self._lib.hx_split_biome_of(self._h, old_biome.encode(), n, arr, f(fveg_c),
                                                 f(fdetritus_c), f(fsoil_c), f(fpermafrost_c),
                                                 f(fnpp_flux0)))
        return self

    def set_outputs(self, variables):
        n = len(variables)
        arr = (ctypes.c_char_p * n)(*[s.encode() for s in variables])
        self._ck(self._lib.hx_set_outputs(self._h, n, arr))
        return self

    def tracking_pools(self):
        names = ctypes.POINTER(ctypes.c_char_p)()
        n = ctypes.c_int()
        self._ck(self._lib.hx_tracking_pools(self._h, ctypes.byref(names), ctypes.byref(n)))
        return [names[i].decode() for i in range(n.value)]

    def tracking_data(self, member, dates, masks=False):
        """-> (values[ny, TP], fractions[ny, TP, TP]) for dates = (year0, year1); with masks=True
        also in_map[ny, TP, TP] (bool: the source is in the pool's map)."""
        y0, y1 = int(min(dates)), int(max(dates))
        tp = len(self.tracking_pools())
        v = np.zeros((y1 - y0 + 1, tp)); f = np.zeros((y1 - y0 + 1, tp, tp))
        dp = ctypes.POINTER(ctypes.c_double)
        w = (tp + 63) // 64   # mask words per pool (2 from 12 biomes on)
        mk = np.zeros((y1 - y0 + 1, tp, w), dtype=np.uint64)
        self._ck(self._lib.hx_tracking_data(self._h, int(member), y0, y1, v.ctypes.data_as(dp),
                                            f.ctypes.data_as(dp),
                                            mk.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))))
        if masks:
            src = np.arange(tp)
            bits = (mk[:, :, src // 64] >> (src % 64).astype(np.uint64)[None, None, :]) & np.uint64(1)
            return v, f, bits.astype(bool)
        return v, f

    def getunits(self, var):
        """getunits(var)  R/units.R"""
        u = ctypes.c_char_p()
        self._ck(self._lib.hx_var_info(self._h, var.encode(), None, ctypes.byref(u)))
        return u.value.decode()

    def component_of(self, var):
        comp = ctypes.c_char_p()
        self._ck(self._lib.hx_var_info(self._h, var.encode(), ctypes.byref(comp), None))
        return comp.value.decode()

    def biomes(self):
        """get_biome_list(core)  R/biome.R:8-16"""
        names = ctypes.POINTER(ctypes.c_char_p)()
        n = ctypes.c_int()
        self._ck(self._lib.hx_biomes(self._h, ctypes.byref(names), ctypes.byref(n)))
        return [names[i].decode() for i in range(n.value)]

    def halocarbons(self):
        names = ctypes.POINTER(ctypes.c_char_p)()
        n = ctypes.c_int()
        self._ck(self._lib.hx_halocarbons(self._h, ctypes.byref(names), ctypes.byref(n)))
        return [names[i].decode() for i in range(n.value)]

    def enable_spinup_record(self, on=True):
        """Keep what the reference's output stream sees after every spinup step (its spinup = 1
        rows, csv_outputstream_visitor.cpp:86-95): the carbon-cycle variables."""
        self._ck(self._lib.hx_enable_spinup_record(self._h, 1 if on else 0))
        return self

    def spinup_record(self, member=0):
        """-> dict capability -> values[steps] (steps 1 .. spinup_steps(member))"""
        names = ctypes.POINTER(ctypes.c_char_p)()
        nv = ctypes.c_int()
        self._ck(self._lib.hx_spinup_record(self._h, int(member), ctypes.byref(names), ctypes.byref(nv),
                                            None, 0, None))
        mx = self.spinup_steps(member)
        vals = np.zeros((max(mx, 1), nv.value))
        steps = ctypes.c_int()
        self._ck(self._lib.hx_spinup_record(self._h, int(member), None, None,
                                            vals.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), max(mx, 1),
                                            ctypes.byref(steps)))
        return {names[i].decode(): vals[:steps.value, i].copy() for i in range(nv.value)}

    def enable_history(self, on=True):
        self._ck(self._lib.hx_enable_history(self._h, 1 if on else 0))
        return self

    def setvar_dated(self, var, years, values, unit=None):
        y = np.ascontiguousarray(np.atleast_1d(np.asarray(years, dtype=np.int32)))
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(values, dtype=np.float64), y.shape))
        self._ck(self._lib.hx_setvar_dated(self._h, var.encode(),
                                           y.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                           v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), y.size,
                                           unit.encode() if unit else None))
        return self

    def setvar_dated_members(self, var, years, values, unit=None):
        """values[year, member]: a different input series for every member."""
        y = np.ascontiguousarray(np.atleast_1d(np.asarray(years, dtype=np.int32)))
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64))
        if v.shape != (y.size, self.n_members):
            raise HectorAmdError("setvar_dated_members: values must be [n_years, n_members]")
        self._ck(self._lib.hx_setvar_dated_members(
            self._h, var.encode(), y.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
            v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), y.size,
            unit.encode() if unit else None))
        return self

    def set_member_sorting(self, on=True):
        self._ck(self._lib.hx_set_member_sorting(self._h, 1 if on else 0))
        return self

    def set_lane_calibration(self, on=True):
        """Lane order by measured cost after the first complete run (hx_set_lane_calibration)."""
        self._ck(self._lib.hx_set_lane_calibration(self._h, 1 if on else 0))
        return self

    def lane_order_source(self):
        """What the lanes are ordered by: "parameter key", "measured cost" (a complete run of this
        core) or "cost model" (fitted to an earlier core's measurements: hx_lane_order_source)."""
        v = ctypes.c_int(0)
        self._ck(self._lib.hx_lane_order_source(self._h, ctypes.byref(v)))
        return ("parameter key", "measured cost", "cost model")[v.value]

    def set_cost_model(self, on=True):
        self._ck(self._lib.hx_set_cost_model(self._h, 1 if on else 0))
        return self

    def lanes_calibrated(self):
        v = ctypes.c_int()
        self._ck(self._lib.hx_lanes_calibrated(self._h, ctypes.byref(v)))
        return bool(v.value)

    def lane_of_member(self):
        out = np.zeros(self.n_members, dtype=np.int32)
        self._ck(self._lib.hx_lane_of_member(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return out

    def reset(self, date=0):
        self._ck(self._lib.hx_reset(self._h, float(date)))
        return self

    def run(self, runtodate=-1, wait=True):
        self._ck(self._lib.hx_run(self._h, float(runtodate)))
        # the Rcpp wrapper's check (src/rcpp_hector.cpp:168-175), after the pending auto-reset
        # like there; Core::run itself -- hx_run -- does nothing for such a date (core.cpp:454-458)
        if runtodate > 0 and runtodate < self.current_date:
            raise HectorAmdError("Requested run date %g is prior to the current date of %d. "
                                 "Run reset() to reset to an earlier date."
                                 % (runtodate, self.current_date))
        if wait:
            self._ck(self._lib.hx_sync(self._h))
        return self

    def sync(self):
        self._ck(self._lib.hx_sync(self._h))

    def stream(self):
        """The core's hipStream_t as an integer (for torch.cuda.ExternalStream): every launch of
        this core -- run, statistics, gathers -- is queued on it, not on torch's current stream."""
        p = ctypes.c_void_p()
        self._ck(self._lib.hx_stream(self._h, ctypes.byref(p)))
        return p.value or 0

    def fetchvars(self, var, dates=None, out=None):
        """-> ndarray [n_years, n_members] for dates = (year0, year1) inclusive.  `out`: a
        C-contiguous float64 array of that shape to reuse (a buffer the host has already touched
        takes the device-to-host copy ~3x faster than a fresh allocation)."""
        y0, y1 = (self.strtdate, self.current_date) if dates is None else \
            (int(min(dates)), int(max(dates)))
        shape = (y1 - y0 + 1, self.n_members)
        if out is None:
            out = np.empty(shape)
        elif out.shape != shape or out.dtype != np.float64 or not out.flags["C_CONTIGUOUS"]:
            raise HectorAmdError("fetchvars: out must be a C-contiguous float64 array of shape %r" % (shape,))
        self._ck(self._lib.hx_fetchvars(self._h, var.encode(), y0, y1,
                                        out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return out

    def device_var(self, var):
        p, npad = ctypes.c_void_p(), ctypes.c_int()
        self._ck(self._lib.hx_device_var(self._h, var.encode(), ctypes.byref(p), ctypes.byref(npad)))
        return p.value, npad.value

    def shards(self):
        """-> (devices[n_shards], offsets[n_shards + 1]): shard s holds members
        offsets[s] .. offsets[s + 1] - 1 on GPU devices[s]."""
        n = ctypes.c_int()
        self._ck(self._lib.hx_shards(self._h, ctypes.byref(n), None, None))
        dev = (ctypes.c_int * n.value)()
        off = (ctypes.c_int * (n.value + 1))()
        self._ck(self._lib.hx_shards(self._h, None, dev, off))
        return list(dev), list(off)

    def device_var_shard(self, shard, var):
        p, npad = ctypes.c_void_p(), ctypes.c_int()
        self._ck(self._lib.hx_device_var_shard(self._h, int(shard), var.encode(), ctypes.byref(p),
                                               ctypes.byref(npad)))
        return p.value, npad.value

    def comm_init_rank(self, n_procs, proc_rank, unique_id):
        """Join an RCCL communicator of n_procs x n_shards ranks (hx_comm_init_rank);
        unique_id: the 128 bytes of comm_unique_id() of ONE process."""
        if len(unique_id) != 128:
            raise HectorAmdError("comm_init_rank: the unique id is 128 bytes")
        self._ck(self._lib.hx_comm_init_rank(self._h, int(n_procs), int(proc_rank), bytes(unique_id)))
        return self

    def comm_info(self):
        """-> (world, first_rank, backend); world 0 = no communicator."""
        w, r, b = ctypes.c_int(), ctypes.c_int(), ctypes.c_char_p()
        self._ck(self._lib.hx_comm_info(self._h, ctypes.byref(w), ctypes.byref(r), ctypes.byref(b)))
        return w.value, r.value, (b.value or b"").decode()

    def ensemble_stats(self, variables, dates=None, d_out=None, host=True):
        """Per-year {count, sum, sumsq, min, max} of `variables` over every member on every GPU
        and every process of the communicator (hx_ensemble_stats: local reductions + ONE RCCL
        all-gather): -> ndarray [n_vars, n_years, 5] (host=True) and / or into the device buffer
        d_out (an integer address on the first shard's GPU)."""
        if isinstance(variables, str):
            variables = [variables]
        y0, y1 = (self.strtdate, self.current_date) if dates is None else \
            (int(min(dates)), int(max(dates)))
        arr = (ctypes.c_char_p * len(variables))(*[v.encode() for v in variables])
        out = np.empty((len(variables), y1 - y0 + 1, 5)) if host else None
        self._ck(self._lib.hx_ensemble_stats(
            self._h, len(variables), arr, y0, y1,
            out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if host else None,
            ctypes.c_void_p(d_out) if d_out else None))
        return out

    def stats_device(self, var, year0, year1, d_ptr):
        self._ck(self._lib.hx_stats_device(self._h, var.encode(), int(year0), int(year1),
                                           ctypes.c_void_p(d_ptr)))

    def status(self):
        out = np.zeros(self.n_members, dtype=np.uint32)
        self._ck(self._lib.hx_status(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint))))
        return out

    def state_row(self, row):
        out = np.empty(self.n_members)
        self._ck(self._lib.hx_state_row(self._h, int(row),
                                        out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return out

    def spinup_steps(self, member=0):
        s = ctypes.c_int()
        self._ck(self._lib.hx_spinup_steps(self._h, int(member), ctypes.byref(s)))
        return s.value

    def last_run_ms(self):
        v = ctypes.c_double()
        self._ck(self._lib.hx_last_run_ms(self._h, ctypes.byref(v)))
        return v.value

    def set_pair_kernel_limit(self, max_members):
        """Ensembles of up to max_members (one biome, no constraints, outputs within
        CO2/tas/forcing/pools/NBP/pH) run on the
        two-wavefront kernel (include/hector_amd.h); 0 switches it off."""
        self._ck(self._lib.hx_set_pair_kernel_limit(self._h, int(max_members)))
        return self

    def wave_clock(self, shard=0):
        """-> int64 array [wavefronts][2]: start and end of every wavefront of the last run's
        year-loop launch, in ticks of the device's constant 100 MHz clock relative to the earliest
        start (hx_wave_clock): the launch's tail, wavefront by wavefront."""
        cap = 2 * ((self.n_members + 63) // 64)
        buf = np.zeros((cap, 2), dtype=np.int64)
        n = ctypes.c_int(0)
        self._ck(self._lib.hx_wave_clock(self._h, int(shard), buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
                                         cap, ctypes.byref(n)))
        return buf[:n.value]

    def set_two_wave_from(self, min_members):
        """Ensembles of at least min_members members (one biome, no carbon
        tracking) run on the flavour of the kernel built for two resident
        wavefronts per SIMD (include/hector_amd.h); < 0: the default (more wavefronts than the
        GPU has SIMDs), 0: never."""
        self._ck(self._lib.hx_set_two_wave_from(self._h, int(min_members)))
        return self

    def set_prewarm(self, ms):
        """Keep the chip's clocks up while run()'s preparation uploads and spins up after an idle
        gap (hx_set_prewarm): at most `ms` milliseconds of a small busy loop, 0 = off."""
        self._ck(self._lib.hx_set_prewarm(self._h, int(ms)))
        return self

    def last_run_prewarmed(self):
        v = ctypes.c_int(0)
        self._ck(self._lib.hx_last_run_prewarmed(self._h, ctypes.byref(v)))
        return bool(v.value)

    def last_run_kernel(self):
        """'run', 'run2' or 'pair': the kernel the last run() launched."""
        s = ctypes.c_char_p()
        self._ck(self._lib.hx_last_run_kernel(self._h, ctypes.byref(s)))
        return s.value.decode()

    def last_run_variant(self):
        """The run-kernel family of the last run(): 0 plain, -2 plain + diagnostics, -1 extended,
        1 extended with the NBP machinery, 2 carbon tracking (hx_last_run_variant)."""
        v = ctypes.c_int()
        self._ck(self._lib.hx_last_run_variant(self._h, ctypes.byref(v)))
        return v.value

    def last_spinup_ms(self):
        v = ctypes.c_double()
        self._ck(self._lib.hx_last_spinup_ms(self._h, ctypes.byref(v)))
        return v.value

    def shutdown(self):
        if self._h:
            self._lib.hx_shutdown(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass


def comm_unique_id(lib_path=None, allow_emulation=False):
    """128 bytes that identify a new RCCL communicator (hx_comm_unique_id): made by ONE process,
    handed to the others by the host's own means (MPI, a torch.distributed store, a file)."""
    lib = _lib.load(lib_path, allow_emulation)
    buf = ctypes.create_string_buffer(128)
    if lib.hx_comm_unique_id(buf) != 0:
        raise HectorAmdError(lib.hx_last_error().decode())
    return buf.raw


def cost_models_export(path, lib_path=None, allow_emulation=False):
    """Write every lane-cost model this process holds to `path` (hx_cost_models_export) -> count."""
    lib = _lib.load(lib_path, allow_emulation)
    n = ctypes.c_int(0)
    if lib.hx_cost_models_export(os.fsencode(path), ctypes.byref(n)) != 0:
        raise HectorAmdError(lib.hx_last_error().decode())
    return n.value


def cost_models_load(path, lib_path=None, allow_emulation=False):
    """Add the lane-cost models of a file to the process's registry (hx_cost_models_load) -> count."""
    lib = _lib.load(lib_path, allow_emulation)
    n = ctypes.c_int(0)
    if lib.hx_cost_models_load(os.fsencode(path), ctypes.byref(n)) != 0:
        raise HectorAmdError(lib.hx_last_error().decode())
    return n.value


# R-style free functions
def newcore(inifile=None, n_members=1, device=0, **kw):
    return Core(inifile, n_members, device, **kw)


def run(core, runtodate=-1):
    return core.run(runtodate)


def reset(core, date=0):
    return core.reset(date)


def shutdown(core):
    core.shutdown()


def isactive(core):
    """isactive(core)  R/hector.R:122-125: false after shutdown()."""
    return bool(core._h)


def startdate(core):
    return core.strtdate


def enddate(core):
    return core.enddate


def getdate(core):
    return core.current_date


def getname(core):
    return core.name


def get_biome_list(core):
    return core.biomes()


def getunits(vars, core):
    """getunits(vars)  R/units.R:10-21: unit strings; None (R: NA + a warning) for unknown names.
    The table lives in the library (hx_var_info), hence the core argument."""
    out = []
    for v in ([vars] if isinstance(vars, str) else vars):
        try:
            out.append(core.getunits(v))
        except HectorAmdError:
            out.append(None)
    return out[0] if isinstance(vars, str) else out


def getfxn(strings):
    """getfxn(str)  R/fxns.R:11-21: the accessor ('BETA()') that returns a capability string;
    None for unknown strings."""
    from . import capabilities
    rev = {}
    for fn, (cap, _per_biome) in capabilities._TABLE.items():
        rev.setdefault(cap, fn + "()")
    out = [rev.get(x) for x in ([strings] if isinstance(strings, str) else strings)]
    return out[0] if isinstance(strings, str) else out


def runscenario(infile, n_members=1, **kw):
    """runscenario(infile)  R/hector.R:57-63: run to the end, return the default variables."""
    core = newcore(infile, n_members, **kw)
    try:
        core.set_outputs(list(DEFAULT_FETCHVARS))
        core.run()
        return fetchvars(core, (core.strtdate, core.enddate))
    finally:
        core.shutdown()


def setvar(core, dates, var, values, unit=None):
    if not _is_na(dates):
        return core.setvar_dated(var, dates, values, unit)
    return core.setvar(var, values, unit)


def _is_na(dates):
    return dates is None or (isinstance(dates, float) and np.isnan(dates))


DEFAULT_FETCHVARS = ("CO2_concentration", "RF_tot", "RF_CO2", "global_tas")   # R/messages.R:4 default_fetchvars


def fetchvars(core, dates, variables=None):
    """fetchvars(core, dates, vars)  R/messages.R:46-88.  dates = a tuple (year0, year1) for the
    whole range or a list of years: -> dict variable -> ndarray [n_dates, n_members]; dates =
    None / NaN (R's NA), for parameters: -> dict variable -> ndarray [n_members]."""
    if variables is None:   # getOption("hector.default.fetchvars"), R/messages.R:47-56
        if _is_na(dates):
            raise HectorAmdError("The default vars (%s) all require dates" % ", ".join(DEFAULT_FETCHVARS))
        variables = list(DEFAULT_FETCHVARS)
    if isinstance(variables, str):
        variables = [variables]
    if _is_na(dates):
        return {v: core.getvar(v) for v in variables}
    # dates outside startDate..current date are dropped, none left is an error (:62-72)
    strt, cur = core.strtdate, core.current_date
    want = [int(d) for d in np.atleast_1d(dates)]
    if len(want) == 2 and isinstance(dates, tuple):      # (year0, year1): the whole range
        want = list(range(min(want), max(want) + 1))
    valid = [d for d in want if strt <= d <= cur]
    if not valid:
        raise HectorAmdError("None of these dates are valid for this core (start=%d, current=%d)"
                             % (strt, cur))
    lo, hi = min(valid), max(valid)
    idx = np.array(valid) - lo
    return {v: core.fetchvars(v, (lo, hi))[idx] for v in variables}


GETDATA, SETDATA = "getData", "setData"      # component_data.hpp:409-410


def sendmessage(core, msgtype, capability, date=None, value=None, unit=None):
    """sendmessage(core, msgtype, capability, date, value, unit)  src/rcpp_hector.cpp:262-350,
    the low-level message bus under setvar/fetchvars.  getData -> list of rows
    (year or None, variable, values[n_members], units); setData -> the core."""
    if msgtype == GETDATA:
        units = core.getunits(capability)
        if _is_na(date):
            return [(None, capability, core.getvar(capability), units)]
        years = [int(y) for y in np.atleast_1d(date)]
        data = core.fetchvars(capability, (min(years), max(years)))
        return [(y, capability, data[y - min(years)], units) for y in years]
    if msgtype == SETDATA:
        return setvar(core, date, capability, value, unit)
    raise HectorAmdError("sendmessage: unknown message type %r" % (msgtype,))


def get_tracking_data(core, member=0):
    """get_tracking_data(core)  R/hector.R: rows (year, component, pool_name, pool_value,
    pool_units, source_name, source_fraction) for one member, trackingDate .. current date."""
    start = int(core.getvar("trackingDate")[0])
    if start == 9999 or core.current_date < start:
        return []
    names = core.tracking_pools()
    v, f, held = core.tracking_data(member, (start, core.current_date), masks=True)
    rows = []
    for iy in range(v.shape[0]):
        for p, pn in enumerate(names):
            comp = "ocean" if pn in ("HL", "LL", "intermediate", "deep") else "simpleNbox"
            for s, sn in enumerate(names):
                if held[iy, p, s]:
                    rows.append((start + iy, comp, pn, v[iy, p], "Pg C", sn, f[iy, p, s]))
    return rows


_BIOME_PARAMS = ("warmingfactor", "beta", "q10_rh", "f_nppv", "f_nppd", "f_litterd")


def split_biome(core, old_biome, new_biomes, fveg_c=None, fdetritus_c=None, fsoil_c=None,
                fpermafrost_c=None, fnpp_flux0=None, **params):
    """split_biome(core, old_biome, new_biomes, ...)  R/biome.R:61-130.  `params`: warmingfactor,
    beta, q10_rh, f_nppv, f_nppd, f_litterd for the new biomes (a scalar, or one value per new
    biome); default: the old biome's."""
    new_biomes = list(new_biomes)
    bad = set(params) - set(_BIOME_PARAMS)
    if bad:
        raise HectorAmdError("split_biome: unknown biome parameter(s) %s" % sorted(bad))
    for name, f in (("fveg_c", fveg_c), ("fdetritus_c", fdetritus_c), ("fsoil_c", fsoil_c),
                    ("fpermafrost_c", fpermafrost_c), ("fnpp_flux0", fnpp_flux0)):
        if f is None:
            continue
        f = np.asarray(f, dtype=float)  # the stopifnot block of the R function
        lo_ok = (f >= 0).all() if name == "fpermafrost_c" else (f > 0).all()
        if f.size != len(new_biomes) or abs(f.sum() - 1.0) > 1e-12 or not lo_ok:
            raise HectorAmdError("split_biome: %s must be %d positive fractions summing to 1"
                                 % (name, len(new_biomes)))
    core.split_biome(new_biomes, fveg_c, fdetritus_c, fsoil_c, fpermafrost_c, fnpp_flux0,
                     old_biome=old_biome)
    for key, val in params.items():
        vals = np.broadcast_to(np.asarray(val, dtype=float), (len(new_biomes),))
        for b, v in zip(new_biomes, vals):
            core.setvar("%s.%s" % (b, key), [v])
    return core


def create_biome(core, biome, veg_c0, detritus_c0, soil_c0, permafrost_c0, npp_flux0,
                 warmingfactor, beta, q10_rh, f_nppv, f_nppd, f_litterd):
    """create_biome(core, biome, ...)  R/biome.R:20-40 (values: a scalar or one per member)."""
    core.create_biome(biome)
    for key, val in (("veg_c", veg_c0), ("detritus_c", detritus_c0), ("soil_c", soil_c0),
                     ("permafrost_c", permafrost_c0), ("npp_flux0", npp_flux0),
                     ("warmingfactor", warmingfactor), ("beta", beta), ("q10_rh", q10_rh),
                     ("f_nppv", f_nppv), ("f_nppd", f_nppd), ("f_litterd", f_litterd)):
        core.setvar("%s.%s" % (biome, key), np.atleast_1d(np.asarray(val, dtype=float)))
    return core


def rename_biome(core, oldname, newname):
    return core.rename_biome(oldname, newname)


def get_biome_inits(core, biome):
    """get_biome_inits(core, biome)  R/biome.R:140-170: initial pools and parameters of a biome
    (per member)."""
    pre = "" if (biome == "global" and core.biomes() == ["global"]) else biome + "."
    out = {k: core.getvar(pre + k) for k in ("veg_c", "detritus_c", "soil_c", "permafrost_c",
                                             "npp_flux0", "f_litterd", "f_nppd", "f_nppv", "beta",
                                             "q10_rh", "warmingfactor")}
    out["thawedp_c"] = np.zeros(core.n_members)  # always empty at t = 0 (simpleNbox.cpp:146)
    return out
