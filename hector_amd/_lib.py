"""ctypes loader for libhector_amd.so (the C ABI in include/hector_amd.h).

The product library is the HIP build in hector_amd/lib/.  There is no CPU
execution path: if the library is missing or reports a backend other than
"hip" the loader raises.  (tests/ may pass an explicit path to the test-only
host-emulation build together with allow_emulation=True.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libhector_amd.so")
DEFAULT_SCENARIO = os.path.join(_HERE, "data", "ssp245.hxs")

_cache = {}


class HectorAmdError(RuntimeError):
    pass


def _elf_dynamic_strings(path, tags):
    """Values of string-valued entries of an ELF64 file's dynamic section (DT_NEEDED = 1,
    DT_SONAME = 14) -> {tag: [strings]}; {} if the file cannot be read that way."""
    import struct
    out = {t: [] for t in tags}
    try:
        with open(path, "rb") as f:
            d = f.read()
        if d[:4] != b"\x7fELF" or d[4] != 2:
            return {}
        shoff, = struct.unpack_from("<Q", d, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", d, 0x3A)
        secs = [struct.unpack_from("<IIQQQQIIQQ", d, shoff + i * shentsize) for i in range(shnum)]
        for (_n, typ, _fl, _ad, off, size, link, _i, _al, ent) in secs:
            if typ != 6:     # SHT_DYNAMIC
                continue
            stroff = secs[link][4]
            for k in range(size // (ent or 16)):
                tag, val = struct.unpack_from("<qQ", d, off + k * 16)
                if tag in out:
                    end = d.index(b"\0", stroff + val)
                    out[tag].append(d[stroff + val:end].decode())
        return out
    except Exception:
        return {}


def _share_torch_hip_runtime(lib_path):
    """One HIP runtime per process.  PyTorch-ROCm wheels carry their own libamdhip64.so (SONAME
    libamdhip64.so.7) which libtorch_hip.so asks for by FILE name: if the system runtime is already
    in the process under its SONAME (because this library was loaded first), the dynamic loader
    does not recognise it, torch brings in a second runtime and its device init fails with "No HIP
    GPUs are available".  The other order works (this library asks for the SONAME and gets torch's
    copy).  So where a torch installation with its own runtime exists, load that copy first --
    without importing torch -- PROVIDED it is the runtime this library was linked against: its
    SONAME has to be one of the library's DT_NEEDED entries (a wheel built for another ROCm major
    version is left alone: the library then runs on the system runtime it was built for, and a
    process that also wants that torch has to import torch first).  HECTOR_AMD_NO_TORCH_HIP=1
    switches the preload off; HECTOR_AMD_VERBOSE=1 says which runtime was chosen."""
    import importlib.util
    import sys
    verbose = os.environ.get("HECTOR_AMD_VERBOSE") == "1"
    if os.environ.get("HECTOR_AMD_NO_TORCH_HIP") == "1":
        if verbose:
            sys.stderr.write("hector_amd: HECTOR_AMD_NO_TORCH_HIP=1, system HIP runtime\n")
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    rt = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(rt):
        return
    soname = (_elf_dynamic_strings(rt, (14,)).get(14) or [None])[0]
    needed = _elf_dynamic_strings(lib_path, (1,)).get(1) or []
    if soname is not None and needed and soname not in needed:
        sys.stderr.write("hector_amd: PyTorch's bundled HIP runtime (%s) is not the one %s was linked "
                         "against (%s): not preloading it; import torch BEFORE hector_amd if both are "
                         "needed in one process\n"
                         % (soname, os.path.basename(lib_path),
                            ", ".join(n for n in needed if "amdhip" in n) or "?"))
        return
    ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
    if verbose:
        sys.stderr.write("hector_amd: HIP runtime %s (%s)\n" % (rt, soname))


def load(path=None, allow_emulation=False):
    path = os.path.abspath(path or DEFAULT_LIB)
    if path in _cache:
        lib = _cache[path]
    else:
        if not os.path.exists(path):
            raise HectorAmdError(
                "hector_amd: native library %s not found -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no Python/CPU fallback" % path)
        if os.sep + os.path.join("tests", "emul") + os.sep not in path:
            _share_torch_hip_runtime(path)
        lib = ctypes.CDLL(path)
        _declare(lib)
        _cache[path] = lib
    backend = lib.hx_backend().decode()
    # The only other backend that exists is the test suite's host-emulation build of the same
    # sources; it is accepted only where the tests build it (tests/emul/) and only on request.
    in_tests = os.sep + os.path.join("tests", "emul") + os.sep in path
    if backend != "hip" and not (allow_emulation and in_tests):
        raise HectorAmdError("hector_amd: library %s has backend %r; only the HIP build is a "
                             "product path" % (path, backend))
    return lib


def _declare(lib):
    c = ctypes
    P = c.c_void_p
    dp = c.POINTER(c.c_double)
    lib.hx_backend.restype = c.c_char_p
    lib.hx_build_info.restype = c.c_char_p
    lib.hx_last_error.restype = c.c_char_p
    sig = {
        "hx_newcore": [c.c_char_p, c.c_int, c.c_int, c.POINTER(P)],
        "hx_newcore_devices": [c.c_char_p, c.c_int, c.POINTER(c.c_int), c.c_int, c.POINTER(P)],
        "hx_shards": [P, c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(c.c_int)],
        "hx_device_var_shard": [P, c.c_int, c.c_char_p, c.POINTER(P), c.POINTER(c.c_int)],
        "hx_stream_shard": [P, c.c_int, c.POINTER(P)],
        "hx_comm_unique_id": [c.c_char_p],
        "hx_comm_init_rank": [P, c.c_int, c.c_int, c.c_char_p],
        "hx_comm_info": [P, c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(c.c_char_p)],
        "hx_ensemble_stats": [P, c.c_int, c.POINTER(c.c_char_p), c.c_int, c.c_int, dp, P],
        "hx_shutdown": [P],
        "hx_setvar": [P, c.c_char_p, dp, c.c_int, c.c_char_p],
        "hx_getvar": [P, c.c_char_p, dp],
        "hx_split_biome_of": [P, c.c_char_p, c.c_int, c.POINTER(c.c_char_p), dp, dp, dp, dp, dp],
        "hx_create_biome": [P, c.c_char_p],
        "hx_delete_biome": [P, c.c_char_p],
        "hx_rename_biome": [P, c.c_char_p, c.c_char_p],
        "hx_split_biome": [P, c.c_int, c.POINTER(c.c_char_p), dp, dp, dp, dp, dp],
        "hx_set_outputs": [P, c.c_int, c.POINTER(c.c_char_p)],
        "hx_output_capabilities": [c.POINTER(c.POINTER(c.c_char_p)), c.POINTER(c.c_int)],
        "hx_set_member_sorting": [P, c.c_int],
        "hx_set_lane_calibration": [P, c.c_int],
        "hx_lanes_calibrated": [P, c.POINTER(c.c_int)],
        "hx_lane_order_source": [P, c.POINTER(c.c_int)],
        "hx_set_cost_model": [P, c.c_int],
        "hx_cost_models_export": [c.c_char_p, c.POINTER(c.c_int)],
        "hx_cost_models_load": [c.c_char_p, c.POINTER(c.c_int)],
        "hx_enable_history": [P, c.c_int],
        "hx_enable_spinup_record": [P, c.c_int],
        "hx_spinup_record": [P, c.c_int, c.POINTER(c.POINTER(c.c_char_p)), c.POINTER(c.c_int), dp, c.c_int,
                             c.POINTER(c.c_int)],
        "hx_tracking_pools": [P, c.POINTER(c.POINTER(c.c_char_p)), c.POINTER(c.c_int)],
        "hx_tracking_data": [P, c.c_int, c.c_int, c.c_int, dp, dp, c.POINTER(c.c_ulonglong)],
        "hx_var_info": [P, c.c_char_p, c.POINTER(c.c_char_p), c.POINTER(c.c_char_p)],
        "hx_biomes": [P, c.POINTER(c.POINTER(c.c_char_p)), c.POINTER(c.c_int)],
        "hx_setvar_dated_members": [P, c.c_char_p, c.POINTER(c.c_int), dp, c.c_int, c.c_char_p],
        "hx_unit_csys": [c.c_int, c.c_int, dp, dp, dp, c.c_double, dp],
        "hx_unit_doeclim_kernel": [c.c_int, c.c_double, c.c_int, dp],
        "hx_run_name": [P, c.POINTER(c.c_char_p)],
        "hx_halocarbons": [P, c.POINTER(c.POINTER(c.c_char_p)), c.POINTER(c.c_int)],
        "hx_setvar_dated": [P, c.c_char_p, c.POINTER(c.c_int), dp, c.c_int, c.c_char_p],
        "hx_lane_of_member": [P, c.POINTER(c.c_int)],
        "hx_reset": [P, c.c_double],
        "hx_run": [P, c.c_double],
        "hx_sync": [P],
        "hx_fetchvars": [P, c.c_char_p, c.c_int, c.c_int, dp],
        "hx_device_var": [P, c.c_char_p, c.POINTER(P), c.POINTER(c.c_int)],
        "hx_stats_device": [P, c.c_char_p, c.c_int, c.c_int, P],
        "hx_status": [P, c.POINTER(c.c_uint)],
        "hx_spinup_steps": [P, c.c_int, c.POINTER(c.c_int)],
        "hx_state_row": [P, c.c_int, dp],
        "hx_dates": [P, c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(c.c_int)],
        "hx_sizes": [P, c.POINTER(c.c_int), c.POINTER(c.c_int)],
        "hx_last_run_ms": [P, dp],
        "hx_last_spinup_ms": [P, dp],
        "hx_stream": [P, c.POINTER(P)],
        "hx_set_pair_kernel_limit": [P, c.c_int],
        "hx_set_two_wave_from": [P, c.c_int],
        "hx_set_prewarm": [P, c.c_int],
        "hx_last_run_prewarmed": [P, c.POINTER(c.c_int)],
        "hx_wave_clock": [P, c.c_int, c.POINTER(c.c_longlong), c.c_int, c.POINTER(c.c_int)],
        "hx_component_output": [P, c.c_char_p, c.POINTER(c.c_int)],
        "hx_last_run_kernel": [P, c.POINTER(c.c_char_p)],
        "hx_last_run_variant": [P, c.POINTER(c.c_int)],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c.c_int


ABI_SYMBOLS = ["hx_backend", "hx_build_info", "hx_last_error", "hx_newcore", "hx_shutdown", "hx_setvar",
               "hx_getvar", "hx_split_biome", "hx_split_biome_of", "hx_create_biome", "hx_delete_biome", "hx_rename_biome", "hx_set_outputs", "hx_output_capabilities",
               "hx_set_member_sorting", "hx_lane_of_member", "hx_enable_history", "hx_enable_spinup_record", "hx_spinup_record", "hx_setvar_dated", "hx_halocarbons", "hx_run_name", "hx_tracking_pools", "hx_tracking_data", "hx_var_info", "hx_biomes", "hx_setvar_dated_members", "hx_unit_csys", "hx_unit_doeclim_kernel", "hx_reset", "hx_run", "hx_sync", "hx_fetchvars", "hx_device_var",
               "hx_stats_device", "hx_status", "hx_spinup_steps", "hx_state_row", "hx_dates", "hx_sizes",
               "hx_last_run_ms", "hx_last_spinup_ms", "hx_stream", "hx_set_pair_kernel_limit", "hx_set_two_wave_from", "hx_wave_clock",
               "hx_last_run_kernel", "hx_last_run_variant", "hx_component_output", "hx_newcore_devices", "hx_shards",
               "hx_device_var_shard", "hx_stream_shard", "hx_comm_unique_id", "hx_comm_init_rank",
               "hx_comm_info", "hx_ensemble_stats", "hx_set_lane_calibration", "hx_lanes_calibrated", "hx_lane_order_source", "hx_set_cost_model",
               "hx_cost_models_export", "hx_cost_models_load", "hx_set_prewarm", "hx_last_run_prewarmed"]
