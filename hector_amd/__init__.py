"""hector_amd -- MI355X-native ensemble integrator for Hector's coupled
carbon-cycle / climate year loop.  HIP kernels + C ABI in
hector_amd/csrc; this package only binds them."""
from ._lib import HectorAmdError, DEFAULT_SCENARIO, DEFAULT_LIB  # noqa: F401
from .core import (Core, newcore, run, reset, shutdown, setvar, fetchvars,  # noqa: F401
                   split_biome, get_tracking_data, create_biome, rename_biome,
                   get_biome_inits, sendmessage, GETDATA, SETDATA, isactive, startdate,
                   enddate, getdate, getname, get_biome_list, getunits, getfxn, runscenario)
from . import capabilities  # noqa: F401

__version__ = "0.1.0"


def build_info(lib_path=None):
    """hx_build_info(): the HIP the native library was built with and the runtime it runs on."""
    from . import _lib
    return _lib.load(lib_path).hx_build_info().decode()
