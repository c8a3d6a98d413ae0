"""Import shim: loads the package directory `sdrpp-tetra-demodulator_amd/` (its name is not a valid
Python identifier) under the module name `sdrpp_tetra_demodulator_amd`."""
import importlib.util
import os
import sys

_NAME = "sdrpp_tetra_demodulator_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sdrpp-tetra-demodulator_amd")


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_DIR, "__init__.py"),
                                                  submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


pkg = load()
