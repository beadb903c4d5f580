"""Loader for the hyphenated package directory `monte-carlo-path-tracing_amd/`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "monte-carlo-path-tracing_amd")


def load_package():
    """Import `monte-carlo-path-tracing_amd/` under the name `mcpt_amd`."""
    if "mcpt_amd" in sys.modules:
        return sys.modules["mcpt_amd"]
    spec = importlib.util.spec_from_file_location(
        "mcpt_amd", os.path.join(PKG_DIR, "__init__.py"),
        submodule_search_locations=[PKG_DIR])
    module = importlib.util.module_from_spec(spec)
    sys.modules["mcpt_amd"] = module
    spec.loader.exec_module(module)
    return module
