"""Import alias for the package directory `geostatinversion.jl_amd/` (its name has a dot, so a plain
`import` cannot reach it).  `import gsi_amd` returns that package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "geostatinversion.jl_amd")
_spec = importlib.util.spec_from_file_location("gsi_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gsi_amd"] = _mod
_spec.loader.exec_module(_mod)
