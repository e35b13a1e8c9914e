"""Import shim: the package directory is `voiceconversion.jl_amd/` (dot in the name), which Python's import
system cannot address directly.  `import voiceconversion_jl_amd` loads that directory as a package under this
name; submodules work as usual (`import voiceconversion_jl_amd.dtw`)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "voiceconversion.jl_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
