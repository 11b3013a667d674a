"""Import helper: the package directory `slam-localization_amd/` carries a hyphen (it mirrors the
reference repository's name), so it cannot be imported with a plain `import`.  This registers it
under the module name `slam_localization_amd`.

    from slkpkg import slk            # slam_localization_amd.slk
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "slam-localization_amd")
_NAME = "slam_localization_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


pkg = load()
slk = pkg.slk
