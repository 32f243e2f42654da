"""`mgnet` -- the reference's import name, served by mgnet_amd.

`from mgnet import add_mgnet_config`, `mgnet.modeling`, `mgnet.geometry`, `mgnet.data`, `mgnet.solver`,
`mgnet.postprocessing`, `mgnet.evaluation` (mgnet/__init__.py:1-3 and the packages beside it) resolve to the modules of
`mgnet_amd` themselves (the same module objects, not copies), so registries, configs and dotted names inside yaml files
(`mgnet.data....`) written for the reference keep working."""
import importlib
import importlib.abc
import importlib.util
import sys

import mgnet_amd as _impl
from mgnet_amd import add_mgnet_config, get_cfg  # noqa: F401

__version__ = _impl.__version__


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """mgnet.<x> -> mgnet_amd.<x>"""

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("mgnet."):
            return None
        real = "mgnet_amd." + fullname[len("mgnet."):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, self)

    def create_module(self, spec):
        real = importlib.import_module("mgnet_amd." + spec.name[len("mgnet."):])
        self._real_spec = getattr(self, "_real_spec", {})
        self._real_spec[spec.name] = (real.__spec__, real.__loader__)
        return real

    def exec_module(self, module):
        # the import system has just stamped the alias spec on the shared module object: put the real one back
        for alias, (spec, loader) in list(self._real_spec.items()):
            if spec is not None and spec.name == module.__name__:
                module.__spec__, module.__loader__ = spec, loader
                del self._real_spec[alias]


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

from mgnet import config, data, evaluation, geometry, modeling, postprocessing, solver  # noqa: E402,F401
