"""Import alias: the package lives in ``gnn-builder_amd/`` (not a valid Python identifier), so
``import gnnbuilder_amd`` resolves its submodules there."""
from pathlib import Path as _Path

_real = _Path(__file__).resolve().parent.parent / "gnn-builder_amd"
__path__ = [str(_real)]
exec(compile((_real / "__init__.py").read_text(), str(_real / "__init__.py"), "exec"))
