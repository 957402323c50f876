"""Drop-in alias: ``lib.*`` resolves to the ``dreamer_amd`` modules of the same name, so a
``main.py`` written against the reference's package layout (``from lib.config import Config``,
``from lib.trainer import train``) runs unchanged with this repository ahead of it on sys.path."""
import importlib
import sys

for _name in ("config", "utils", "world_model", "actor", "critic", "replay_buffer", "trainer"):
    _mod = importlib.import_module(f"dreamer_amd.{_name}")
    sys.modules[f"{__name__}.{_name}"] = _mod
    setattr(sys.modules[__name__], _name, _mod)
