"""Import helper: the product package directory is ``compute-engine_amd`` (hyphen)."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
amd = importlib.import_module("compute-engine_amd")
