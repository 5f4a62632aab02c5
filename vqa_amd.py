"""Importable alias for the package directory ``visual-question-answering_amd/`` (the hyphen
makes the directory name itself not importable): ``import vqa_amd`` loads that package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "visual-question-answering_amd")
_spec = importlib.util.spec_from_file_location("vqa_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vqa_amd"] = _mod
_spec.loader.exec_module(_mod)
