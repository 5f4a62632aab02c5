"""TEST INFRASTRUCTURE ONLY -- import the reference's ``model.py`` in THIS container.

``/root/reference`` does not exist on the GPU box, so nothing that runs there (``-m gpu``
tests, ``smoke()``, ``bench.py``) may call this.  It is used by ``make_golden.py`` and by
the container-only tests in ``tests/test_oracle_vs_reference.py`` (skipped when the
reference is absent).

model.py:4 imports ``torchvision`` which is not installed; an empty stub module (plus a
``vgg11_bn`` factory with torchvision's published cfg-"A"+BatchNorm layer list, used only
when a full net is built) lets it import.  Nothing of the reference is copied.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_DIR = "/root/reference"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_DIR, "model.py"))


def _vgg11_bn_factory(pretrained=False, **_kw):
    """torchvision's vgg11_bn topology: cfg A = [64,M,128,M,256,256,M,512,512,M,512,512,M],
    each conv 3x3/pad 1 followed by BatchNorm2d + ReLU(inplace); only ``.features`` is used by
    the reference (model.py:236)."""
    import torch.nn as nn

    cfg = [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"]
    layers, c_in = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c_in, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
            c_in = v
    holder = nn.Module()
    holder.features = nn.Sequential(*layers)
    return holder


def import_reference_model():
    """Returns the reference's ``model`` module (cached in sys.modules as ``_vqa_ref_model``)."""
    if "_vqa_ref_model" in sys.modules:
        return sys.modules["_vqa_ref_model"]
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_DIR)
    sys.dont_write_bytecode = True          # the reference tree is read-only
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvm.vgg11_bn = _vgg11_bn_factory
        tv.models = tvm
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tvm
    import importlib.util

    spec = importlib.util.spec_from_file_location("_vqa_ref_model", os.path.join(REFERENCE_DIR, "model.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["_vqa_ref_model"] = mod
    spec.loader.exec_module(mod)
    return mod
