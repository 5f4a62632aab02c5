import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests/` here."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_report_header(config):
    """Which GPU ran the suite (a result that differs box to box can then be tied to a device: DESIGN.md section 4)."""
    import torch

    if not torch.cuda.is_available():
        return "gpu: none visible"
    p = torch.cuda.get_device_properties(0)
    return "gpu: %s uuid %s (%d CUs)" % (p.name, getattr(p, "uuid", "unknown"), p.multi_processor_count)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The same line at the END of the run: `-q` drops the header, and a failure that comes and goes between boxes of the
    pool is only worth anything with the device it happened on."""
    import torch

    if torch.cuda.is_available():
        p = torch.cuda.get_device_properties(0)
        terminalreporter.write_line("gpu: %s uuid %s (%d CUs), exit status %s" % (p.name, getattr(p, "uuid", "unknown"),
                                                                                   p.multi_processor_count, int(exitstatus)))
