import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# PyTorch-ROCm wheels bundle their own copy of the HIP / HSA runtime.  When libgraphaudio_hip.so (linked against /opt/rocm) has
# initialised the GPU first, a torch imported LATER brings a second runtime into the process and reports "no ROCm-capable
# device".  Loaded the other way round both use torch's copy and coexist (this is also the order bench.py uses).  Tests that
# hand torch tensors to the library (tests/test_gpu_device_out.py) therefore need torch in the process before any context.
try:
    import torch  # noqa: F401
except Exception:  # torch is optional for everything else
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The figures the at-size tests measured (tests/_report.py), printed with -q too and written next to the GPU run's other
    outputs when that directory exists."""
    from tests._report import FIGURES
    if not FIGURES:
        return
    terminalreporter.section("parity figures measured by this run")
    for line in FIGURES:
        terminalreporter.write_line(line)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        try:
            import json
            with open(os.path.join(out_dir, "parity_figures.json"), "w") as f:
                json.dump(FIGURES, f, indent=1)
        except OSError:
            pass
