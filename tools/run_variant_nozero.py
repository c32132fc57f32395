import sys; sys.path.insert(0,'.')
from graphaudio_amd import _capi
_capi.use_library("tools/variants/nozero.so")
import pytest
sys.exit(pytest.main(["tests/test_gpu_fuzz.py::test_session_42867_with_formulation_d_forced","tests/test_gpu_coarse.py","-q","-k","42867 or onset"]))
