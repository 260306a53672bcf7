"""The engine and PyTorch share one HIP runtime whichever is loaded first (fithic_amd/_loader.py:_share_torch_hip_runtime)."""
import subprocess
import sys
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ENGINE_FIRST = """
import numpy as np
from fithic_amd import _capi
ctx = _capi.Context(0)
q = ctx.bh_array(np.array([0.03, 0.4, 0.7, 0.01]), 10.0)
import torch
t = torch.arange(10, device="cuda").sum().item()
g = torch.Generator(device="cuda")
assert t == 45 and abs(q[3] - 0.1) < 1e-15
print("ok")
"""

TORCH_FIRST = """
import torch
t = torch.arange(10, device="cuda").sum().item()
import numpy as np
from fithic_amd import _capi
ctx = _capi.Context(0)
q = ctx.bh_array(np.array([0.03, 0.4, 0.7, 0.01]), 10.0)
assert t == 45 and abs(q[3] - 0.1) < 1e-15
print("ok")
"""


@pytest.mark.parametrize("code", [ENGINE_FIRST, TORCH_FIRST], ids=["engine_first", "torch_first"])
def test_engine_and_torch_coexist_in_either_import_order(code):
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
