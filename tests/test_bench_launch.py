"""bench.py's launch convention on a box without enough GPUs: one JSON diagnostic line, never a traceback (CPU-only checks;
the self-launched RCCL run itself is tests/test_gpu_bench.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), capture_output=True, text=True, env=env, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r, lines


def _visible():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("gpus", [2, 8])
def test_too_few_gpus_is_a_one_line_json_diagnostic(gpus):
    if _visible() >= gpus:
        pytest.skip("this box has the GPUs")
    r, lines = _bench("--gpus", str(gpus))
    assert r.returncode == 2 and len(lines) == 1, (r.returncode, r.stdout, r.stderr[-2000:])
    out = json.loads(lines[0])
    assert out["value"] is None and out["n_gpus"] == gpus and out["visible_devices"] == _visible()
    assert "Traceback" not in r.stderr


def test_no_gpu_at_all_is_a_diagnostic_too():
    if _visible() >= 1:
        pytest.skip("this box has a GPU")
    for extra in ({}, {"FHX_FORCE_DIST": "1"}):
        r, lines = _bench("--gpus", "1", env_extra=extra)
        assert r.returncode == 2 and len(lines) == 1
        assert json.loads(lines[0])["value"] is None and "Traceback" not in r.stderr
