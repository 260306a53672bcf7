"""Console entry point: `fithic ...` (pyproject.toml) and `python -m fithic_amd ...` both land here - the counterpart of the
reference's `fithic = fithic.fithic:main` (/root/reference/setup.py:14-16)."""
import sys

from . import _loader


def _device_of(argv):
    for i, a in enumerate(argv):
        if a == "--device" and i + 1 < len(argv) and argv[i + 1].isdigit():
            return int(argv[i + 1])
        if a.startswith("--device=") and a[9:].isdigit():
            return int(a[9:])
    return 0


def main():
    # the library and the HIP runtime come up on a thread while numpy and the package are imported (fithic_amd/_loader.py)
    if "--gpus" not in sys.argv and not any(a.startswith("--gpus=") for a in sys.argv):
        _loader.warm_in_background(_device_of(sys.argv))
    from .cli import main as cli_main
    return cli_main()
