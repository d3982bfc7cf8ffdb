"""GPU: a short run of the randomised parity sweep (tools/fuzz_gpu.py): random single convs (channel mixes, odd sizes,
upsample+concat, activations, both precisions, planar output), conv backwards, random networks (every norm / pool / interp
mix, random feature taps), registration-feature kernels and projection heads against the CPU references."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_sweep_has_no_failures():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_gpu.py"), "25", "7"], capture_output=True, text=True,
                         timeout=300)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:]
    assert out.returncode == 0 and " 0 failures" in tail, out.stdout[-2000:] + out.stderr[-2000:]
