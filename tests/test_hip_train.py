"""GPU: the whole pipeline under a real optimizer (tools/train_synthetic.py): loader -> encoders -> unit -> head, the
reference's objective (cross-entropy + fused auxiliary losses), Adam.  The loss has to fall and the accuracy to leave chance:
a wrong sign or a missing term in any backward path shows up here even if it slipped through a tolerance."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synthetic_training_learns():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_synthetic.py"), "30"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "training check ok" in r.stdout
