"""Opt-in kernels that have not been validated on hardware yet run only with CAT_EXPERIMENTAL=1 (each in its own process, because
the library reads its opt-in switches once)."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get('CAT_EXPERIMENTAL') != '1', reason='set CAT_EXPERIMENTAL=1')]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lds_tile_conv_forward():
    env = dict(os.environ, CAT_CONV_TILE='2')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'debug', 'check_conv_tile.py')], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
