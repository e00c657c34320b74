"""CPU, build container only: `python oracle/make_golden.py` (no arguments, ONE command) regenerates every fixture under tests/golden/
byte for byte from the imported reference (/root/reference through oracle/ref_shims.py) and HF transformers.  This is what pins the
oracle: if the recipe, a shim or a generator drifts, the committed vectors no longer come from the reference.  Skipped where the
reference tree is absent (the GPU box)."""
import filecmp
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present (golden generation runs in the build container)")
def test_one_command_regenerates_every_golden_fixture_byte_for_byte(tmp_path):
    pytest.importorskip("transformers")
    env = dict(os.environ, SWAT_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py")], cwd=str(tmp_path), env=env, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    gold = os.path.join(ROOT, "tests", "golden")
    committed = sorted(os.listdir(gold))
    assert sorted(os.listdir(tmp_path)) == committed
    for name in committed:
        assert filecmp.cmp(os.path.join(gold, name), os.path.join(tmp_path, name), shallow=False), f"{name} differs from the committed fixture"
