"""f1 experiment (profiles/r03/f1): the two-squarings-per-launch forward (ADVCHAIN_FUSE2) against one launch per squaring.
The knob is read once per process, so both settings run tools/ab/fuse2_ab.py in a subprocess."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("shape", ["81", "42"])
def test_fused_pairs_equal_the_per_squaring_launches(shape, tmp_path):
    ref = str(tmp_path / "ref.pt")
    env = {k: v for k, v in os.environ.items() if not k.startswith("ADVCHAIN_FUSE2")}
    subprocess.run([sys.executable, os.path.join(ROOT, "tools/ab/fuse2_ab.py"), "--batch", "2", "--save", ref], check=True, env=env,
                   stdout=subprocess.DEVNULL)
    out = str(tmp_path / "fused.pt")
    env2 = dict(env, ADVCHAIN_FUSE2="2", ADVCHAIN_FUSE2_SHAPE=shape)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools/ab/fuse2_ab.py"), "--batch", "2", "--save", out], check=True, env=env2,
                   stdout=subprocess.DEVNULL)
    a, b = torch.load(ref), torch.load(out)
    assert torch.equal(a["fields"][0], b["fields"][0])          # level 1: the same arithmetic
    # level 2 is compiled in another context (fma contraction): 1-2 ulp, doubled by every further squaring
    for m in range(1, a["fields"].shape[0]):
        assert float((a["fields"][m] - b["fields"][m]).abs().max()) < 2e-7 * 2 ** m, m
    assert float((a["pos"] - b["pos"]).abs().max()) < 5e-5
    assert float((a["disp"] - b["disp"]).abs().max()) < 1e-3
