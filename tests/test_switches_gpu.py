"""A/B switches that select another launch shape of a default kernel stay what the measurements were taken on: the fast blend's oracle tests
(tests/test_fast_blend_gpu.py: image, integers, gradients) are run again in a subprocess with the record-parallel backward in workgroups of two and of
four waves (`GSR_RP_WAVES_PER_WG`, DESIGN.md 7.6: the default is one wave per workgroup; the lists are walked from their ends in every shape), and the
forward's -- integers and lists bit-exact -- with the rank passes' chunks cut the two other ways a frame can get them (balanced by weight on
every frame, `GSR_RANK_BALANCED=2`; dealt round-robin in groups of eight, `GSR_RANK_ILV=8`; the small test scenes never report an uneven frame)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.fast_blend]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(200)
@pytest.mark.parametrize("waves", [2, 4])
def test_backward_oracle_tests_with_wider_backward_workgroups(waves):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, GSR_RP_WAVES_PER_WG=str(waves), PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_fast_blend_gpu.py"),
           "-k", "backward_vs_oracle"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=190)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1500:]


@pytest.mark.timeout(280)
@pytest.mark.parametrize("env", [{"GSR_RANK_BALANCED": "2"}, {"GSR_RANK_ILV": "8"}])
def test_forward_oracle_tests_with_the_other_rank_pass_chunkings(env):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    e = dict(os.environ, PYTHONPATH=ROOT, **env)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_fast_blend_gpu.py"),
           os.path.join(ROOT, "tests", "test_fullsize_gpu.py"), "-k", "forward_vs_oracle or benchmarked_step"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=270)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1500:]
