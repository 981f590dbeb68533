"""The forward blend's CONTINUATION of deep quadrant walks (csrc/gsr_forward.hip: k_render<true, 1 / 2>) is off by default -- measured
slower than the lone walk on both bench scenes -- and selected per process by GSR_CONT_CHUNKS / GSR_CONT_MODE.  So that the code behind the switch
stays what the measurements were taken on, the fast blend's own oracle tests are run again in a subprocess with it ON: the hand-over already at entry
60 (every quadrant deeper than one chunk is parked and finished four chunks at a time, on the small scenes too), in both placements, plus the
hand-over the measurements used (entry 180) -- on `deep_stack`, `dense_tile_xl` and the other scenes of tests/test_fast_blend_gpu.py (image with
check_image's constants, integers bit-exact, gradients: the backward walks what the continuation left in final_T / n_contrib / the checkpoints) and on
the two full-size benchmark frames (tests/test_fullsize_gpu.py: ellipsoid and template-like, leaf and FLAME-row gradients against the oracle)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.fast_blend]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(290)
@pytest.mark.parametrize("mode,chunks,full_size", [(1, 1, False), (2, 1, False), (1, 3, True)])
def test_fast_blend_oracle_tests_with_the_continuation_on(mode, chunks, full_size):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, GSR_CONT_CHUNKS=str(chunks), GSR_CONT_MODE=str(mode), PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_fast_blend_gpu.py")]
    if full_size:
        cmd.append(os.path.join(ROOT, "tests", "test_fullsize_gpu.py") + "::test_config3_benchmarked_step_in_the_benchmarked_mode")
    cmd += ["-k", "forward_vs_oracle or backward_vs_oracle or benchmarked_step"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1500:]
