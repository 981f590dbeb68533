"""The product's DEFAULT mode (fast blend) through the paths the other suites exercise in the exact mode: recorded steps, frame lanes,
the bound / leaves entries, a second backward through one graph, side streams.  These tests compare the rasterizer with ITSELF (eager
against replay, bound entry against accessors, first against second backward), so their bit-equalities hold in either blend mode; the
functions are the other suites', run here under the `fast_blend` marker (tests/conftest.py selects the exact kernels for everything else)."""
import pytest

from tests import test_fullsize_gpu as FS
from tests import test_graph_gpu as G
from tests import test_render_paths_gpu as RP

pytestmark = [pytest.mark.gpu, pytest.mark.fast_blend]


def test_mode_is_the_default():
    from gaussianavatars_amd import rasterizer as R

    assert R._fast_blend == 1


def test_recorded_step_equals_eager_step():
    G.test_graphed_step_equals_the_eager_step_frame_by_frame()


def test_overflow_reported_and_recovered():
    G.test_overflowing_frame_is_reported_and_recapture_recovers()


def test_sticky_overflow():
    G.test_an_overflow_in_an_earlier_replay_stays_reported()


def test_recorded_lanes():
    G.test_recorded_lanes_on_separate_streams_do_not_disturb_each_other()


def test_recorded_step_with_an_optimiser():
    G.test_recorded_step_in_a_training_loop_with_an_optimiser()


@pytest.mark.parametrize("N", [30003])
def test_bound_entry_equals_accessors(N):
    RP.test_bound_entry_equals_accessors_plus_rasterizer(N)


def test_bound_entry_with_band_ranks():
    RP.test_bound_entry_on_a_frame_large_enough_for_band_ranks()


def test_unbound_leaves_entry():
    RP.test_unbound_leaves_entry_equals_torch_activations_plus_rasterizer()


def test_every_splat_pruned():
    RP.test_bound_entry_with_every_splat_pruned()


def test_second_backward_through_the_same_graph():
    FS.test_second_backward_through_the_same_graph()


def test_side_stream_and_interleaved_forwards():
    FS.test_non_default_stream_and_interleaved_forwards()
