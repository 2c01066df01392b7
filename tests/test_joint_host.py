"""Host logic of the joint GNN -> CNN predictor: the crop from per-axis plane flags equals the
reference's dilate-then-any rule on arbitrary volumes (fixture-pinned in test_reference_golden)."""
import io
from contextlib import redirect_stdout

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from data_processing import image_processing
from oracle import joint_ref


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 9), st.integers(1, 9), st.integers(1, 9), st.floats(0.0, 0.3), st.integers(0, 2 ** 31 - 1))
def test_plane_flag_route_equals_dilated_mask_route(x, y, z, density, seed):
    rng = np.random.default_rng(seed)
    preds = (rng.random((x, y, z)) < density) * rng.integers(1, 4, (x, y, z))
    mask = preds != 0
    with redirect_stdout(io.StringIO()):
        got = image_processing.tumor_crop_from_plane_flags(mask.any(axis=(1, 2)), mask.any(axis=(0, 2)),
                                                           mask.any(axis=(0, 1)))
        want = joint_ref.determine_tumor_crop_ref(preds)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))


def test_cnn_refinement_net_checkpoint_layout():
    import torch

    from model.networks import CnnRefinementNet

    torch.manual_seed(0)
    mine, ref = CnnRefinementNet(8, 4, [16]), joint_ref.RefCnnRefinementNet(8, 4, [16])
    assert list(mine.state_dict()) == list(ref.state_dict()) == [
        "conv_layers.0.weight", "conv_layers.0.bias", "conv_layers.1.weight", "conv_layers.1.bias"]
    ref.load_state_dict(mine.state_dict())
    x = torch.randn(1, 8, 6, 5, 7)
    with torch.no_grad():
        assert torch.equal(mine(x), ref(x))      # same torch modules, same arithmetic on the CPU
