"""K12 (occupancy variant), K16, K17 and the joint GNN -> CNN predictor against the CPU oracle."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

import gts
from gts import ops
from oracle import graph_ref, joint_ref
from tests.dataset_util import write_dataset

pytestmark = pytest.mark.gpu
DEV = "cuda"
BG = [[1.0, -1.0, -1.0, -1.0]]


@pytest.fixture(scope="module", autouse=True)
def _lib(hip_lib):
    assert torch.cuda.is_available()
    return hip_lib


def _volume(shape, n_nodes, seed, tumour=0.02):
    rng = np.random.default_rng(seed)
    svs = rng.integers(-1, n_nodes, size=shape).astype(np.int16)
    logits = rng.standard_normal((n_nodes, 4)).astype(np.float32)
    logits[:, 0] += np.where(rng.random(n_nodes) < tumour, -4.0, 4.0)      # mostly healthy nodes
    return svs, logits


@pytest.mark.parametrize("shape", [(18, 15, 13), (1, 1, 1), (5, 64, 3), (40, 33, 29)])
def test_argmax_projection_with_plane_flags_bit_exact(shape):
    svs, logits = _volume(shape, 50, seed=sum(shape))
    vox, flags = ops.project_argmax_occupancy(torch.from_numpy(svs).to(DEV), torch.from_numpy(logits).to(DEV))
    want = graph_ref.project_nodes_to_img_ref(svs, logits.argmax(1))
    assert np.array_equal(vox.cpu().numpy(), want)
    assert torch.equal(vox, ops.project_argmax(torch.from_numpy(svs).to(DEV), torch.from_numpy(logits).to(DEV)))
    mask = want != 0
    for got, axes in zip(flags, ((1, 2), (0, 2), (0, 1))):
        assert np.array_equal(got.cpu().numpy().astype(bool), mask.any(axis=axes))


def _box(shape, seed, device=DEV):
    rng = np.random.default_rng(seed)
    idx = [np.flatnonzero(rng.random(n) < 0.6) for n in shape]
    idx = [i if len(i) else np.array([0]) for i in idx]
    return ops.CropBox(*idx, shape, device)


@pytest.mark.parametrize("ci,ct", [(4, 4), (3, 2), (0, 4), (1, 5)])
def test_crop_concat_bit_exact(ci, ct):
    shape, n_nodes = (19, 14, 23), 41
    rng = np.random.default_rng(ci * 10 + ct)
    svs = rng.integers(-1, n_nodes, size=shape).astype(np.int16)
    table = rng.standard_normal((n_nodes, ct)).astype(np.float32)
    bg = rng.standard_normal((1, ct)).astype(np.float32)
    img = rng.standard_normal(shape + (ci,)).astype(np.float32)
    box = _box(shape, seed=3)
    voxel = torch.cat([torch.from_numpy(table), torch.from_numpy(bg)])[svs.astype(np.int64)]
    want = joint_ref.combine_logits_and_image_ref(voxel, torch.from_numpy(img), box.as_ix())
    got = ops.crop_concat(torch.from_numpy(img).to(DEV) if ci else None, torch.from_numpy(svs).to(DEV),
                          torch.from_numpy(table).to(DEV), torch.from_numpy(bg).to(DEV).reshape(-1), box)
    assert got.shape == (1, ci + ct, *box.shape) and got.is_contiguous()
    assert torch.equal(got.cpu(), want.contiguous())


def test_crop_concat_whole_volume_and_bad_boxes():
    shape = (6, 7, 8)
    svs, logits = _volume(shape, 9, seed=1)
    img = np.random.default_rng(0).standard_normal(shape + (4,)).astype(np.float32)
    box = ops.CropBox(np.arange(6), np.arange(7), np.arange(8), shape, DEV)
    got = ops.crop_concat(torch.from_numpy(img).to(DEV), torch.from_numpy(svs).to(DEV),
                          torch.from_numpy(logits).to(DEV), torch.tensor(BG[0], device=DEV), box)
    voxel = graph_ref.project_logits_to_img_ref(svs, logits).astype(np.float32)
    assert np.array_equal(got[0].cpu().numpy(), np.moveaxis(np.concatenate([img, voxel], -1), -1, 0))
    for bad in ((np.array([0, 6]), [0], [0]), ([1, 1], [0], [0]), ([2, 1], [0], [0]), ([-1], [0], [0])):
        with pytest.raises(gts.GtsError):
            ops.CropBox(*bad, shape, DEV)
    with pytest.raises(gts.GtsError):
        ops.crop_concat(None, torch.from_numpy(svs).to(DEV)[:5], torch.from_numpy(logits).to(DEV),
                        torch.tensor(BG[0], device=DEV), box)


@pytest.mark.parametrize("relabel", [False, True])
def test_argmax_scatter_bit_exact(relabel):
    shape = (21, 17, 12)
    box = _box(shape, seed=8)
    rng = np.random.default_rng(2)
    scores = rng.integers(-2, 3, size=(4, *box.shape)).astype(np.float32)      # plenty of ties: first maximum wins
    table = np.array([0, 2, 1, 4], dtype=np.int16)
    got = ops.argmax_scatter(torch.from_numpy(scores).to(DEV)[None], box,
                             torch.from_numpy(table).to(DEV) if relabel else None)
    want = np.zeros(shape, dtype=np.int16)
    labels = scores.argmax(axis=0)
    want[box.as_ix()] = table[labels] if relabel else labels
    assert got.dtype == torch.int16 and np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(labels, torch.argmax(torch.from_numpy(scores), dim=0).numpy())


def _nets(seed):
    from model.networks import CnnRefinementNet, init_graph_net
    from utils.hyperparam_helpers import EvalParamSet

    torch.manual_seed(seed)
    gnn = init_graph_net("GSpool", EvalParamSet(20, 4, [32, 32], None, None)).to(DEV).eval()
    cnn = CnnRefinementNet(8, 4, [16]).to(DEV).eval()
    return gnn, cnn


def test_predict_one_sample_matches_oracle(tmp_path):
    from data_processing.data_loader import ImageGraphDataset
    from scripts import generate_joint_predictions as joint
    from utils.hyperparam_helpers import DEFAULT_BACKGROUND_NODE_LOGITS

    data = str(tmp_path / "data") + "/"
    write_dataset(data, 3)
    with redirect_stdout(io.StringIO()):
        ds = ImageGraphDataset(data, "BraTS_", read_image=True, read_graph=True, read_label=False)
    gnn, cnn = _nets(0)
    ref_cnn = joint_ref.RefCnnRefinementNet(8, 4, [16])
    ref_cnn.load_state_dict({k: v.cpu() for k, v in cnn.state_dict().items()})
    for mri, graph, feats, img in ds:
        svs = ds.get_supervoxel_partitioning(mri)
        with redirect_stdout(io.StringIO()):
            got = joint.predict_one_sample(gnn, cnn, graph, feats, img, svs)
        with torch.no_grad():
            node_logits = gnn(graph.to(DEV), torch.FloatTensor(feats).to(DEV)).cpu().numpy()
        want, crop, cnn_in, refined = joint_ref.predict_one_sample_ref(
            node_logits, ref_cnn, np.ascontiguousarray(img), np.ascontiguousarray(svs),
            DEFAULT_BACKGROUND_NODE_LOGITS)
        assert got.dtype == np.int16 and got.shape == svs.shape
        # voxels whose refined top-2 margin exceeds the fp32 conv tolerance must agree exactly
        top2 = torch.topk(refined[0], 2, dim=0).values
        clear = np.zeros(svs.shape, dtype=bool)
        clear[crop] = ((top2[0] - top2[1]) > 1e-3).numpy()
        outside = np.ones(svs.shape, dtype=bool)
        outside[crop] = False
        assert clear.mean() > 0.5 * (~outside).mean()
        assert np.array_equal(got[clear], want[clear]) and not got[outside].any()
        # the K16 tensor itself is bit-exact, and MIOpen's convolutions agree with torch CPU within 1e-4
        box = ops.CropBox(*[c.reshape(-1) for c in crop], svs.shape, DEV)
        mine_in = ops.crop_concat(torch.from_numpy(np.ascontiguousarray(img)).to(DEV),
                                  torch.from_numpy(np.ascontiguousarray(svs)).to(DEV),
                                  torch.from_numpy(node_logits).to(DEV),
                                  torch.tensor(DEFAULT_BACKGROUND_NODE_LOGITS, device=DEV).reshape(-1), box)
        assert torch.equal(mine_in.cpu(), cnn_in.contiguous())
        with torch.no_grad():
            assert torch.allclose(cnn(mine_in).cpu(), refined, rtol=1e-4, atol=1e-4)


def test_joint_cli_writes_brats_volumes(tmp_path):
    from data_processing import nifti_io
    from data_processing.data_loader import ImageGraphDataset
    from data_processing.image_processing import uncrop_to_brats_size
    from data_processing.labels import swap_labels_to_brats
    from model.networks import CnnRefinementNet, init_graph_net
    from scripts import generate_joint_predictions as joint
    from utils.hyperparam_helpers import EvalParamSet

    data, out = str(tmp_path / "data") + "/", str(tmp_path / "out")
    write_dataset(data, 2)
    torch.manual_seed(1)
    gnn = init_graph_net("GSpool", EvalParamSet(20, 4, [256] * 4, None, None))
    cnn = CnnRefinementNet(8, 4, [16])
    torch.save(gnn.state_dict(), tmp_path / "gnn.pt")
    torch.save(cnn.state_dict(), tmp_path / "cnn.pt")
    with redirect_stdout(io.StringIO()) as log:
        joint.main(["-d", data, "-p", "BraTS_", "-o", out, "-g", str(tmp_path / "gnn.pt"),
                    "-c", str(tmp_path / "cnn.pt")])
        ds = ImageGraphDataset(data, "BraTS_", read_image=True, read_graph=True, read_label=False)
        gnn_d, cnn_d = joint.load_nets("GSpool", str(tmp_path / "gnn.pt"), str(tmp_path / "cnn.pt"))
        for mri, graph, feats, img in ds:
            saved = nifti_io.read_nifti(os.path.join(out, f"{mri}.nii.gz"), np.int16)
            pred = joint.predict_one_sample(gnn_d, cnn_d, graph, feats, img, ds.get_supervoxel_partitioning(mri))
            assert saved.shape == (240, 240, 155) and set(np.unique(saved)) <= {0, 1, 2, 4}
            assert np.array_equal(saved, swap_labels_to_brats(uncrop_to_brats_size(ds.get_crop(mri), pred)))
    assert "Finished saving predictions" in log.getvalue()
