"""ImageGraphDataset + collate on a synthetic on-disk dataset (host side, CPU)."""
import os

import numpy as np
import pytest
import torch

from data_processing.data_loader import ImageGraphDataset, minibatch_graphs
from oracle import graph_ref
from tests.dataset_util import write_dataset


def test_dataset_items_and_collate(tmp_path):
    root = str(tmp_path) + "/"
    truth = write_dataset(root, 3)
    ds = ImageGraphDataset(root, "BraTS_", read_image=False, read_graph=True, read_label=True)
    assert len(ds) == 3 and sorted(ds.all_ids) == sorted(truth)
    mri_id, g, feats, labels = ds[0]
    svs, node_labels = truth[mri_id]
    assert feats.dtype == np.float64 and feats.shape == (g.n, 20)
    assert np.array_equal(labels, node_labels)
    assert g.ndata["norm"].shape == (g.n, 1)
    assert np.array_equal(g.ndata["norm"].numpy(), graph_ref.norm_ref(graph_ref.RefGraph(g.src, g.dst, g.n)))
    assert g.min_in_degree >= 1                               # self loops
    assert np.array_equal(ds.get_supervoxel_partitioning(mri_id), svs)
    assert ds.get_voxel_labels(mri_id).dtype == np.int16
    assert len(ds.get_crop(mri_id)) == 3
    again = ds[0]
    assert again[1] is g                                      # parsed once, cached
    ids, bg, bf, bl = minibatch_graphs([ds[i] for i in range(3)])
    assert bg.n == sum(ds[i][1].n for i in range(3)) and bg.batch_size == 3
    assert bf.dtype == torch.float32 and bl.dtype == torch.int64 and bf.shape == (bg.n, 20)
    unlabeled = ImageGraphDataset(root, "BraTS_", read_image=False, read_graph=True, read_label=False)
    assert len(unlabeled[0]) == 3
    both = ImageGraphDataset(root, "BraTS_", read_image=True, read_graph=True, read_label=True)
    item = both[0]
    assert len(item) == 6 and item[4].shape == (24, 20, 16, 4) and item[4].dtype == np.float32


def test_binary_graph_image_cache(tmp_path, monkeypatch):
    from data_processing import graph_io

    root = str(tmp_path / "data") + "/"
    write_dataset(root, 2)
    cache = str(tmp_path / "cache")
    first = ImageGraphDataset(root, "BraTS_", read_image=False, cache_graphs=False, cache_dir=cache)
    a = first[0]
    calls = []
    monkeypatch.setattr(graph_io, "load_networkx_graph", lambda fp: calls.append(fp) or (_ for _ in ()).throw(AssertionError))
    second = ImageGraphDataset(root, "BraTS_", read_image=False, cache_graphs=False, cache_dir=cache)
    b = second[0]                                             # served from the image: no JSON parse
    assert not calls and a[0] == b[0]
    for name in ("indptr", "indices", "t_indptr", "t_indices", "t_slot", "t_pos", "src", "dst"):
        assert np.array_equal(getattr(a[1], name), getattr(b[1], name)), name
    assert np.array_equal(a[2], b[2]) and a[2].dtype == b[2].dtype and np.array_equal(a[3], b[3])
    assert torch.equal(a[1].ndata["norm"], b[1].ndata["norm"])


def test_crop_files_load_without_the_general_pickle_loader(tmp_path):
    """The reference writes `_crop.npy` as a pickled object array (scripts/preprocess_dataset.py:130).
    It is read through an arrays-only unpickler; a plain `.npz` triple is preferred when present; a
    file that smuggles any other global is refused."""
    import pickle

    from data_processing.data_loader import load_crop, save_crop

    mask = np.zeros((30, 20, 10), dtype=bool)
    mask[3:9, 4:15, 2:7] = True
    crop = np.ix_(mask.any(axis=(1, 2)), mask.any(axis=(0, 2)), mask.any(axis=(0, 1)))
    legacy = str(tmp_path / "a_crop.npy")
    np.save(legacy, np.array(crop, dtype=object), allow_pickle=True)
    got = load_crop(legacy)
    assert len(got) == 3 and all(np.array_equal(a, b) for a, b in zip(got, crop))
    assert np.zeros((30, 20, 10))[got].shape == (6, 11, 5)
    save_crop(str(tmp_path / "b_crop.npz"), crop)
    got = load_crop(str(tmp_path / "b_crop.npy"))                   # only the .npz exists
    assert all(np.array_equal(a, b) for a, b in zip(got, crop))

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))

    bad = str(tmp_path / "c_crop.npy")
    with open(bad, "wb") as fh:
        np.lib.format.write_array_header_1_0(fh, {"descr": "|O", "fortran_order": False, "shape": (3,)})
        pickle.dump([Evil(), 1, 2], fh)
    with pytest.raises(ValueError, match="refusing"):
        load_crop(bad)
