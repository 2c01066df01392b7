"""Dataset of preprocessed MRIs and the mini-batch collate function for the GNN path.

Counterpart of /root/reference/data_processing/data_loader.py:38-115,165-169.  Same class
name, constructor flags, item layout `(mri_id, graph, features[, labels])` and collate
output `(ids, batched_graph, FloatTensor, LongTensor)`; the graph objects are `gts.Graph`
(int32 CSR, built once per sample) instead of DGLGraphs.

Two additions (the reference re-parses JSON -> networkx -> graph for every sample of every
epoch, which would starve the GPU path long before its kernels matter):
  * parsed samples are cached in memory (`cache_graphs=True`);
  * `cache_dir=...` keeps a binary image of every parsed sample (`{id}.gts.npz`: COO edge
    list in DGL edge order, features, labels) so later runs skip JSON/networkx entirely.
    The image is rebuilt whenever the JSON file is newer.
"""
import glob
import os

import numpy as np
import torch

import gts
from data_processing import graph_io, nifti_io


class ImageGraphDataset(torch.utils.data.Dataset):
    """Iterates the sample folders written by the preprocessing script: each holds
    `{id}_nxgraph.json`, `{id}_supervoxels.nii.gz`, `{id}_input.nii.gz`, optionally
    `{id}_label.nii.gz` and `{id}_crop.npy`."""

    def __init__(self, dataset_root_dir, mri_start_string, read_image=True, read_graph=True,
                 read_label=True, cache_graphs=True, cache_dir=None):
        self.dataset_root_dir = dataset_root_dir
        self.all_ids = self.get_all_mris_in_dataset(dataset_root_dir, mri_start_string)
        self.read_image = read_image
        self.read_graph = read_graph
        self.read_label = read_label
        assert self.read_graph or self.read_image
        self._graph_cache = {} if cache_graphs else None
        self.cache_dir = cache_dir
        if cache_dir is not None:
            os.makedirs(cache_dir, exist_ok=True)

    def get_all_mris_in_dataset(self, dataset_root_dir, mri_start_string):
        folders = glob.glob(f"{dataset_root_dir}**/{mri_start_string}*/", recursive=True)
        print(f"Found {len(folders)} MRIs")
        return [fp.split(os.sep)[-2] for fp in folders]

    def _path(self, mri_id, suffix):
        return f"{self.dataset_root_dir}{os.sep}{mri_id}{os.sep}{mri_id}{suffix}"

    def sample_costs(self):
        """One cost per sample for size-aware sharding under data parallelism (gts.dist.shard_indices): the bytes of
        its node-link JSON — proportional to nodes + edges, known without parsing anything, the same on every rank."""
        return [os.path.getsize(self._path(mri_id, "_nxgraph.json")) if self.read_graph else 1 for mri_id in self.all_ids]

    def get_one(self, mri_id):
        parts = []
        if self.read_graph:
            parts += self.get_graph(mri_id)
        if self.read_image:
            parts += self.get_image(mri_id)
        return (mri_id, *parts)

    def get_graph(self, mri_id):
        """networkx JSON -> (graph, features [N,F] float64, labels [N]) ; graph.ndata['norm'] =
        in_degree^-0.5 with inf -> 0, shape [N,1] (reference data_loader.py:67-83)."""
        if self._graph_cache is not None and mri_id in self._graph_cache:
            return list(self._graph_cache[mri_id])
        json_path = self._path(mri_id, "_nxgraph.json")
        image = self._load_image(mri_id, json_path)
        if image is not None:
            G, features, labels = image
        else:
            nx_graph = graph_io.load_networkx_graph(json_path)
            features = np.array([nx_graph.nodes[n]["features"] for n in nx_graph.nodes])
            labels = np.array([nx_graph.nodes[n]["label"] for n in nx_graph.nodes]) \
                if all("label" in nx_graph.nodes[n] for n in nx_graph.nodes) else None
            G = gts.from_networkx(nx_graph)
            self._save_image(mri_id, G, features, labels)
        norm = torch.pow(G.in_degrees().float(), -0.5)
        norm[torch.isinf(norm)] = 0
        G.ndata["norm"] = norm.unsqueeze(1)
        item = [G, features]
        if self.read_label:
            if labels is None:
                raise KeyError(f"{mri_id}: graph nodes carry no 'label' but read_label=True")
            item.append(labels)
        if self._graph_cache is not None:
            self._graph_cache[mri_id] = tuple(item)
        return item

    def _image_path(self, mri_id):
        return os.path.join(self.cache_dir, f"{mri_id}.gts.npz")

    def _load_image(self, mri_id, json_path):
        if self.cache_dir is None:
            return None
        path = self._image_path(mri_id)
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(json_path):
            return None
        with np.load(path, allow_pickle=False) as z:
            G = gts.Graph(z["src"], z["dst"], int(z["num_nodes"]))
            labels = z["labels"] if "labels" in z.files else None
            return G, z["features"], labels

    def _save_image(self, mri_id, G, features, labels):
        if self.cache_dir is None:
            return
        arrays = dict(src=G.src, dst=G.dst, num_nodes=np.int64(G.n), features=features)
        if labels is not None:
            arrays["labels"] = labels
        tmp = self._image_path(mri_id) + ".tmp.npz"
        np.savez(tmp, **arrays)
        os.replace(tmp, self._image_path(mri_id))

    def get_voxel_labels(self, mri_id):
        return nifti_io.read_nifti(self._path(mri_id, "_label.nii.gz"), np.int16)

    def get_image(self, mri_id):
        img = nifti_io.read_nifti(self._path(mri_id, "_input.nii.gz"), np.float32)
        return [img, self.get_voxel_labels(mri_id)] if self.read_label else [img]

    def get_supervoxel_partitioning(self, mri_id):
        return nifti_io.read_nifti(self._path(mri_id, "_supervoxels.nii.gz"), np.int16)

    def get_crop(self, mri_id):
        """np.ix_-shaped index triple of the sample inside the (240, 240, 155) BraTS volume."""
        return load_crop(self._path(mri_id, "_crop.npy"))

    def __iter__(self):
        return (self.get_one(mri_id) for mri_id in self.all_ids)

    def __getitem__(self, index):
        return self.get_one(self.all_ids[index])

    def __len__(self):
        return len(self.all_ids)


def save_crop(path_npz, crop):
    """Write a crop (np.ix_ triple or three index vectors) as three plain int64 arrays."""
    x, y, z = (np.asarray(a, dtype=np.int64).reshape(-1) for a in crop)
    np.savez(path_npz, x=x, y=y, z=z)


class _ArraysOnlyUnpickler(__import__("pickle").Unpickler):
    """Unpickler for the reference's `_crop.npy` (np.save of a tuple of three differently shaped
    index arrays = a pickled object array, scripts/preprocess_dataset.py:130).  It can rebuild
    numpy arrays / dtypes / scalars and nothing else: any other global in the stream raises."""

    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("numpy", "ndarray"), ("numpy", "dtype")}

    def find_class(self, module, name):
        if (module, name) not in self._ALLOWED:
            raise ValueError(f"crop file asks for {module}.{name}: not an array, refusing to load it")
        return super().find_class(module, name)


def load_crop(path_npy):
    """Crop of a sample as an np.ix_-shaped triple.  Reads `{id}_crop.npz` (three plain index
    arrays, `save_crop`) when it exists; otherwise the reference's `{id}_crop.npy`, without ever
    running numpy's general pickle loader on it."""
    path_npz = path_npy[:-4] + ".npz"
    if os.path.exists(path_npz):
        with np.load(path_npz, allow_pickle=False) as z:
            return np.ix_(z["x"], z["y"], z["z"])
    try:
        parts = np.load(path_npy, allow_pickle=False)              # a plain (non-object) array
    except ValueError:
        with open(path_npy, "rb") as fh:
            major, _minor = np.lib.format.read_magic(fh)
            (np.lib.format.read_array_header_1_0 if major == 1 else np.lib.format.read_array_header_2_0)(fh)
            parts = _ArraysOnlyUnpickler(fh).load()
    parts = [np.asarray(a) for a in parts]
    if len(parts) != 3 or any(a.dtype.kind not in "iub" for a in parts):
        raise ValueError(f"{path_npy}: expected three index arrays")
    if all(a.dtype.kind == "b" for a in parts):
        return np.ix_(*[a.reshape(-1) for a in parts])
    return np.ix_(*[a.reshape(-1).astype(np.int64) for a in parts])


def minibatch_graphs(samples):
    """Collate [(id, graph, feats, labels), ...] into one block-diagonal batch
    (reference data_loader.py:165-169)."""
    mri_ids, graphs, features, labels = map(list, zip(*samples))
    # same values as torch.FloatTensor(np.concatenate(features)): each float64 block is rounded to
    # fp32 while it is copied into place (one pass instead of concatenate-then-convert)
    rows = [np.asarray(f) for f in features]
    feats = np.empty((sum(len(f) for f in rows),) + rows[0].shape[1:], dtype=np.float32)
    at = 0
    for f in rows:
        feats[at:at + len(f)] = f
        at += len(f)
    return (mri_ids, gts.batch(graphs), torch.from_numpy(feats),
            torch.from_numpy(np.concatenate(labels).astype(np.int64, copy=False)))
