"""Writes a tiny synthetic preprocessed dataset in the on-disk layout the reference's
preprocessing script produces (per sample folder: *_nxgraph.json, *_supervoxels.nii.gz,
*_label.nii.gz, *_input.nii.gz, *_crop.npy)."""
import os

import networkx as nx
import numpy as np

from data_processing import graph_io, nifti_io


def write_sample(root, mri_id, seed, shape=(24, 20, 16), cube=4, in_feats=20):
    rng = np.random.default_rng(seed)
    gx, gy, gz = (s // cube for s in shape)
    n = gx * gy * gz
    ids = np.arange(n, dtype=np.int16).reshape(gx, gy, gz)
    svs = np.repeat(np.repeat(np.repeat(ids, cube, 0), cube, 1), cube, 2)
    svs[:2] = -1                                              # some background voxels
    node_labels = rng.choice(4, size=n, p=[0.6, 0.2, 0.1, 0.1])
    voxel_labels = np.append(node_labels, 0)[svs].astype(np.int16)
    G = nx.Graph()
    for i in range(n):
        G.add_node(i, features=[float(v) for v in rng.standard_normal(in_feats)], label=int(node_labels[i]))
    idx = np.arange(n).reshape(gx, gy, gz)
    for axis in range(3):
        a = np.take(idx, np.arange(idx.shape[axis] - 1), axis=axis).ravel()
        b = np.take(idx, np.arange(1, idx.shape[axis]), axis=axis).ravel()
        G.add_edges_from(zip(a.tolist(), b.tolist()), weight=1.0)
    for i in range(n):
        G.add_edge(i, i, weight=1.0)                          # touching-adjacency graphs carry self loops
    folder = os.path.join(root, mri_id)
    os.makedirs(folder, exist_ok=True)
    graph_io.save_networkx_graph(G, os.path.join(folder, f"{mri_id}_nxgraph.json"))
    nifti_io.save_as_nifti(svs, os.path.join(folder, f"{mri_id}_supervoxels.nii.gz"))
    nifti_io.save_as_nifti(voxel_labels, os.path.join(folder, f"{mri_id}_label.nii.gz"))
    nifti_io.save_as_nifti(rng.standard_normal(shape + (4,)).astype(np.float32),
                           os.path.join(folder, f"{mri_id}_input.nii.gz"))
    mask = np.zeros((240, 240, 155), dtype=bool)
    mask[10:10 + shape[0], 20:20 + shape[1], 30:30 + shape[2]] = True
    crop = np.ix_(mask.any(axis=(1, 2)), mask.any(axis=(0, 2)), mask.any(axis=(0, 1)))
    np.save(os.path.join(folder, f"{mri_id}_crop.npy"), np.array(crop, dtype=object), allow_pickle=True)
    return svs, node_labels


def write_dataset(root, n_samples, prefix="BraTS_"):
    return {f"{prefix}{i:03d}": write_sample(root, f"{prefix}{i:03d}", seed=i) for i in range(n_samples)}
