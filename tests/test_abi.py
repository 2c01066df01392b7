"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/gts_hip.h declares (no compute calls here)."""
import ctypes
import os
import subprocess

from gts import _lib, build


def test_library_builds_and_exports_declared_symbols(hip_lib):
    declared = _lib.declared_symbols()
    assert len(declared) >= 10
    assert set(declared) == set(_lib.SIGNATURES), "header and ctypes table disagree"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in gts_hip.h but not exported"
    assert hip_lib.gts_abi_version() == _lib.ABI_VERSION


def test_library_is_gfx950_code(hip_lib):
    out = subprocess.run(["strings", "-n", "6", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_argument_errors_do_not_need_a_gpu(hip_lib):
    """Argument validation happens before any launch: checkable on the CPU box."""
    assert hip_lib.gts_spmm_max_fwd_f32(None, None, None, None, None, 0, 0, 4, 4, None) == -1
    one = ctypes.c_void_p(16)
    assert hip_lib.gts_spmm_max_fwd_f32(one, one, one, one, one, 3, 0, 4, 4, None) == -3
    assert hip_lib.gts_spmm_sum_f32(one, one, one, one, None, None, None, 0, -1, 4, None) == -2
    assert hip_lib.gts_project_rows_i16(one, one, one, one, 10, 10, 5, None) == -3
    assert hip_lib.gts_gat_fwd_f32(one, one, one, one, one, 0.2, None, None, 0, one, one, 4, 0, 4, None) == -2
    assert hip_lib.gts_gat_fwd_f32(one, one, one, one, one, 0.2, None, None, 7, one, one, 4, 4, 4, None) == -3
    assert b"NULL" in hip_lib.gts_error_string(-1)
    # tuning knobs take the kernel forms the library carries and nothing else (the rejected ones are built from tools/diag/)
    assert hip_lib.gts_set_option(2, 7) == -3 and hip_lib.gts_set_option(1, 9) == -3 and hip_lib.gts_set_option(9, 1) == -3
    assert hip_lib.gts_set_option(2, 6) == 0 and hip_lib.gts_set_option(2, -1) == 0
    # zero-sized problems are accepted without touching memory or the device
    assert hip_lib.gts_spmm_max_fwd_f32(one, one, one, one, None, 0, 1, 0, 4, None) == 0


def test_cluster_schedule_rejects_a_malformed_csr(hip_lib):
    """gts_cluster_schedule is a HOST entry point of the C ABI: ids outside [0, n), extents that go down or that do not
    match the transpose must come back as GTS_ERR_SHAPE, not index the builder's tables."""
    import numpy as np

    def run(indptr, indices, t_indptr, t_indices):
        arrs = [np.asarray(a, dtype=np.int32) for a in (indptr, indices, t_indptr, t_indices)]
        out = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
        return hip_lib.gts_cluster_schedule(*[a.ctypes.data_as(ctypes.c_void_p) for a in arrs], None, 3, 8, 16, 64, None, 0,
                                            ctypes.byref(out[0]), ctypes.byref(out[1]), ctypes.byref(out[2]))
    good = ([0, 1, 2, 3], [1, 2, 0], [0, 1, 2, 3], [2, 0, 1])
    assert run(*good) == 0
    assert run([0, 1, 2, 3], [1, 7, 0], good[2], good[3]) == -2            # neighbour id past the last row
    assert run([0, 1, 2, 3], [1, -1, 0], good[2], good[3]) == -2
    assert run(good[0], good[1], [0, 1, 2, 3], [2, 0, 3]) == -2            # ... in the transpose
    assert run([0, 2, 1, 3], good[1], good[2], good[3]) == -2              # extents go down
    assert run([0, 1, 2, 3], good[1], [0, 1, 2, 2], good[3]) == -2         # transpose holds another number of edges
    assert run([1, 1, 2, 3], good[1], good[2], good[3]) == -2              # extents do not start at 0


def test_product_refuses_cpu_tensors(hip_lib):
    import numpy as np
    import pytest
    import torch

    import gts
    from gts import ops

    g = gts.Graph(np.array([0]), np.array([1]), 2)
    with pytest.raises(gts.GtsError, match="MI355X only"):
        ops.spmm_max_fwd(g, torch.zeros(2, 4))


def test_header_cites_reference_lines():
    text = open(_lib.HEADER_PATH).read()
    for cite in ("model/networks.py:25", "data_processing/graph_io.py:21-24",
                 "scripts/generate_gnn_predictions.py:55-61"):
        assert cite in text
    assert os.path.exists(build.LIB_PATH)
