"""CPU oracle for the GNN node-classification hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gnn-tumor-seg_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / reported CPU baseline.

PARITY STATUS
-------------
The reference (rsinghlab/GNN-Tumor-Seg) holds no tests, golden vectors or
expected outputs for this path, and its arithmetic lives in DGL, which is not
installed and not installable offline (``import dgl`` -> ModuleNotFoundError;
an ordinary missing dependency, DGL version unpinned: README.md:20 "DGL>=0.4").

* Integer / copy functions that ARE importable from the reference
  (``data_processing/graph_io.py:21-24 project_nodes_to_img``,
  ``utils/hyperparam_helpers.py``, ``utils/training_helpers.py``) pin this
  oracle through the fixtures under ``tests/golden/`` produced by
  ``tests/golden/make_reference_fixtures.py``:  **parity pinned**.
* The DGL layer arithmetic (SAGEConv / GATConv / edge_softmax / batch /
  from_networkx) is restated from DGL's published algorithm (documented in
  each function):  **parity unpinned** for those functions.
"""
