"""CPU oracle for the GNN-distillation hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a pure-PyTorch (CPU, fp32 / int64) restatement of the algorithm the
reference executes on its hot path:

  * ``sparse.py``     torch-sparse 0.6.8/0.6.9 ``SparseTensor`` semantics the reference relies on
                      (``/root/reference/arxiv_pyg/gnn.py:236-249``, ``mag_pyg/gnn.py:149-162``)
  * ``nn.py``         PyG 1.6.3/1.7.0 ``GCNConv`` / ``SAGEConv`` / ``GATConv`` (``arxiv_pyg/gnn.py:28-35,61-67``,
                      ``ppi_pyg/gnn.py:86-132``), the slice of ``MessagePassing`` the reference's own ``RGCNConv``
                      subclass needs, and that ``RGCNConv`` (``mag_pyg/gnn.py:25-68``)
  * ``utils.py``      ``torch_geometric.utils.softmax`` / ``subgraph`` (``arxiv_pyg/criterion.py:5``)
  * ``criterion.py``  the six distillation losses (``arxiv_pyg/criterion.py:8-149``) and the PPI
                      multi-label ``kd_criterion`` (``ppi_pyg/criterion.py:8-18``)
  * ``models.py``     ``GCN`` / ``SAGE`` / projection heads and the train / eval step
                      (``arxiv_pyg/gnn.py:23-218``); the PPI teacher models ``GAT`` / ``TeacherNet`` and the PPI
                      epoch / micro-F1 loops (``ppi_pyg/gnn.py:23-117,185-288``); ``RGCN`` forward / inference
                      (``mag_pyg/gnn.py:71-168``); the arxiv GAT teacher ``ArxivGAT`` / ``nn.DGLGATConv`` and the artefact-producing
                      ``teacher_evaluate`` with label reuse (``arxiv_dgl/models.py:95-313``, ``arxiv_dgl/gat.py:104-107,151-183``)

Third-party arithmetic (PyG, torch-sparse, torch-scatter) is NOT vendored by the reference and is not
installable here; it is restated from the published algorithm of the versions contemporary with the
reference's pin (PyTorch 1.7.1, ``/root/reference/README.md:37-59``).

PARITY PINNING STATUS
  * the six criteria and the ``GCN``/``SAGE``/``train()``/``test()`` bodies are PINNED: golden vectors
    in ``tests/golden/`` were produced by importing the reference's own ``arxiv_pyg/criterion.py``,
    ``ppi_pyg/criterion.py`` and ``arxiv_pyg/gnn.py`` in the build container
    (``tests/golden/make_golden.py``), with only the un-installable third-party imports shimmed; likewise
    the PPI ``GAT`` / ``TeacherNet`` / ``train()`` / ``test()`` bodies (``ppi_pyg/gnn.py``) and the MAG
    ``RGCNConv`` / ``RGCN`` forward and inference (``mag_pyg/gnn.py``), and the arxiv ``GAT`` / ``GATConv`` teacher forward
    (``arxiv_dgl/models.py``, DGL's message-passing built-ins shimmed; the golden caught that ``er`` is formed from the
    UNSCALED destination features under ``use_symmetric_norm``);
  * the PyG / torch-sparse operator semantics underneath (``GCNConv``, ``SAGEConv``, ``GATConv``,
    ``MessagePassing.propagate``, ``SparseTensor``, ``utils.softmax``) are "parity unpinned": the reference holds no tests, fixtures or golden vectors
    for them and the packages cannot be imported; they are pinned only by
    hand-computed known-answer tests in ``tests/test_oracle_known_answers.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and there only as the checker / the timed CPU baseline.  The product package never does.
"""
