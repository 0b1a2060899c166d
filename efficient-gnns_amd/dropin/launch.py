#!/usr/bin/env python3
"""Run one of the reference's scripts UNCHANGED on the MI355X operators:

    python /path/to/repo/efficient-gnns_amd/dropin/launch.py /path/to/efficient-gnns/arxiv_pyg/gnn.py --gnn gcn ...

``python gnn.py`` puts the script's own directory at sys.path[0], so the reference's ``criterion.py`` would shadow the
drop-in one whatever PYTHONPATH says.  This launcher builds the path explicitly: repo root, then the shim directory that
matches the script's flavour -- ``dropin/`` (class-index CE/KL losses: arxiv_pyg, mag_pyg) or ``dropin/ppi_pyg/`` +
``dropin/`` (multi-label BCE ``kd_criterion``: any script directory whose own criterion.py is BCE based) -- and only
then the script's directory (its logger.py etc. still resolve).  Use ``--keep-criterion`` to shim the operators
(GCNConv, SparseTensor, ...) but keep the script's own criterion.py, ``--plain-torch-modules`` to leave torch.nn.BatchNorm1d /
torch.nn.Linear of the script's own model code on PyTorch's kernels (default: the package's, scoped to the script's run, see accel.py),
``--no-deferred-activations`` for accel with one launch per torch call (no lazy.py objects).
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def shim_path(script: str, keep_criterion: bool = False) -> list:
    """sys.path entries (in order) to put in front for ``script``."""
    sdir = os.path.dirname(os.path.abspath(script))
    own = os.path.join(sdir, "criterion.py")
    multilabel = os.path.exists(own) and "binary_cross_entropy_with_logits" in open(own).read()
    front = [ROOT]
    if keep_criterion:
        front += [sdir, HERE]            # the script's criterion.py wins; operators still come from dropin/
    else:
        front += ([os.path.join(HERE, "ppi_pyg")] if multilabel else []) + [HERE, sdir]
    return front


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    keep = "--keep-criterion" in argv
    if keep:
        argv.remove("--keep-criterion")
    plain = "--plain-torch-modules" in argv
    if plain:
        argv.remove("--plain-torch-modules")
    eager = "--no-deferred-activations" in argv        # accel with one launch per torch call (round 5's form): accel.LAZY = False
    if eager:
        argv.remove("--no-deferred-activations")
    if not argv:
        raise SystemExit(__doc__)
    script = argv[0]
    sys.path[:0] = shim_path(script, keep)
    sys.argv = argv
    if plain:
        runpy.run_path(script, run_name="__main__")
        return
    # torch.nn.BatchNorm1d / torch.nn.Linear of the script's own model code on the package's kernels (dropin/accel.py), SCOPED to the run of
    # the script: torch's own methods are back when it returns or raises (an embedding process is left as it was found)
    import importlib.util
    spec = importlib.util.spec_from_file_location("egnn_dropin_accel", os.path.join(HERE, "accel.py"))
    accel = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(accel)
    accel.LAZY = not eager
    with accel.scope():
        runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
