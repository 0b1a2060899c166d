"""Runs the repository's bench.py on the CPU with the gloo tests' oracle stand-ins in place of the HIP kernels.

Test scaffolding (used by tests/test_capi_and_host.py::test_plain_bench_command_starts_the_ranks_itself): bench.py re-launches
``sys.argv[0]`` -- this file -- as N ranks, every rank patches the kernel entry points (tests/test_dist_gloo.py) and then executes
bench.py as ``__main__``.  Only the host logic is exercised: launcher, rank environment, partition plan, collectives, JSON line."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    if "WORLD_SIZE" in os.environ:       # a rank: the stand-ins first (the parent only spawns)
        from test_dist_gloo import _patch_ops_with_oracle
        _patch_ops_with_oracle()
    os.environ["EGNN_BENCH_ENTRY"] = os.path.abspath(__file__)    # what bench.py's own launcher starts as a rank
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
