"""Builds libegnn_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the repo snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libegnn_hip.so")
SOURCES = ["api.hip", "spmm.hip", "spmm_blk.hip", "graph.hip", "losses.hip", "gemm.hip", "gemm_skinny.hip", "nce.hip", "pairwise.hip", "edge_softmax.hip",
           "fused_bn.hip", "graph_build.hip", "probe.hip", "feature_losses.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"] + os.environ.get("EGNN_EXTRA_FLAGS", "").split()


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_stamp() -> str:
    """Identity of the kernel library as a build-independent value: first 16 hex digits of the SHA-256 over the compile flags and the
    bytes of every source the library is built from (csrc/*.hip, csrc/*.h, include/egnn_hip.h).  The binary itself carries its build
    time (egnn_version_string), so two builds of the same sources differ byte for byte; measurements that must be tied to "these
    kernels" (profiles/spmm_traffic*.json) are stamped with this value instead."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(os.path.dirname(HERE), "include", "egnn_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "egnn_hip.h"))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    def compile_one(src: str) -> str:
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(obj, [path] + headers):
            cmd = [hipcc, *FLAGS, "-c", path, "-o", obj]
            if verbose:
                print("[egnn build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print("[egnn build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
