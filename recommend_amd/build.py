"""Build librecommend_amd.so (gfx950) in-tree with hipcc.

The library is the C-ABI declared in include/recommend_amd.h; it is built here
on a CPU-only host (hipcc cross-compiles) and travels to the GPU box as a file.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "librecommend_amd.so")
SOURCES = ["host.cpp", "graph.hip", "spmm.hip", "dense.hip", "dense_wide.hip", "glue.hip", "optim.hip", "bpr.hip", "eval.hip", "eval_filter.hip", "knn.hip", "proj.hip"]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-fno-fast-math", "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, "..", "include", "recommend_amd.h"))
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src + ".o")
        cmd = [hipcc, "-x", "hip"] + [f for f in FLAGS if f != "-shared"] + \
              ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
