"""Build libmagpo_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(os.path.dirname(HERE), "build")
OUT = os.path.join(HERE, "libmagpo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value"] + os.environ.get("MAGPO_EXTRA_FLAGS", "").split()


def _newer(src, dst):
    return not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force: bool = False, verbose: bool = False, out: str = OUT, extra_flags=(), build_dir: str = BUILD, only: str = "") -> str:
    """``out`` / ``extra_flags`` / ``build_dir``: experiment builds (debug timers, A/B macros) next to the product library, e.g.
    ``python -m magpo_amd.build --out exp_libs/prof.so --flags=-DMAGPO_ACT_PROF`` (scripts pick them up through MAGPO_LIB).
    ``only`` (experiment builds): comma-separated source-name prefixes that are compiled with the extra flags; every other object is
    taken from the product build directory as it is (``--only act_fused``: 4 translation units instead of 19)."""
    os.makedirs(build_dir, exist_ok=True)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp"))
    objs, jobs = [], []
    pref = tuple(p for p in only.split(",") if p)
    for s in srcs:
        name = os.path.basename(s)[:-4]
        if pref and not name.startswith(pref):
            objs.append(os.path.join(BUILD, name + ".o"))   # the product build's object
            continue
        o = os.path.join(build_dir, name + ".o")
        objs.append(o)
        if force or pref or _newer(s, o) or any(_newer(h, o) for h in hdrs):
            jobs.append([HIPCC, *FLAGS, *extra_flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(out):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--out", default=OUT)
    ap.add_argument("--flags", default="")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    bd = BUILD if a.out == OUT else os.path.join(BUILD, "exp_" + os.path.basename(a.out).replace(".so", ""))
    print(build(force=a.force, verbose=True, out=a.out, extra_flags=a.flags.split(), build_dir=bd, only=a.only))
