"""Build libssmi.so (HIP, gfx950 only) in-tree: python -m semantic_search_amd.build

Each .hip file is compiled to an object in parallel, then linked into
csrc/libssmi.so.  hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libssmi.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-DNDEBUG"]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True, jobs: int = 8) -> Path:
    sources = sorted(CSRC.glob("*.hip"))
    headers = sorted(CSRC.glob("*.h")) + [CSRC.parent.parent / "include" / "ssmi.h"]
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)

    def compile_one(src: Path):
        obj = objdir / (src.stem + ".o")
        if force or _stale(obj, [src] + headers):
            cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(compile_one, sources))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs),
               "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
