"""Lists the private-segment (scratch) size of every kernel in libssmi's objects.

A kernel that spills reloads registers through scratch with `s_waitcnt vmcnt(0)` in front of the use, which
drains its memory pipeline: the per-query IVF scan ran at 0.55 of HBM peak instead of 0.66 until its spill of the
query registers was found.  The hot kernels therefore must compile to zero scratch (tests/test_no_spills_cpu.py).

usage: audit_scratch.py [build dir]      prints `<bytes> <kernel>` for every kernel with scratch"""
import os, re, subprocess, sys, tempfile, shutil

LLVM = "/opt/rocm/lib/llvm/bin"

def kernels_with_scratch(build_dir):
    out = {}
    tmp = tempfile.mkdtemp()
    try:
        for o in sorted(f for f in os.listdir(build_dir) if f.endswith(".o")):
            dst = os.path.join(tmp, o)
            shutil.copy(os.path.join(build_dir, o), dst)
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], capture_output=True, cwd=tmp)
            for co in (f for f in os.listdir(tmp) if f.startswith(o) and "amdgcn" in f):
                notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, co)],
                                       capture_output=True, text=True).stdout
                name = None
                for line in notes.split("\n"):
                    m = re.match(r"\s+\.name:\s+(\S+)", line)
                    if m: name = m.group(1)
                    m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
                    if m: size = int(m.group(1))
                    m = re.match(r"\s+\.symbol:\s+(\S+)\.kd", line)
                    if m: out[m.group(1)] = size
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "semantic-search_amd", "csrc", "build")
    ks = kernels_with_scratch(d)
    print(len(ks), "kernels")
    for k, v in sorted(ks.items()):
        if v: print(v, k)
