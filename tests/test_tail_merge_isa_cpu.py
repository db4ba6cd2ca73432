"""The merge tail (csrc/tail_merge.h) hands a block's k-list to another workgroup without a release/acquire fence
pair: it relies on write-through (sc1) stores drained by `s_waitcnt vmcnt(0)` before the ticket atomic and on
L1-bypassing (sc1) loads behind it.  Those are properties of the generated code, so the generated code is checked
(tools/check_tail_merge.py) for every kernel that carries a tail: K1 (fp32 / fp16 rows) and the per-query IVF scans."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "semantic-search_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("src,min_kernels", [("scan_f32.hip", 20), ("scan_f16.hip", 20), ("ivf.hip", 60)])
def test_published_lists_are_write_through_and_drained_before_the_ticket(src, min_kernels, tmp_path, capsys):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-DNDEBUG", "-S",
                    "--cuda-device-only", "-o", str(out), os.path.join(CSRC, src)], check=True, capture_output=True)
    spec = importlib.util.spec_from_file_location("check_tail_merge", os.path.join(ROOT, "tools", "check_tail_merge.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(str(out)) == 0
    report = capsys.readouterr().out
    n = int(report.strip().split("\n")[-1].split()[0])
    assert n >= min_kernels, report   # the check saw the kernels it is meant for


def test_the_checker_reports_a_broken_hand_off(tmp_path):
    """A listing with a plain published store, no drain and a plain merge load must be reported (the checker checks)."""
    bad = tmp_path / "bad.s"
    bad.write_text("_ZN4ssmi4fakeEv:\n\tglobal_store_dwordx2 v[0:1], v[2:3], off\n\tglobal_atomic_add v1, v1, v4, s[0:1] sc0\n"
                   "\tglobal_load_dwordx2 v[0:1], v[2:3], off\n.Lfunc_end0:\n")
    spec = importlib.util.spec_from_file_location("check_tail_merge", os.path.join(ROOT, "tools", "check_tail_merge.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(str(bad)) == 1
