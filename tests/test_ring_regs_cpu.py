"""The register-ring kernels (K2r mfma_r128.hip, K8m ivf_lists_mfma.hip) load database rows with opaque
`global_load_dwordx4` statements whose destination registers are written long after the instruction.  The
compiler must neither copy nor reuse those registers before the counted wait of their ring slot, and must not
spill them; it knows nothing of either rule, so its output is checked: tools/check_ring_regs.py walks the gfx950
assembly of every instantiation.  (An earlier form of both kernels had three alternative wait statements per
step; their results met in a phi and the compiler copied ring registers in front of the wait: rare stale
operands, found as run-to-run differences of the batched IVF search at 10M rows.)"""
import importlib.util
import os
import re
import subprocess
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "semantic-search_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("src", ["mfma_r128.hip", "ivf_lists_mfma.hip"])
def test_ring_registers_untouched_while_in_flight(src, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-DNDEBUG", "-S",
                    "--cuda-device-only", "-o", str(out), os.path.join(CSRC, src)], check=True, capture_output=True)
    text = out.read_text()
    scratch = re.findall(r"; ScratchSize: (\d+)", text)
    assert len(scratch) == 18 and all(s == "0" for s in scratch), scratch  # 9 shapes x (fp16, bf16); a spilled ring register stores stale bits
    spec = importlib.util.spec_from_file_location("check_ring_regs", os.path.join(ROOT, "tools", "check_ring_regs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(str(out)) == 0
