"""Static checks on the ISA of k_cnet1w (pytorch-glow_amd/csrc/cnet1w_sh.hip).  Its f.0 / f.4 accumulators live in VGPRs through
inline-asm MFMAs, and hipcc pads no hazard around an asm statement: scripts/lint_asm_mfma.py checks the two wait-state rules such
an MFMA has towards compiler-generated code on the generated assembly, and this test keeps the register budget honest (one wave
per SIMD, no scratch: a spill inside the LDS-DMA loop would also break its counted vmcnt waits).  CPU only: hipcc cross-compiles."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytorch-glow_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_cnet1w_asm_mfma_hazards_and_register_budget(tmp_path):
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-pragma-unroll-threshold=1000000",
             "--cuda-device-only", "-S"]
    out = os.path.join(str(tmp_path), "cnet1w.s")
    subprocess.run([HIPCC, *flags, os.path.join(CSRC, "cnet1w_sh.hip"), "-o", out], check=True, cwd=CSRC)
    asm = open(out).read()
    assert "v_mfma_f32_32x32x16_f16" in asm
    # every kernel of the file: no scratch, no spills, 512 registers (256 + 256) at most
    for m in re.finditer(r"\.private_segment_fixed_size:\s*(\d+)", asm):
        assert int(m.group(1)) == 0, "k_cnet1w must not use scratch memory"
    for m in re.finditer(r"\.vgpr_spill_count:\s*(\d+)", asm):
        assert int(m.group(1)) == 0
    assert "scratch_" not in asm
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "lint_asm_mfma.py"), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert re.search(r"(\d+) inline-asm MFMAs", r.stdout) and int(re.search(r"(\d+) inline-asm MFMAs", r.stdout).group(1)) > 100, r.stdout
