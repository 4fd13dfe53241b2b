"""Static discipline of the hand-counted rings in tower_seq.hip, checked on the compiler's .s (tools/check_asm_ring.py): hipcc neither
counts the memory operations of an `asm` statement nor keeps out of the registers they are still loading
(cdna_hip_programming.md 5.7), so every build is audited - no GPU needed, hipcc cross-compiles gfx950 here."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def tower_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "tower_seq.s"
    src = os.path.join(ROOT, "dl-dkd_amd", "csrc", "tower_seq.hip")
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S",
                    "--cuda-device-only", "-o", str(out), src], check=True, capture_output=True)
    return str(out)


def _run(path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_ring
    return check_asm_ring.main(path)


def test_tower_rings_are_disciplined(tower_asm, capsys):
    assert _run(tower_asm) == 0
    out = capsys.readouterr().out
    assert out.count("0 violations") == 6           # gallery (fp32 h0; bf16 h0 = persistent), rows, query, two stamped diagnostic builds
    assert out.count("1440 ring reads") == 5 and out.count("1152 ring reads") == 1
    assert out.count("352 LDS-DMAs in the stream") == 1      # the persistent kernel: 42 chunks + the next item's chunks 0 and 1   # every weight fragment goes through the ring once


def test_checker_catches_a_register_touched_before_its_wait(tower_asm, tmp_path):
    """The checker is not vacuous: touch the destination of the first asm ring read right after it is issued."""
    lines = open(tower_asm).read().splitlines()
    inside = False
    for i, ln in enumerate(lines):
        if "TW_STREAM_BEGIN" in ln:
            inside = True
        m = re.match(r"\s*ds_read_b128 v\[(\d+):\d+\]", ln)
        if inside and m and lines[i - 1].strip().startswith(";;#ASMSTART"):
            lines.insert(i + 2, f"\tv_mov_b32_e32 v{m.group(1)}, 0")          # after ;;#ASMEND
            break
    bad = tmp_path / "bad.s"
    bad.write_text("\n".join(lines))
    assert _run(str(bad)) == 1
    lines2 = open(tower_asm).read().splitlines()
    for i, ln in enumerate(lines2):
        if "TW_STREAM_BEGIN" in ln:
            lines2.insert(i + 5, "\tglobal_load_dword v1, v2, s[0:1]")        # a vector-memory load inside the stream
            break
    bad2 = tmp_path / "bad2.s"
    bad2.write_text("\n".join(lines2))
    assert _run(str(bad2)) == 1
