"""Static checks on the gfx950 assembly of the wide GEMM kernels (no GPU needed: hipcc cross-compiles).  They pin three properties that
were each found broken once by reading the ISA and that no numerical test can see reliably:
  * no vector instruction touches the destination of an LDS read that may still be in flight (inline-asm fragment reads are invisible
    to the compiler's s_waitcnt insertion; tools/isa_hazard_audit.py models the in-order lgkmcnt counter and walks every loop twice);
  * no LDS-DMA load sits in a waterfall loop (a scalar offset that the compiler kept in a VGPR);
  * the kernels of the default train step have no private segment (a kernel with scratch between kernels without it pays a queue-side
    scratch set-up per dispatch: 0.5 ms each, measured with the persistent variant)."""
import importlib.util
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "scl-deepfake-audio-detection_amd", "csrc", "gemm_w8.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    out = str(tmp_path_factory.mktemp("isa") / "gemm_w8.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S", "-o", out, SRC],
                   check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(asm_text):
    for km in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)\.Lfunc_end", asm_text, re.S | re.M):
        lines = [l.split(";")[0].strip() for l in km.group(2).splitlines()]
        yield km.group(1), [l for l in lines if l]


def test_no_use_of_a_register_with_an_lds_read_in_flight(asm):
    spec = importlib.util.spec_from_file_location("isa_hazard_audit", os.path.join(ROOT, "tools", "isa_hazard_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.audit(asm) == 0


def test_no_waterfall_loop_around_a_buffer_load(asm):
    found = []
    for name, lines in _kernels(asm):
        for i, l in enumerate(lines):
            if l.startswith("s_cbranch_execnz"):
                tgt = l.split()[-1] + ":"
                for k in range(i - 1, max(0, i - 16), -1):
                    if lines[k] == tgt:
                        if any("v_readfirstlane" in x for x in lines[k:i]) and any(x.startswith("buffer_load") for x in lines[k:i]):
                            found.append(name)
                        break
    assert not found, sorted(set(found))


def test_default_path_kernels_have_no_private_segment(asm):
    meta = re.findall(r"- \.agpr_count.*?\.wavefront_size: 64", asm, re.S)
    assert meta
    seen = 0
    for blk in meta:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if "scl_gemm_w8p_kernel" in name:      # the persistent variant is opt-in (and documented as spilling)
            continue
        if "scl_gemm_w8" not in name:
            continue
        seen += 1
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)) == 0, name
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)) == 0, name
    assert seen == 19      # ping-pong + single barrier, 4 layouts, 208- and 256-row tiles; the grouped weight-gradient kernel; round 6: the two 112-row single-barrier kernels (NN, NT)
