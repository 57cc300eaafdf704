"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol declared in include/scl_hip.h; the ctypes table and the header agree; argument validation
returns error codes instead of crashing; the product path refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    hdr = open(os.path.join(ROOT, "include", "scl_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(scl_[a-z0-9_]+)\s*\(", hdr))


def test_library_loads_and_exports_every_declared_symbol():
    from scl_amd import lib
    L = lib.load()
    names = header_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), "libscl_hip.so does not export %s" % n
    assert names == set(lib.all_symbol_names()), (names ^ set(lib.all_symbol_names()))
    assert L.scl_version() >= 100


def test_struct_layout_matches_header():
    from scl_amd import lib
    assert ctypes.sizeof(lib.SclOperand) == 56
    assert ctypes.sizeof(lib.SclGemmDesc) == 2 * 56 + 4 * 8 + 5 * 8 + 8 * 4 + 4 * 4 + 8 + 8  # pointers, strides, ints, floats/seed/pad, colsum_part
    assert lib.SclGemmDesc.colsum_part.offset == ctypes.sizeof(lib.SclGemmDesc) - 8
    assert lib.SclGemmDesc.C.offset == 112 and lib.SclGemmDesc.flags.offset == 112 + 32 + 40 + 32
    # SclBtseBio: 147 + 6 pointers, two int64, 147 + 11 int32
    assert ctypes.sizeof(lib.SclBtseBio) == 153 * 8 + 16 + 158 * 4 and lib.SclBtseBio.go.offset == 153 * 8 + 16
    assert lib.SclBtseBio.n_layers.offset == 153 * 8 + 16 + 147 * 4 and lib.SclBtseBio.bio.offset == 147 * 8


def test_argument_validation_returns_error_codes_without_touching_the_gpu():
    from scl_amd import lib
    L = lib.load()
    d = lib.SclGemmDesc()
    assert L.scl_gemm_bf16(ctypes.byref(d), None) == -1
    assert b"M,N,K" in L.scl_last_error()
    assert L.scl_layernorm_fwd(None, 1, None, None, None, None, None, None, 4, 8, 8, 8, 1e-5, 0, None) == -1
    assert L.scl_prof_enable(99, 1) == -1
    assert L.scl_fir_nblocks(64000) == 16 and L.scl_supcon_nchunks(25472) == 32 and L.scl_supcon_nchunks(1 << 20) == 512


def test_product_path_refuses_cpu():
    from scl_amd.model_linear import Model
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Model({"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}, "cpu")


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "scl-deepfake-audio-detection_amd")
    offenders = []
    for base in (pkg, os.path.join(ROOT, "datautils"), os.path.join(ROOT, "model")):
        for fn in os.listdir(base):
            if fn.endswith(".py"):
                src = open(os.path.join(base, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    offenders.append(fn)
    src = open(os.path.join(ROOT, "main.py")).read()
    assert not offenders and not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M)


def test_shipped_library_holds_only_default_path_kernels():
    """The GEMM experiments that lost their A/Bs (two workgroups per CU, persistent blocks, 256x128 ring, 256x256 ping-pong, start stagger)
    are compiled only with -DSCL_EXPERIMENTS: the shipped library must not carry their code objects (unused instantiations cost the
    default path through the instruction cache) nor their environment switches."""
    from scl_amd import lib
    L = lib.load()
    assert L.scl_build_flags() == 0, "libscl_hip.so was built with -DSCL_EXPERIMENTS; rebuild without SCL_BUILD_DEFINES"
    data = open(lib.LIB_PATH, "rb").read()
    for name in (b"scl_gemm_w8p_kernel", b"scl_gemm_x2_kernel", b"scl_gemm_p8_kernel", b"scl_gemm_big_kernel"):
        assert data.count(name) == 0, name
    for env in (b"SCL_GEMM_PERSIST", b"SCL_W8_STAGGER", b"SCL_GEMM_X2"):
        assert env not in data, env
    for name in (b"scl_gemm_w8s_kernel", b"scl_gemm_dma_kernel", b"scl_gemm_f32_kernel", b"rs_conv_kernel", b"attn_fwd_kernel"):
        assert data.count(name) > 0, name
