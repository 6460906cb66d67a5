"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the
header declares, and the product path refuses to run without a GPU (no silent fallback)."""
import ctypes
import os
import re

import pytest
import torch

from adafocus_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()


def test_header_symbols_exported():
    _ensure_built()
    header = open(os.path.join(ROOT, "include", "adafocus.h")).read()
    declared = set(re.findall(r"\b(adaf_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.adaf_version.restype = ctypes.c_int
    assert lib.adaf_version() == 303      # 303: adaf_set_global_option / adaf_get_global_option replace adaf_set_option / adaf_get_option, adaf_mobilenetv2_set_dtype removed (round 6)


def test_loader_declares_prototypes():
    _ensure_built()
    lib = _lib.load_library()
    assert lib.adaf_resnet50_workspace_bytes.restype is ctypes.c_size_t
    # pure host helpers are callable without a device
    assert lib.adaf_gru_cls_workspace_bytes(2, 8, 1024) == (2 * 8 * 3072 + 2 * 3072 + 2 * 8 * 1024) * 4
    assert lib.adaf_resnet50_workspace_bytes(None, 4, 96) == 5 * 4 * 48 * 48 * 64 * 4
    assert lib.adaf_resnet50_workspace_bytes(None, 1, 98) == 5 * 25 * 25 * 256 * 4   # odd stem size: stage-1 map is larger


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_fails_loudly():
    _ensure_built()
    from adafocus_amd import hip_ops, utils
    with pytest.raises(_lib.AdafError):
        utils.get_patch(torch.zeros(1, 3, 224, 224), torch.zeros(1, 2), 96)
    with pytest.raises(_lib.AdafError):
        hip_ops.maxpool3x3s2(torch.zeros(1, 8, 8, 64))
    h = ctypes.c_void_p()
    assert _lib.load_library().adaf_create(0, ctypes.byref(h)) == -3   # ADAF_E_ARCH: no gfx950 device


def test_no_kernel_converts_to_fp16_with_a_single_rounding(tmp_path):
    """fp16 STORAGE means: the fp32 result, rounded, then converted.  The compiler may fold `(_Float16)(a * b)` into v_fma_mixlo_f16 /
    v_fma_mixhi_f16 -- one rounding of the exact product -- and it does so per ELEMENT of an unrolled epilogue (round 4: 15 of 16
    accumulators of the expand kernel), so a frame's features depended on which MFMA row its pixels landed on and the "bit-identical"
    kernel forms only agreed while the compiler made the same choice in each.  adaf_f16_of (csrc/adaf_internal.h) closes that; this test
    disassembles every gfx950 code object of the built library and fails if the fold comes back anywhere."""
    import glob
    import shutil
    import subprocess
    _ensure_built()
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump in this image")
    shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    cos = sorted(glob.glob(str(tmp_path / "lib.so.*gfx950*")))
    assert cos, os.listdir(tmp_path)
    bad = {}
    for co in cos:
        dis = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
            elif "v_fma_mixlo_f16" in line or "v_fma_mixhi_f16" in line:
                bad[cur] = bad.get(cur, 0) + 1
    assert not bad, bad


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "adafocus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


# Kernels of the product library that may use scratch (private segment), with the most bytes each is allowed: a ratchet -- an entry may
# only shrink or disappear.  Everything else must compile to ZERO scratch (a spill in an MFMA loop is a 10-30 % loss that no test notices).
# Keys: the demangled name up to the argument list, anonymous namespace stripped.
SCRATCH_ALLOWED = {
    "ef_expand_kernel<float, 8, false>": 104, "ef_expand_kernel<float, 8, true>": 84, "ef_expand_kernel<float, 7, false>": 48,
    "ef_expand_kernel<float, 7, true>": 40, "ef_expand_kernel<_Float16, 8, true>": 76, "ef_expand_kernel<_Float16, 8, false>": 44,
    "ef_expand_kernel<_Float16, 7, true>": 28,
    "mb_block_w_kernel<32>": 8,
}


def kernel_scratch_table(lib_path, workdir):
    """{short kernel name: (private_segment_fixed_size, vgpr_count, vgpr_spill_count)} for every gfx950 kernel of the library, from
    the code objects' metadata notes (llvm-objdump --offloading + llvm-readelf --notes)."""
    import glob
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin/"
    shutil.copy(lib_path, os.path.join(workdir, "lib.so"))
    subprocess.run([llvm + "llvm-objdump", "--offloading", "lib.so"], cwd=workdir, check=True, capture_output=True)
    table = {}
    for co in sorted(glob.glob(os.path.join(workdir, "lib.so.*gfx950*"))):
        notes = subprocess.run([llvm + "llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
        for k in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", k).group(1)
            priv = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", k).group(1))
            vg = int(re.search(r"\.vgpr_count:\s+(\d+)", k).group(1))
            sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", k)
            table[name] = (priv, vg, int(sp.group(1)) if sp else 0)
    names = list(table)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    short = {}
    for n, d in zip(names, dem):
        d = d.replace("(anonymous namespace)::", "").replace("void ", "", 1)
        m = re.match(r"^(.*?>)\(", d) or re.match(r"^([^(]+)\(", d)
        key = m.group(1) if m else d
        if key.startswith("_ZN"):      # c++filt could not demangle (_Float16 template arguments): spell the ef_expand family by hand
            mm = re.match(r"_ZN12_GLOBAL__N_116ef_expand_kernelIDF16_Li(\d)ELb([01])EEE", key)
            if mm:
                key = "ef_expand_kernel<_Float16, %s, %s>" % (mm.group(1), "true" if mm.group(2) == "1" else "false")
        short[key] = table[n]
    return short


def test_no_kernel_uses_scratch_outside_the_allow_list(tmp_path):
    """Every kernel of the product library (all are reachable from the default plans since the experiment tiles moved to
    tools/exp/build_exp_tiles.sh) compiles WITHOUT scratch, except the allow-listed ones, which may not grow."""
    _ensure_built()
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("no llvm-readelf in this image")
    table = kernel_scratch_table(_lib.LIB_PATH, str(tmp_path))
    assert len(table) > 300
    over = {k: v for k, v in table.items() if v[0] > SCRATCH_ALLOWED.get(k, 0)}
    assert not over, over
    stale = {k: lim for k, lim in SCRATCH_ALLOWED.items() if k not in table}
    assert not stale, ("allow-list entries without a kernel (renamed or gone: delete them)", stale)
    # no 256 x 256 conv tiles in the product (they spill 900-1500 VGPRs and no automatic rule selects them)
    assert not [k for k in table if k.startswith("conv_gemm_glds_kernel<256, 256")]
