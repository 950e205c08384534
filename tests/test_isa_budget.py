"""Register budgets of the frame kernels, read from the compiler's assembly (no GPU needed: hipcc cross-compiles).
The launch geometry in npr_api.cpp (rs_waves_per_cu: 28 / 24 / 16 wavefronts per CU) counts on 7 / 6 / 4 wavefronts per
SIMD, i.e. at most 72 / 80 / 128 VGPRs; one register more and a launch holds a sixth fewer reads than it was sized for
(DESIGN.md 5.1d).  The two-wavefront kernel shares the chip with the one-wavefront kernel's budget."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def rs_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "rs.s"
    src = os.path.join(ROOT, "nanopore_amd", "csrc")
    # the Makefile's flags (nanopore_amd/csrc/Makefile CXXFLAGS)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
                           "-I" + os.path.join(ROOT, "include"), "-I" + src, "-x", "hip", "--cuda-device-only", "-S",
                           os.path.join(src, "npr_kernel_rs.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernel_meta(asm, mangled_part):
    m = re.search(r"\.amdhsa_kernel (\S*%s\S*)\n(.*?)\.end_amdhsa_kernel" % re.escape(mangled_part), asm, re.S)
    assert m, mangled_part
    body = m.group(2)
    get = lambda key: int(re.search(r"\.amdhsa_%s (\d+)" % key, body).group(1))
    return get("next_free_vgpr"), get("private_segment_fixed_size"), get("group_segment_fixed_size")


@pytest.mark.parametrize("name,vgpr_max", [("7k_dp_rsILi1E", 72), ("7k_dp_rsILi2E", 80), ("7k_dp_rsILi4E", 128),
                                           ("12k_dp_pair_rsILi1E", 72), ("12k_dp_pair_rsILi2E", 80), ("12k_dp_pair_rsILi4E", 128)])
def test_frame_kernels_keep_their_register_budget(rs_asm, name, vgpr_max):
    vgpr, scratch, lds = _kernel_meta(rs_asm, name)
    assert vgpr <= vgpr_max, (name, vgpr)
    assert scratch <= 16, (name, scratch)  # k_dp_rs<2> spills two registers to stay at six wavefronts per SIMD; nothing else spills
    assert lds <= 2048, (name, lds)        # tables + model: static LDS, far below what would cap the wavefronts per CU
