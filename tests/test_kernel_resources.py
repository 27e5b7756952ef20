"""No kernel of the SHIPPED library may use scratch memory or spill vector registers.

Why a test: hipcc ignores `#pragma unroll` silently once a loop's unrolled size passes its budget (-pragma-unroll-threshold, 16384 by default).  The fully unrolled walks of
csrc/conv_planes.hip index their register rings with the loop counter; past the budget those rings become dynamically indexed arrays in scratch memory and the launch runs 20x
slower with correct results (round 6: DispNet conv3's input gradient, 727 us instead of 37.5; profiles/r06_experiments.txt #10).  Nothing but the code object's metadata
shows it, so this reads the metadata of every kernel in libmadnet_hip.so: private_segment_fixed_size (scratch bytes per lane) and vgpr_spill_count must be 0."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd", "madnet_hip", "libmadnet_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_metadata(lib, tmp):
    """[(kernel symbol, scratch bytes, VGPR spills, VGPRs)] of every gfx950 kernel in the library's .hip_fatbin section (one bundle per translation unit)"""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(MAGIC, blob)]
    out = []
    for k, i in enumerate(starts):
        part, co = os.path.join(tmp, "b%d.bin" % k), os.path.join(tmp, "c%d.co" % k)
        open(part, "wb").write(blob[i:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co],
                       check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
        for m in re.finditer(r"\.private_segment_fixed_size:\s+(\d+).*?\.symbol:\s+(\S+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", notes, re.S):
            out.append((m.group(2), int(m.group(1)), int(m.group(4)), int(m.group(3))))
    return out


def test_no_kernel_of_the_shipped_library_uses_scratch_or_spills(tmp_path):
    if not os.path.exists(LIB):
        pytest.skip("libmadnet_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"):
        if not os.path.exists(os.path.join(LLVM, tool)):
            pytest.skip("%s is missing from %s" % (tool, LLVM))
    ks = kernel_metadata(LIB, str(tmp_path))
    assert len(ks) >= 600, len(ks)                      # the library holds ~640 kernel instances: fewer means the metadata was not read
    names = [k[0] for k in ks]
    assert any("conv_planes_s2bwd_kernel" in n for n in names) and any("wgrad_stream" in n for n in names)
    bad = [k for k in ks if k[1] or k[2]]
    assert not bad, "kernels with scratch / VGPR spills (symbol, scratch bytes per lane, spills, VGPRs): %s" % bad[:8]
    assert max(k[3] for k in ks) <= 512
