"""Static resource table of every gfx950 kernel in csrc/ (no GPU needed): hipcc --cuda-device-only -S per source, then the AMDGPU metadata of each kernel --
VGPRs, AGPRs, scratch bytes, VGPR / SGPR spills, static LDS.  A kernel with scratch or VGPR spills is a performance bug; this is the check that there is none.
    python scripts/kernel_resources.py [--all] > profiles/rNN_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-I.", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "--cuda-device-only", "-S"]     # = csrc/Makefile


def main():
    show_all = "--all" in sys.argv
    tmp = tempfile.mkdtemp()
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    # per-file flags of the Makefile ("conv_planes.o: CXXFLAGS += -mllvm -pragma-unroll-threshold=65536"): the table must describe the kernels that ship
    extra = {}
    for m in re.finditer(r"^(\w+)\.o:\s*CXXFLAGS\s*\+=\s*(.+)$", open(os.path.join(CSRC, "Makefile")).read(), re.M):
        extra[m.group(1) + ".hip"] = m.group(2).split()
    procs = [(s, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + extra.get(os.path.basename(s), []) + [os.path.basename(s), "-o", os.path.join(tmp, os.path.basename(s) + ".s")],
                                  cwd=CSRC, stderr=subprocess.DEVNULL)) for s in srcs]
    rows = []
    for s, p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed on %s" % s)
        txt = open(os.path.join(tmp, os.path.basename(s) + ".s")).read()
        for m in re.finditer(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)", txt, re.S):
            g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, m.group(0)).group(1)
            rows.append([os.path.basename(s), g("name")] + [int(g(k)) for k in ("vgpr_count", "agpr_count", "private_segment_fixed_size",
                                                                                "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size")])
    names = subprocess.run(["c++filt"], input="\n".join(r[1] for r in rows), capture_output=True, text=True).stdout.strip().split("\n")
    for r, n in zip(rows, names):
        r[1] = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
    bad = [r for r in rows if r[4] > 0 or r[5] > 0]
    print("%d kernels in %d sources; with scratch or VGPR spills: %d; with SGPR spills (v_writelane, no memory): %d; most VGPRs: %d"
          % (len(rows), len(srcs), len(bad), sum(1 for r in rows if r[6] > 0), max(r[2] for r in rows)))
    print("%-18s %5s %5s %7s %6s %6s %6s  kernel" % ("source", "vgpr", "agpr", "scratch", "vspill", "sspill", "lds"))
    for r in sorted(rows, key=lambda r: (r[0], r[1])):
        if show_all or r[4] > 0 or r[5] > 0 or r[6] > 0 or r[2] > 128:
            print("%-18s %5d %5d %7d %6d %6d %6d  %s" % (r[0], r[2], r[3], r[4], r[5], r[6], r[7], r[1][:150]))
    if not show_all:
        print("(listed: kernels with spills or more than 128 VGPRs = one wave per SIMD pair of 256-thread workgroups; --all lists every kernel)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
