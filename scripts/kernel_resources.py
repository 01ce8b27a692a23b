"""Register / scratch / LDS figures of every kernel of a built library, from the code object's metadata.

    python scripts/kernel_resources.py gym_anm_amd/_build/libanm_<topology>.so [name filter]
"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    lib = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "gfx950.co")
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + lib, "--output=" + co],
                       check=False, capture_output=True)
        if not os.path.exists(co) or os.path.getsize(co) == 0:
            # the fat binary sits in a section of the shared library
            sec = os.path.join(td, "fat.bin")
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + sec, lib], check=True)
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + sec, "--output=" + co], check=True)
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count")[1:]:
        blk = ".agpr_count" + blk
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt and flt not in dem:
            continue
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))  # noqa: E731
        print("%-70s vgpr %3d agpr %3d sgpr %3d scratch %5d B lds %6d B" % (dem[:70], g("vgpr_count"), g("agpr_count"), g("sgpr_count"),
                                                                         g("private_segment_fixed_size"), g("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
