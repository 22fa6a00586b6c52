"""Per-kernel register / LDS / scratch usage of a built object (parallelwavegan_amd/csrc/build/<name>.hip.o).

Usage: python tools/kernel_resources.py conv1d [conv1d_wgrad ...]
Reads the gfx950 code object out of the object's .hip_fatbin section and prints the AMDGPU metadata notes:
kernel name, VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs, scratch bytes, static LDS.  Used to check that an edit of
a hot kernel did not change its register allocation (hipcc's allocation for the 2x2-tile convolution waves flips
between 174 and 256 VGPRs on unrelated edits, csrc/conv1d.hip).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    out = {}
    for block in notes.split("  - .agpr_count:")[1:]:
        block = ".agpr_count:" + block
        get = lambda key: (re.search(r"\." + key + r":\s*(\S+)", block) or [None, "?"])[1]
        name = subprocess.check_output(["c++filt", get("name")], text=True).strip()
        out[name] = dict(vgpr=get("vgpr_count"), agpr=get("agpr_count"), sgpr=get("sgpr_count"),
                         vspill=get("vgpr_spill_count"), sspill=get("sgpr_spill_count"),
                         scratch=get("private_segment_fixed_size"), lds=get("group_segment_fixed_size"))
    return out


def main():
    for name in sys.argv[1:] or ["conv1d"]:
        obj = name if os.path.exists(name) else os.path.join(ROOT, "parallelwavegan_amd", "csrc", "build", name + ".hip.o")
        for k, r in sorted(resources(obj).items()):
            k = re.sub(r"^void pwg::", "", k)
            print(f"{r['vgpr']:>4} v {r['agpr']:>3} a {r['sgpr']:>3} s  spill {r['vspill']:>3}/{r['sspill']:>3}  "
                  f"scratch {r['scratch']:>5}  lds {r['lds']:>6}  {k}")


if __name__ == "__main__":
    main()
