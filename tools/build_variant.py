"""Build a variant of libpwgkernels.so with extra compile-time definitions for ONE source file, for same-box A/B runs
of kernel experiments (the other objects are the in-tree build's).

    python tools/build_variant.py prio2 conv1d -DPWG_PRIO=2
      -> parallelwavegan_amd/libpwgkernels_prio2.so      (use: PWG_KERNEL_LIB=<that path> python tools/...)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parallelwavegan_amd.csrc import build as kbuild  # noqa: E402


def main():
    tag, src, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
    kbuild.build(verbose=False)  # the regular objects
    name = src + ".hip"
    obj = os.path.join(kbuild.HERE, "build", f"{name}.{tag}.o")
    cmd = [kbuild.HIPCC] + kbuild.FLAGS + kbuild.EXTRA.get(name, []) + defs + ["-c", os.path.join(kbuild.HERE, name), "-o", obj]
    subprocess.check_call(cmd)
    objs = [os.path.join(kbuild.HERE, "build", os.path.basename(s) + ".o") for s in kbuild.sources()]
    objs = [obj if os.path.basename(o) == name + ".o" else o for o in objs]
    lib = os.path.join(kbuild.PKG, f"libpwgkernels_{tag}.so")
    subprocess.check_call([kbuild.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
