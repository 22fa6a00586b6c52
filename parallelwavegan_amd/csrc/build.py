"""Build libpwgkernels.so for gfx950 with hipcc (in-tree, no JIT cache).

Usage: python -m parallelwavegan_amd.csrc.build [--force]
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, "libpwgkernels.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# the MFMA kernels never produce or consume NaNs on purpose; without IEEE-mode sNaN quieting the
# in-loop LeakyReLU is v_mul + v_max instead of three instructions (see csrc/conv1d.hip, ACT == 1)
EXTRA = {"conv1d.hip": ["-fno-honor-nans", "-mno-amdgpu-ieee"],
         "conv1d_wgrad.hip": ["-fno-honor-nans", "-mno-amdgpu-ieee"],
         "resunit.hip": ["-fno-honor-nans", "-mno-amdgpu-ieee"],
         # no FMA contraction: the spectra of the two signals of a frame are formed by the same source expressions and
         # must round identically (loss and gradient of (x, x) are exactly 0, as with torch.stft); hipcc otherwise
         # contracts re*re + im*im differently in the two inlined copies
         "stft_fft.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(HERE, "*.h")) + glob.glob(
        os.path.join(PKG, "..", "include", "*.h")
    )
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        hdrs = glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(PKG, "..", "include", "*.h"))
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(p) for p in [src] + hdrs)):
            continue
        cmd = [HIPCC] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
