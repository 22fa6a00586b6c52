"""gcc build of the plain-C oracle -> oracle/c/libpwgoracle.so (TEST INFRASTRUCTURE)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "pwg_oracle.c")
LIB = os.path.join(HERE, "libpwgoracle.so")


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"])
    return LIB


if __name__ == "__main__":
    print(build())
