#!/usr/bin/env python3
"""Sample the GPU's shader clock / power while a command runs (is the gap to the fp32 MFMA peak DVFS or kernel?).
usage: sample_clocks.py OUT.txt -- <command ...>     (GPU box; reads rocm-smi's sysfs sources directly)"""
import glob
import os
import subprocess
import sys
import time


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return ""


def current_sclk(dev):
    # pp_dpm_sclk lists the DPM levels, the active one carries '*'
    for line in read(os.path.join(dev, "pp_dpm_sclk")).splitlines():
        if line.rstrip().endswith("*"):
            return line.split(":")[1].strip().rstrip("*").strip()
    return "?"


def main():
    out = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    devs = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if os.path.exists(os.path.join(d, "pp_dpm_sclk"))]
    hw = {d: (glob.glob(os.path.join(d, "hwmon/hwmon*")) or [None])[0] for d in devs}
    p = subprocess.Popen(cmd)
    t0 = time.time()
    rows = []
    while p.poll() is None:
        for d in devs:
            h = hw[d]
            power = read(os.path.join(h, "power1_average")) or read(os.path.join(h, "power1_input")) if h else ""
            freq = read(os.path.join(h, "freq1_input")) if h else ""
            busy = read(os.path.join(d, "gpu_busy_percent"))
            rows.append(f"t={time.time() - t0:7.2f} s  {d.split('/')[4]:>7s}  sclk_level {current_sclk(d):>10s}  freq1_input {freq:>12s} Hz  "
                        f"power {power:>10s} uW  busy {busy:>3s} %")
        time.sleep(0.1)
    with open(out, "w") as f:
        f.write(f"# {' '.join(cmd)}\n# devices: {devs}\n" + "\n".join(rows) + "\n")
    sys.exit(p.returncode)


if __name__ == "__main__":
    main()
