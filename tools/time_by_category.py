"""Group the per-(kernel, shape) table of tools/profile_train_shapes.py (one eager, serial C3 / C5 step) into the layer
categories of profiles/r05_c3_time_by_category.txt; with two files, before / after side by side.

    python tools/time_by_category.py before.txt [after.txt] > profiles/r06_c3_time_by_category.txt
"""
import re
import sys

ROW = re.compile(r"\s*([\d.]+) ms\s+[\d.]+% cum\s+[\d.]+%\s+n=\s*(\d+)\s+([\d.]+) TF\s+(\d+) GB/s\s+(.*)")


def category(name):
    if name.startswith("conv1d_wgrad_kernel"):
        if re.search(r"Co1024 Ci1024 k5 ", name):
            return "weight gradient, 1024 x 1024 k5"
        if re.search(r" s[2-9]\d* ", name):
            return "weight gradient, strided layers (stride 3 / 4 / 8)"
        return "weight gradient, other"
    if name.startswith("conv1d_mfma_dma_kernel") or name.startswith("conv1d_mfma_kernel"):
        m = re.search(r"cols(\d+)", name)
        cols = int(m.group(1)) if m else 0
        if re.search(r"\(x[2-9]\d*ph\)", name) or re.search(r" s[2-9]\d* ", name) and "dgrad" in name:
            return "fwd / dgrad, polyphase (transposed layers, data gradient of strided layers)"
        if re.search(r"^conv1d_mfma_dma_kernel B1 ", name):
            return "fwd / dgrad, batch-folded scale-discriminator tail layers (one item)"
        if cols <= 40:
            return "fwd / dgrad, <= 40 columns per item"
        if cols <= 128:
            return "fwd / dgrad, 41 - 128 columns per item (period discriminators, 512 / 1024 channels)"
        if cols <= 600:
            return "fwd / dgrad, 129 - 600 columns per item"
        return "fwd / dgrad, > 600 columns per item (generator, first discriminator layers)"
    if name.startswith("gconv_"):
        return "grouped k = 41 layers (gconv)"
    if name.startswith("resunit_kernel"):
        return "residual units (no-grad generator pass)"
    return "helpers without matrix work (finishers, activation backward, optimizers, losses ...)"


def load(path):
    cat = {}
    for line in open(path):
        m = ROW.match(line)
        if not m:
            continue
        ms, n, tf = float(m.group(1)), int(m.group(2)), float(m.group(3))
        c = cat.setdefault(category(m.group(5).strip()), [0.0, 0, 0.0])
        c[0] += ms
        c[1] += n
        c[2] += ms * tf  # -> TFLOP-weighted
    return cat


files = sys.argv[1:]
tabs = [load(f) for f in files]
keys = sorted(set().union(*tabs), key=lambda k: -tabs[0].get(k, [0])[0])
hdr = f"{'category':88s}" + "".join(f" | {'ms':>7s} {'launches':>8s} {'TFLOP/s':>8s}" for _ in tabs)
print(hdr)
for k in keys:
    line = f"{k:88s}"
    for t in tabs:
        ms, n, w = t.get(k, [0.0, 0, 0.0])
        line += f" | {ms:7.2f} {n:8d} {(w / ms if ms else 0.0):8.1f}"
    print(line)
line = f"{'total':88s}"
for t in tabs:
    tot = sum(v[0] for v in t.values())
    mat = sum(v[0] for k_, v in t.items() if not k_.startswith("helpers"))
    line += f" | {tot:7.2f} (matrix kernels {mat:.2f})"
print(line)
