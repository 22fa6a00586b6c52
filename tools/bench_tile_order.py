"""A/B of the convolution kernel's two logical tile orders (csrc/conv1d.hip: tile_of_workgroup / choose_tile_order) on the
weight-heavy HiFi-GAN discriminator shapes at the C3 training batch, plus generator-sized control shapes.

The order is read once per process (PWG_TILE_ORDER=0: x-window-major everywhere, the order of rounds 1-4; unset: the
planner's choice), so run it twice and compare:

    PWG_TILE_ORDER=0 python tools/bench_tile_order.py > order0.txt
    python tools/bench_tile_order.py > auto.txt
    python tools/bench_tile_order.py --compare order0.txt auto.txt

Per shape: the plan (tile, slices, order), the time of back-to-back launches (operands warm in the L2s / Infinity
Cache), the time with a 1 GiB fill between launches (operands from HBM, the situation inside a training step, where
a layer's weights were last touched hundreds of launches earlier), and a checksum of the output: the two orders must
agree BIT FOR BIT (the order only decides which workgroup computes which tile).  GPU box only.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def compare(a, b):
    def load(p):
        rows = {}
        for line in open(p):
            f = line.rstrip("\n").split("|")
            if len(f) == 6:
                rows[f[0].strip()] = [s.strip() for s in f[1:]]
        return rows

    ra, rb = load(a), load(b)
    tot = [0.0, 0.0, 0.0, 0.0]
    print(f"{'shape':46s} {'plan (B)':24s} {'warm us A -> B':>22s} {'cold us A -> B':>22s}  bits")
    ok = True
    for name in ra:
        if name not in rb:
            continue
        pa, wa, ca, ha, _ = ra[name]
        pb, wb, cb, hb, _ = rb[name]
        wa, wb, ca, cb = float(wa), float(wb), float(ca), float(cb)
        same = ha == hb
        ok &= same
        changed = "im" in pb.split()
        if changed:
            tot[0] += wa
            tot[1] += wb
            tot[2] += ca
            tot[3] += cb
        print(f"{name:46s} {pb:24s} {wa:9.1f} -> {wb:9.1f} {'*' if changed else ' '} {ca:9.1f} -> {cb:9.1f} {'*' if changed else ' '}  "
              f"{'same' if same else 'DIFFERENT'}")
    print(f"item-major shapes, one launch each: warm {tot[0]:.1f} -> {tot[1]:.1f} us, cold {tot[2]:.1f} -> {tot[3]:.1f} us; "
          f"outputs {'bit-identical' if ok else 'DIFFER'}")
    return 0 if ok else 1


def build_shapes(T=8192):
    shapes = []
    for p in (2, 5, 11):
        rows = -(-T // p)
        r = [rows]
        for _ in range(4):
            r.append((r[-1] + 4 - 5) // 3 + 1)
        shapes.append((f"mpd p{p} 512->1024 (5,1) s3", dict(c_in=512, c_out=1024, t_in=r[3], t_out=r[4], k=5, stride=3, pad=2, width=p)))
        shapes.append((f"mpd p{p} 1024->1024 (5,1)", dict(c_in=1024, c_out=1024, t_in=r[4], t_out=r[4], k=5, pad=2, width=p)))
        shapes.append((f"mpd p{p} dgrad of 512->1024 s3", dict(c_in=1024, c_out=512, t_in=r[4], t_out=r[3], k=5, stride=3, pad=2, width=p, transposed=True)))
        shapes.append((f"mpd p{p} dgrad of 1024->1024", dict(c_in=1024, c_out=1024, t_in=r[4], t_out=r[4], k=5, pad=2, width=p, transposed=True)))
        shapes.append((f"mpd p{p} 1024->1 (3,1)", dict(c_in=1024, c_out=1, t_in=r[4], t_out=r[4], k=3, pad=1, width=p)))
    for t in (32, 17, 9):
        shapes.append((f"msd T{t} 1024->1024 k5", dict(c_in=1024, c_out=1024, t_in=t, t_out=t, k=5, pad=2)))
        shapes.append((f"msd T{t} dgrad of 1024->1024 k5", dict(c_in=1024, c_out=1024, t_in=t, t_out=t, k=5, pad=2, transposed=True)))
        shapes.append((f"msd T{t} 1024->1024 k41 g16", dict(c_in=1024, c_out=1024, t_in=t, t_out=t, k=41, pad=20, groups=16)))
        shapes.append((f"msd T{t} 512->1024 k41 s4 g16", dict(c_in=512, c_out=1024, t_in=4 * t - 3 if t != 32 else 128, t_out=t, k=41, stride=4, pad=20, groups=16)))
    # generator layers of the C3 step (B16 x 32 frames) and controls at the inference batch
    shapes.append(("gen convT 512->256 k16 s8 T32", dict(c_in=512, c_out=256, t_in=32, t_out=256, k=16, stride=8, pad=4, transposed=True)))
    shapes.append(("gen res 256 k11 T256", dict(c_in=256, c_out=256, t_in=256, t_out=256, k=11, pad=5)))
    shapes.append(("gen res 128 k11 T2048", dict(c_in=128, c_out=128, t_in=2048, t_out=2048, k=11, pad=5)))
    shapes.append(("ctrl res 128 k11 T51200", dict(c_in=128, c_out=128, t_in=51200, t_out=51200, k=11, pad=5)))
    shapes.append(("ctrl res 256 k7 T6400", dict(c_in=256, c_out=256, t_in=6400, t_out=6400, k=7, pad=3)))
    return shapes


def main():
    import hashlib

    import torch

    from parallelwavegan_amd import ops

    B = 16
    dev = torch.device("cuda:0")
    shapes = build_shapes()

    flush = torch.empty(1 << 28, device=dev)  # 1 GiB: larger than the 256 MB Infinity Cache
    gen = torch.Generator(device=dev).manual_seed(7)
    for name, p in shapes:
        w_ = p.get("width", 1)
        g = p.get("groups", 1)
        tr = p.get("transposed", False)
        st = p.get("stride", 1)
        desc = ops.make_conv_desc(B, p["c_in"], p["c_out"], p["t_in"], p["t_out"], p["k"], stride=st, pad_left=p["pad"],
                                  groups=g, transposed=tr, width=w_, pre_act="leaky_relu", pre_slope=0.1)
        wshape = (p["c_in"], p["c_out"] // g, p["k"]) if tr else (p["c_out"], p["c_in"] // g, p["k"])
        w = torch.randn(wshape, device=dev, generator=gen) * 0.03
        wp = ops.pack_weight(desc, w)
        x = torch.randn(B, p["c_in"], p["t_in"] * w_, device=dev, generator=gen)
        bias = torch.randn(p["c_out"], device=dev, generator=gen)
        y = torch.empty(B, p["c_out"], p["t_out"] * w_, device=dev)
        plan = ops.conv1d_plan(desc)
        tag = f"{plan['family']} c{plan['tile_config']} s{plan['ksplit']}{' im' if plan['item_major'] else ''}"
        run = lambda: ops.conv1d_forward(desc, x, wp, bias, out=y)  # noqa: E731
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        warm = e0.elapsed_time(e1) / 20 * 1e3
        cold = 0.0
        for _ in range(6):
            flush.fill_(1.0)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            cold += e0.elapsed_time(e1) * 1e3 / 6
        digest = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16]
        flops = 2.0 * p["c_in"] * (p["c_out"] // g) * p["k"] * (p["t_in"] if tr else p["t_out"]) * w_ * B
        print(f"{name:46s}| {tag:24s}| {warm:9.1f}| {cold:9.1f}| {digest}| {flops / warm / 1e6:6.1f} TF warm {flops / cold / 1e6:6.1f} TF cold",
              flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--compare":
        sys.exit(compare(sys.argv[2], sys.argv[3]))
    main()
