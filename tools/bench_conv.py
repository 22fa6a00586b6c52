"""Micro-benchmark of the conv1d family on the HiFi-GAN V1 problem set.
Sweeps tile configurations per distinct problem; prints achieved TFLOP/s.  GPU box only.
usage: bench_conv.py B F [sweep]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parallelwavegan_amd import ops

CFG = {0:(128,128,8),1:(128,128,16),2:(128,128,4),3:(64,256,8),4:(64,256,16),5:(32,256,8),6:(32,256,16),7:(32,512,8),
       8:(32,512,16),9:(128,64,8),10:(32,128,8),11:(64,256,4),12:(64,128,8),13:(64,64,8),14:(128,32,8),15:(64,64,16),
       16:(32,128,16),17:(64,64,4),18:(128,256,4),19:(128,256,8)}

PRE = None if os.environ.get("PWG_BENCH_ACT") == "0" else "leaky_relu"  # PWG_BENCH_ACT=0: no in-loop pre-activation (round 6 A/B)

def timeit(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 800
    sweep = len(sys.argv) > 3
    dev = torch.device("cuda:0")
    rows = []
    T = F
    rows.append(("input 80->512 k7", 1, dict(c_in=80, c_out=512, kernel=7, dil=1, T=T)))
    ch = 512
    for s, k in zip((8, 8, 2, 2), (16, 16, 4, 4)):
        rows.append((f"convT {ch}->{ch//2} k{k} s{s}", 1, dict(c_in=ch, c_out=ch // 2, kernel=k, stride=s, T=T, transposed=True)))
        ch //= 2; T *= s
        for ks in (3, 7, 11):
            for d in (1, 5):
                rows.append((f"res {ch} k{ks} d{d}", 4 if d == 1 else 2, dict(c_in=ch, c_out=ch, kernel=ks, dil=d, T=T, res=True)))
    rows.append(("output 32->1 k7", 1, dict(c_in=32, c_out=1, kernel=7, dil=1, T=T)))
    tot_ms = tot_fl = 0.0
    for name, count, p in rows:
        k = p["kernel"]
        if p.get("transposed"):
            s = p["stride"]; t_out = p["T"] * s
            desc = ops.make_conv_desc(B, p["c_in"], p["c_out"], p["T"], t_out, k, stride=s, pad_left=s // 2 + s % 2, transposed=True, pre_act=PRE, pre_slope=0.1)
            w = torch.randn(p["c_in"], p["c_out"], k, device=dev) * 0.05
            flops = 2.0 * p["c_in"] * p["c_out"] * k * p["T"] * B
            m_g = p["c_out"] * s
        else:
            d = p["dil"]; t_out = p["T"]
            desc = ops.make_conv_desc(B, p["c_in"], p["c_out"], p["T"], t_out, k, dilation=d, pad_left=(k - 1) // 2 * d, pre_act=PRE, pre_slope=0.1)
            w = torch.randn(p["c_out"], p["c_in"], k, device=dev) * 0.05
            flops = 2.0 * p["c_in"] * p["c_out"] * k * t_out * B
            m_g = p["c_out"]
        wp = ops.pack_weight(desc, w)
        x = torch.randn(B, p["c_in"], p["T"], device=dev)
        bias = torch.randn(p["c_out"], device=dev)
        add1 = torch.randn(B, p["c_out"], t_out, device=dev) if p.get("res") else None
        y = torch.empty(B, p["c_out"], t_out, device=dev)
        ms = timeit(lambda: ops.conv1d_forward(desc, x, wp, bias, add1, out=y))
        line = f"{name:26s} T={p['T']:7d} default {ms*1e3:8.1f} us {flops/ms/1e9:6.1f} TF"
        if sweep:
            ref = ops.conv1d_forward_cfg(desc, x, wp, bias, add1, tile_config=0, use_dma=False).clone()
            res = []
            for cid, (bm, bn, ck) in CFG.items():
                if bm > 32 and m_g <= 32: continue
                if bm > 64 and m_g <= 64: continue
                if bm < 128 and m_g >= 256: continue
                for dma in (1, 0):
                    try:
                        out = ops.conv1d_forward_cfg(desc, x, wp, bias, add1, out=y, tile_config=cid, use_dma=dma)
                        torch.cuda.synchronize()
                        err = (out - ref).abs().max().item()
                        t = timeit(lambda: ops.conv1d_forward_cfg(desc, x, wp, bias, add1, out=y, tile_config=cid, use_dma=dma), reps=5)
                        res.append((t, cid, dma, err))
                    except RuntimeError as e:
                        pass
            res.sort()
            line += " | best: " + "  ".join(f"c{c}{'D' if dm else 'R'}({CFG[c][0]}x{CFG[c][1]}x{CFG[c][2]}) {flops/t/1e9:.0f}TF e{er:.0e}" for t, c, dm, er in res[:4])
            best = res[0][0]
        else:
            best = ms
        tot_ms += ms * count; tot_fl += flops * count
        print(line, flush=True)
    print(f"TOTAL(default) {tot_ms:.2f} ms/step  {tot_fl/tot_ms/1e9:.1f} TFLOP/s  -> {B*F*256/tot_ms*1e3/1e6:.1f} Msamples/s")

if __name__ == "__main__":
    main()
