#!/usr/bin/env python3
"""HBM bytes per launch from the raw rocprofv3 PMC sums of tools/evidence_round5.sh (gpurun_out/prof_r05/pmc_raw.json) next
to the algorithmic bytes of the same launches.  FETCH_SIZE / WRITE_SIZE are in KiB; the calibration factors are
measured in the same run on known byte counts in the product kernels' access patterns (tools/probes/pmc_calib_dma.hip).
usage: pmc_round5_summary.py pmc_raw.json infer_bench.json out.json"""
import json
import sys


def main():
    raw = json.load(open(sys.argv[1]))
    bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    true_kib = 1024 * 1024  # every calibration kernel reads and writes 1 GiB per launch
    cal = {}
    for k in ("dma_copy<4>", "dma_copy<16>", "gld16_copy"):
        cal[k] = {"fetch_factor": true_kib / raw["calib"]["FETCH_SIZE"][k]["avg_KiB_per_dispatch"],
                  "write_factor": true_kib / raw["calib"]["WRITE_SIZE"][k]["avg_KiB_per_dispatch"]}
    ff = cal["dma_copy<16>"]["fetch_factor"]  # all three patterns agree to 1e-4 (and with the guide's x2)
    wf = cal["dma_copy<16>"]["write_factor"]
    out = {"command": "tools/evidence_round5.sh: rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace -- <cmd>",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB (x1024 -> bytes)",
           "calibration": {"per_pattern": cal, "fetch_factor_used": ff, "write_factor_used": wf,
                           "method": "tools/probes/pmc_calib_dma.bin: 1 GiB copied per launch with 4-B LDS-DMA, 16-B LDS-DMA and 16-B "
                                     "global loads: FETCH_SIZE reports exactly 1/2 of the bytes in every pattern (the guide's "
                                     "gfx950 correction), WRITE_SIZE the exact byte count.  The 1.454 of rounds 1-2 came from a "
                                     "convolution whose re-reads were partly served on-die: it under-stated every read figure "
                                     "of those rounds by 27 %."},
           "kernels": {}}

    def entry(run, fam, alg_read, alg_write, note):
        f, w = raw[run]["FETCH_SIZE"][fam], raw[run]["WRITE_SIZE"][fam]
        rd, wr = f["avg_KiB_per_dispatch"] * 1024 * ff, w["avg_KiB_per_dispatch"] * 1024 * wf
        e = {"dispatches": f["dispatches"], "FETCH_SIZE_KiB_per_launch_raw": f["avg_KiB_per_dispatch"],
             "WRITE_SIZE_KiB_per_launch_raw": w["avg_KiB_per_dispatch"], "read_bytes_per_launch": rd,
             "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "note": note}
        if alg_read is not None:
            e.update(algorithmic_read_bytes=alg_read, algorithmic_write_bytes=alg_write,
                     algorithmic_bytes_per_launch=alg_read + alg_write,
                     traffic_over_algorithmic=(rd + wr) / (alg_read + alg_write),
                     reads_over_algorithmic=rd / alg_read, writes_over_algorithmic=wr / alg_write)
        return e

    kern = bench["roofline"].get("kernels", {})
    for fam in ("conv1d_mfma_dma_kernel", "resunit_kernel"):
        alg = kern.get(fam, {}).get("algorithmic_bytes_per_launch")
        f, w = raw["bench"]["FETCH_SIZE"][fam], raw["bench"]["WRITE_SIZE"][fam]
        rd, wr = f["avg_KiB_per_dispatch"] * 1024 * ff, w["avg_KiB_per_dispatch"] * 1024 * wf
        e = {"dispatches": f["dispatches"], "launches_per_forward": f["dispatches"] / 5.0,
             "FETCH_SIZE_KiB_per_launch_raw": f["avg_KiB_per_dispatch"], "WRITE_SIZE_KiB_per_launch_raw": w["avg_KiB_per_dispatch"],
             "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
        if alg:
            e["algorithmic_bytes_per_launch"] = alg
            e["traffic_over_algorithmic"] = (rd + wr) / alg
        out["kernels"][fam] = e
    out["workload"] = bench["config"]["workload"]
    dom = bench["roofline"]["kernel"]
    out["kernel"] = dom
    out.update({k: v for k, v in out["kernels"][dom].items()})
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, e in out["kernels"].items():
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items() if a in (
            "hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic", "hbm_bytes_per_sample",
            "reads_over_algorithmic", "writes_over_algorithmic", "launches_per_forward")})


if __name__ == "__main__":
    main()
