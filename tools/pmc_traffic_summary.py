#!/usr/bin/env python3
"""HBM bytes per launch of the inference kernels from the raw rocprofv3 PMC sums (tools/profile_round.sh ->
pmc_hbm.json) and the algorithmic bytes bench.py reports for the same workload (infer_bench.json).
FETCH_SIZE / WRITE_SIZE are in KiB; the read side is scaled by the calibration factor measured with
tools/pmc_calib.sh on a convolution with a known byte count (see profiles/r02_pmc_hbm_traffic.json history).
usage: pmc_traffic_summary.py pmc_hbm.json infer_bench.json out.json [tag]"""
import json
import sys

FETCH_FACTOR = 1.454269565988355  # tools/pmc_calib.sh, rounds 1 and 2 (same value both times)


def main():
    raw = json.load(open(sys.argv[1]))
    bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    tag = sys.argv[4] if len(sys.argv) > 4 else ""
    kernels = bench["roofline"].get("kernels", {})
    out = {"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace -- python bench.py "
                      "--steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-train --no-latency --no-extra-configs  "
                      f"(tools/profile_round.sh {tag})",
           "workload": bench["config"]["workload"],
           "units": "FETCH_SIZE / WRITE_SIZE are in KiB (x1024 -> bytes), MI355X_MICROARCH.md HBM section",
           "calibration": {"fetch_factor": FETCH_FACTOR, "write_factor": 1.0,
                           "method": "tools/pmc_calib.sh: conv C=32 k=3 T=204800 B=16 with a known byte count "
                                     "(reads 838.87 MB, writes 419.43 MB); the counter under-reports LDS-DMA reads"},
           "kernels": {}}
    for fam in raw["FETCH_SIZE"]:
        f, w = raw["FETCH_SIZE"][fam], raw["WRITE_SIZE"].get(fam)
        if w is None:
            continue
        rd = f["avg_per_dispatch"] * 1024 * FETCH_FACTOR
        wr = w["avg_per_dispatch"] * 1024
        entry = {"dispatches": f["dispatches"],
                 "launches_per_forward": f["dispatches"] / 5.0,  # --warmup 1 --steps 2 + 2 event-profiled forwards "FETCH_SIZE_KiB_per_launch_raw": f["avg_per_dispatch"],
                 "WRITE_SIZE_KiB_per_launch_raw": w["avg_per_dispatch"], "read_bytes_per_launch": rd,
                 "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
        kb = kernels.get(fam)
        if kb and "algorithmic_bytes_per_launch" in kb:
            entry["algorithmic_bytes_per_launch"] = kb["algorithmic_bytes_per_launch"]
            entry["traffic_over_algorithmic"] = (rd + wr) / kb["algorithmic_bytes_per_launch"]
        out["kernels"][fam] = entry
    dom = bench["roofline"]["kernel"]
    if dom in out["kernels"]:
        out["kernel"] = dom
        out.update({k: v for k, v in out["kernels"][dom].items()})
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
