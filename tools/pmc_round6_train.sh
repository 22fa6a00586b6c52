#!/bin/bash
# Round-6 counter pass on the TRAINING shapes (the round-4 script with SQ_WAIT_ANY added): one eager, single-stream C3 (and C5) training run
# per pass -- FETCH_SIZE | WRITE_SIZE | SQ set -- per kernel family: HBM bytes per launch (FETCH_SIZE x 2: the gfx950
# correction measured in round 3, tools/probes/pmc_calib_dma.hip), MFMA instructions, MFMA-pipe busy estimate
# (instructions x 64 clk / (duration x clock x 1024 SIMDs) is not available under PMC timing, so the ratio
# SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES) is reported), wave cycles waiting for issue, LDS bank conflicts.
# --kernel-trace is the only trace domain next to --pmc.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_r06; mkdir -p $O
TAGS=${1:-"c3"}
for T in $TAGS; do
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY"; do
  P=$(echo $PASS | cut -d" " -f1)
  PWG_NO_GRAPH=1 PWG_NO_BRANCH=1 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $O/${T}_$P -o p -- python $R/tools/train_replay.py $T 5 > $O/${T}_$P.log 2>&1
done
done
python - <<PY
import csv, glob, collections, json, re
fams = ("conv1d_mfma_dma_kernel", "conv1d_wgrad_kernel", "conv1d_mfma_kernel", "resunit_kernel", "gconv_fwd_kernel", "gconv_dgrad_kernel",
        "gconv_wgrad_kernel", "mel_fft_kernel", "stft_fft_kernel", "bank_pack_kernel", "reduce_slabs", "splitk_finish_kernel",
        "act_backward_kernel", "conv1d_small_cin", "adam_multi")
out = {"command": "tools/pmc_round6_train.sh: PWG_NO_GRAPH=1 PWG_NO_BRANCH=1 rocprofv3 --pmc <pass> --kernel-trace -- python tools/train_replay.py <cfg> 5  "
                  "(3 passes per config: FETCH_SIZE | WRITE_SIZE | SQ set; 5 eager single-stream training steps of the recipe's own batch; every dispatch of the run per kernel family). "
                  "HBM read = FETCH_SIZE KiB x 2 (gfx950 correction, profiles/r03_pmc_hbm_traffic.json calibration), write = WRITE_SIZE KiB.",
       "configs": {}}
for T in "$TAGS".split():
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob("$O/%s_*/**/*counter_collection.csv" % T, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            fam = next((x for x in fams if x in k), None)
            if fam is None:
                continue
            a = agg[fam][row["Counter_Name"]]
            a[0] += 1; a[1] += float(row["Counter_Value"])
    res = {}
    for fam, d in agg.items():
        def s(c): return d[c][1] if c in d else None
        def n(c): return d[c][0] if c in d else 0
        e = {"dispatches": max(n(c) for c in d)}
        if s("FETCH_SIZE") is not None: e["HBM_read_MB_per_launch_x2_corrected"] = 2.0 * s("FETCH_SIZE") * 1024 / n("FETCH_SIZE") / 1e6
        if s("WRITE_SIZE") is not None: e["HBM_write_MB_per_launch"] = s("WRITE_SIZE") * 1024 / n("WRITE_SIZE") / 1e6
        if s("SQ_INSTS_MFMA"): e["mfma_instructions_per_launch"] = s("SQ_INSTS_MFMA") / n("SQ_INSTS_MFMA")
        if s("SQ_BUSY_CYCLES"): e["mfma_busy_over_sq_busy_x4simd"] = s("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * s("SQ_BUSY_CYCLES"))
        if s("SQ_WAVE_CYCLES"): e["wave_cycles_waiting_for_issue_frac"] = s("SQ_WAIT_INST_ANY") / s("SQ_WAVE_CYCLES")
        if s("SQ_WAVE_CYCLES") and s("SQ_WAIT_ANY") is not None: e["wave_cycles_parked_waitcnt_or_barrier_frac"] = s("SQ_WAIT_ANY") / s("SQ_WAVE_CYCLES")
        if s("SQ_LDS_IDX_ACTIVE"): e["LDS_bank_conflict_cycles_per_LDS_active_cycle"] = s("SQ_LDS_BANK_CONFLICT") / s("SQ_LDS_IDX_ACTIVE")
        res[fam] = e
    out["configs"][T] = res
json.dump(out, open("$O/train_pmc_summary.json", "w"), indent=1)
print(json.dumps({t: {k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in r.items() if "conv1d" in k} for t, r in out["configs"].items()}, indent=1))
PY
