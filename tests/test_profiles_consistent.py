"""The committed PMC traffic summary that bench.py cites (`roofline.traffic`) is reproducible from the committed raw
counter sums and the bench line of the same profiling run (tools/pmc_traffic_summary.py) -- CPU only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_summary_is_derived_from_the_committed_raw_counters(tmp_path):
    out = tmp_path / "traffic.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic_summary.py"),
                           os.path.join(ROOT, "profiles", "r02_i_pmc_hbm_raw.json"),
                           os.path.join(ROOT, "profiles", "r02_i_infer_bench.json"), str(out), "r02_i"],
                          stdout=subprocess.DEVNULL)
    new = json.load(open(out))
    old = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_hbm_traffic.json")))
    assert new["kernel"] == old["kernel"] == "conv1d_mfma_dma_kernel"
    for fam in ("conv1d_mfma_dma_kernel", "resunit_kernel"):
        for key in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "launches_per_forward"):
            assert abs(new["kernels"][fam][key] - old["kernels"][fam][key]) <= 1e-9 * old["kernels"][fam][key]
    # what bench.py reads
    assert old["hbm_bytes_per_launch"] == old["kernels"]["conv1d_mfma_dma_kernel"]["hbm_bytes_per_launch"]
    # the bench line of the round cites exactly this number for the same launch mix
    bench = json.load(open(os.path.join(ROOT, "profiles", "r02_i_bench.json")))
    assert bench["roofline"]["traffic"] == old["hbm_bytes_per_launch"]
    assert bench["roofline"]["launches_per_step"] == old["launches_per_forward"]


def test_round3_traffic_summary_is_derived_from_the_committed_raw_counters(tmp_path):
    """Same for round 3 (tools/pmc_round3.sh -> tools/pmc_round3_summary.py: FETCH_SIZE x 2 on gfx950, calibrated in
    the same run), the file bench.py's `roofline.traffic` cites."""
    out = tmp_path / "traffic.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_round3_summary.py"),
                           os.path.join(ROOT, "profiles", "r03_pmc_hbm_raw.json"),
                           os.path.join(ROOT, "profiles", "r03_pmc_infer_bench.json"), str(out)],
                          stdout=subprocess.DEVNULL)
    new = json.load(open(out))
    old = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_hbm_traffic.json")))
    for fam in ("conv1d_mfma_dma_kernel", "resunit_kernel"):
        for key in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch"):
            assert abs(new["kernels"][fam][key] - old["kernels"][fam][key]) <= 1e-9 * old["kernels"][fam][key]
    assert abs(old["calibration"]["fetch_factor_used"] - 2.0) < 1e-3 and abs(old["calibration"]["write_factor_used"] - 1.0) < 1e-3
    conv = old["kernels"]["conv1d_mfma_dma_kernel"]
    assert conv["traffic_over_algorithmic"] < 1.3  # (XCD-aware tile order; 1.88 before it)


def test_round4_traffic_summary_is_derived_from_the_committed_raw_counters(tmp_path):
    """Round 4: the same passes (tools/pmc_round3.sh) re-run on the round's final kernels; bench.py cites this file first."""
    out = tmp_path / "traffic.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_round3_summary.py"),
                           os.path.join(ROOT, "profiles", "r04_pmc_hbm_raw.json"),
                           os.path.join(ROOT, "profiles", "r04_pmc_infer_bench.json"), str(out)],
                          stdout=subprocess.DEVNULL)
    new = json.load(open(out))
    old = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_hbm_traffic.json")))
    for fam in ("conv1d_mfma_dma_kernel", "resunit_kernel"):
        for key in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch"):
            assert abs(new["kernels"][fam][key] - old["kernels"][fam][key]) <= 1e-9 * old["kernels"][fam][key]
    assert abs(old["calibration"]["fetch_factor_used"] - 2.0) < 1e-3 and abs(old["calibration"]["write_factor_used"] - 1.0) < 1e-3
    assert old["kernel"] == "conv1d_mfma_dma_kernel" and old["launches_per_forward"] == 47.0
    assert old["kernels"]["conv1d_mfma_dma_kernel"]["traffic_over_algorithmic"] < 1.3


def test_round5_traffic_summary_kernel_trace_and_bench_line_agree(tmp_path):
    """Round 5 (tools/evidence_round5.sh): (i) the traffic summary bench.py cites first is reproducible from the committed
    raw counter sums and the detail record of the same profiling run; (ii) the committed stdout line of the round cites
    exactly that number; (iii) the conv family's average launch duration in the rocprofv3 kernel trace agrees with the
    HIP-event figure `roofline.avg_launch_us` of the same command (the contract's cross-check) within 2 %."""
    import csv

    out = tmp_path / "traffic.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_round5_summary.py"),
                           os.path.join(ROOT, "profiles", "r05_pmc_hbm_raw.json"),
                           os.path.join(ROOT, "profiles", "r05_infer_bench_detail.json"), str(out)],
                          stdout=subprocess.DEVNULL)
    new = json.load(open(out))
    old = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_hbm_traffic.json")))
    for fam in ("conv1d_mfma_dma_kernel", "resunit_kernel"):
        for key in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch"):
            assert abs(new["kernels"][fam][key] - old["kernels"][fam][key]) <= 1e-9 * old["kernels"][fam][key]
    assert abs(old["calibration"]["fetch_factor_used"] - 2.0) < 1e-3 and abs(old["calibration"]["write_factor_used"] - 1.0) < 1e-3
    assert old["kernel"] == "conv1d_mfma_dma_kernel" and old["launches_per_forward"] == 47.0
    assert old["kernels"]["conv1d_mfma_dma_kernel"]["traffic_over_algorithmic"] < 1.3
    line = json.load(open(os.path.join(ROOT, "profiles", "r05_zz_bench_line.json")))
    assert line["roofline"]["traffic_source"] == "profiles/r05_pmc_hbm_traffic.json"
    assert abs(line["roofline"]["traffic"] - old["hbm_bytes_per_launch"]) <= 1.0
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-3
    # kernel trace of the inference-only command vs the event timing of the same run
    with open(os.path.join(ROOT, "profiles", "r05_infer_kernel_stats.csv")) as f:
        rows = [r for r in csv.reader(l for l in f if not l.startswith("#"))]
    fam = next(r for r in rows if r[0].startswith("FAMILY pwg::conv1d_mfma_dma_kernel"))
    trace_avg_us = float(fam[3])
    ev = json.load(open(os.path.join(ROOT, "profiles", "r05_infer_bench_detail.json")))["roofline"]["avg_launch_us"]
    assert int(fam[1]) == 470 and abs(trace_avg_us - ev) <= 0.02 * trace_avg_us, (trace_avg_us, ev)


def test_round6_traffic_summary_kernel_trace_and_bench_line_agree(tmp_path):
    """Round 6 (tools/evidence_round6.sh, tools/collect_round6.sh): (i) the traffic summary bench.py cites first is
    reproducible from the committed raw counter sums and the detail record of the same profiling run; (ii) the conv family's
    average launch duration in the rocprofv3 kernel trace agrees with the HIP-event figure of the same command and with the
    committed stdout line of the round within 2 %; (iii) the line's roofline fraction is achieved / peak of those figures."""
    import csv

    out = tmp_path / "traffic.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_round5_summary.py"),
                           os.path.join(ROOT, "profiles", "r06_pmc_hbm_raw.json"),
                           os.path.join(ROOT, "profiles", "r06_infer_bench_detail.json"), str(out)],
                          stdout=subprocess.DEVNULL)
    new = json.load(open(out))
    old = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_hbm_traffic.json")))
    for fam in ("conv1d_mfma_dma_kernel", "resunit_kernel"):
        for key in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch"):
            assert abs(new["kernels"][fam][key] - old["kernels"][fam][key]) <= 1e-9 * old["kernels"][fam][key]
    assert abs(old["calibration"]["fetch_factor_used"] - 2.0) < 1e-3 and abs(old["calibration"]["write_factor_used"] - 1.0) < 1e-3
    assert old["kernel"] == "conv1d_mfma_dma_kernel" and old["launches_per_forward"] == 47.0
    assert old["kernels"]["conv1d_mfma_dma_kernel"]["traffic_over_algorithmic"] < 1.3
    trace_avg = None
    with open(os.path.join(ROOT, "profiles", "r06_infer_kernel_stats.csv")) as f:
        for row in csv.reader(r for r in f if not r.startswith("#")):
            if row and row[0].startswith("FAMILY") and "conv1d_mfma_dma_kernel" in row[0]:
                trace_avg = float(row[3])
    assert trace_avg is not None
    infer = json.loads(open(os.path.join(ROOT, "profiles", "r06_infer_bench_line.json")).read())
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06_final_bench_line.json")).read())
    for rec in (infer, line):
        r = rec["roofline"]
        assert r["kernel"] == "conv1d_mfma_dma_kernel" and r["bound"] == "mfma" and r["peak"] == 157.3
        assert abs(r["avg_launch_us"] - trace_avg) <= 0.02 * trace_avg, (r["avg_launch_us"], trace_avg)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
        assert abs(r["achieved"] - r["flops_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12) <= 0.01 * r["achieved"]
    assert line["cpu_baseline"]["kind"] == "reference" and line["parity"]["ok"] and line["train_ok"]
    # the final line was taken AFTER the counter summary was committed: it cites exactly that file's number
    assert line["roofline"]["traffic_source"] == "profiles/r06_pmc_hbm_traffic.json"
    assert abs(line["roofline"]["traffic"] - old["hbm_bytes_per_launch"]) <= 0.01
