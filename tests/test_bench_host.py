"""Host-side pieces of bench.py and of the CPU-baseline plumbing (no GPU needed)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("pwg_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# An 8-rank RCCL INFO log as NCCL_DEBUG=INFO / NCCL_DEBUG_SUBSYS=INIT,TUNING,GRAPH writes it.  The line FORMATS are those
# of the NCCL 2.2x sources RCCL is built from (init.cc "comm %p rank %d nranks %d cudaDev %d ... Init COMPLETE",
# "%d coll channels, %d collnet channels, %d nvls channels, %d p2p channels ...", transport/p2p.cc "Channel %02d/%01d :
# %d[%d] -> %d[%d] via P2P/IPC", enqueue.cc "%s: %ld Bytes -> Algo %s proto %s channel{Lo..Hi}={%d..%d}" -- older
# builds print the numeric ids, "Algo %d proto %d"); host names, pointers and sizes are made up.  No such log could be
# captured on the one-GPU boxes (a world of one short-circuits its collectives before the tuner logs anything).
RCCL_LOG_8 = """\
node0:4127:4127 [0] NCCL INFO Bootstrap : Using eth0:10.0.0.5<0>
node0:4127:4127 [0] NCCL INFO NET/Plugin: Failed to find ncclNetPlugin_v8 symbol.
node0:4127:4306 [0] NCCL INFO comm 0x5581a2c40e10 rank 0 nRanks 8 nNodes 1 localRanks 8 localRank 0 MNNVL 0
node0:4127:4306 [0] NCCL INFO Channel 00/32 :    0   1   2   3   4   5   6   7
node0:4127:4306 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1 [1] 1/-1/-1->0->-1
node0:4127:4306 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC
node0:4127:4306 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC
node0:4127:4306 [0] NCCL INFO Connected all rings
node0:4127:4306 [0] NCCL INFO 32 coll channels, 0 collnet channels, 0 nvls channels, 32 p2p channels, 4 p2p channels per peer
node0:4127:4306 [0] NCCL INFO comm 0x5581a2c40e10 rank 0 nranks 8 cudaDev 0 nvmlDev 0 busId 5000 commId 0x3c1e - Init COMPLETE
node0:4127:4127 [0] NCCL INFO AllReduce: 67108864 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..31}
node0:4127:4127 [0] NCCL INFO AllReduce: 67108864 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..31}
node0:4127:4127 [0] NCCL INFO AllReduce: 13107200 Bytes -> Algo TREE proto LL128 channel{Lo..Hi}={0..15}
node0:4127:4127 [0] NCCL INFO AllReduce: 8 Bytes -> Algo 0 proto 0 time 11.200000
node0:4127:4127 [0] NCCL INFO Broadcast: 4096 Bytes -> Algo 1 proto 2 time 7.900000
"""


def test_rccl_log_summary_parses_a_multi_rank_log(tmp_path):
    """the first 8-GPU bench line must not come back with ``algo_proto_counts: {}``."""
    b = _bench()
    path = tmp_path / "rccl.log"
    path.write_text(RCCL_LOG_8)
    b.RCCL_LOG["path"] = str(path)
    s = b.rccl_log_summary()
    assert s["nranks"] == 8 and s["coll_channels"] == 32
    assert s["transports"] == ["P2P/IPC"]
    assert s["algo_proto_counts"] == {"AllReduce:Ring/Simple": 2, "AllReduce:Tree/LL128": 1, "AllReduce:Tree/LL": 1,
                                      "Broadcast:Ring/Simple": 1}
    assert len(s["sample_lines"]) == 5 and "67108864" in s["sample_lines"][0]
    # what the scaling judgement needs: did the big buckets run as a ring, and over how many channels
    assert s["large_allreduce"] == {"bytes": 67108864, "algo": "Ring", "proto": "Simple", "channels": 32}
    b.RCCL_LOG["path"] = str(tmp_path / "absent.log")
    assert b.rccl_log_summary() is None


def test_train_config_table_names_every_baseline_training_config():
    b = _bench()
    assert set(b.TRAIN_CONFIGS) == {"c2", "c3", "c4", "c5"}
    for tag, name in b.TRAIN_CONFIGS.items():
        conf = b.load_conf(name)
        assert conf["batch_max_steps"] % conf["hop_size"] == 0, tag
    c5 = b.load_conf(b.TRAIN_CONFIGS["c5"])
    # reference egs/libritts/voc1/conf/hifigan.v1.yaml:40,100-102,128-129
    assert c5["generator_params"]["upsample_scales"] == [5, 5, 4, 3]
    assert (c5["mel_loss_params"]["fft_size"], c5["mel_loss_params"]["hop_size"], c5["mel_loss_params"]["win_length"]) == (2048, 300, 1200)
    assert (c5["batch_size"], c5["batch_max_steps"], c5["sampling_rate"]) == (16, 8400, 24000)
    assert round(b.hifigan_macs_per_sample(b.load_conf("hifigan.v1")["generator_params"])) == 1199424


def test_staged_reference_copy_is_byte_identical_to_the_reference():
    """oracle/make_ref.py stages the reference's package for the CPU baseline: sha256 of every file == the manifest
    taken at staging time, and (in the build container) == the file under /root/reference."""
    from oracle import make_ref

    if not os.path.isdir(os.path.join(make_ref.DST_ROOT, "parallel_wavegan")):
        pytest.skip("oracle/_ref not staged (python oracle/make_ref.py)")
    assert make_ref.verify()
    if os.path.isdir(os.path.join(make_ref.SRC_ROOT, "parallel_wavegan")):
        m = make_ref.stage()
        for rel, h in m["files"].items():
            assert make_ref._sha(os.path.join(make_ref.SRC_ROOT, rel)) == h, rel


def test_product_package_never_imports_the_oracle_or_the_staged_reference():
    bad = []
    for base, _, names in os.walk(os.path.join(ROOT, "parallelwavegan_amd")):
        for n in names:
            if n.endswith(".py"):
                with open(os.path.join(base, n)) as f:
                    src = f.read()
                for needle in ("import oracle", "from oracle", "oracle._ref", "ref_shim", "ref_run"):
                    if needle in src:
                        bad.append((os.path.join(base, n), needle))
    assert not bad, bad


def _canned_record(world=1):
    """A full bench record of the shape main() builds: the committed round-4 record plus the keys round 5 added."""
    import json

    with open(os.path.join(ROOT, "profiles", "r04_z_bench.json")) as f:
        out = json.load(f)
    out["train_config"] = "c3" if world == 1 else "c5"
    out["detail_file"] = "gpurun_out/bench_detail.json"
    out["roofline"]["launches_timed"] = 141
    out["train"]["cpu_baseline"].setdefault("batch", 16)
    out["n_gpus"] = world
    # round 6: B = 1 latencies, the MB-MelGAN inference leg and the two further CPU baselines ride in the line
    out["latency"] = {"B1_F100": {"ms": 1.5012345, "samples_per_s": 1.7e7, "rtf": 1e-3},
                      "B1_F800": {"ms": 5.3912345, "samples_per_s": 3.8e7, "rtf": 1e-3}}
    leg = {"batch": {"samples_per_s": 123456789.123, "parity": {"max_abs_vs_oracle": 2.123456e-6}},
           "B1_F100": {"samples_per_s": 23456789.123}}
    out.setdefault("configs", {})["c1_pwg_inference"] = leg
    out["configs"]["c4_mbmelgan_inference"] = leg
    out["cpu_baseline"]["others"] = {"pwg_v1_c0": {"value": 51234.5678}, "mb_melgan_v2": {"value": 812345.678}}
    if world > 1:  # the distributed detail a --gpus 8 run carries
        out["train"]["dist"] = {
            "backend": "nccl", "world_size": world,
            "exposed_comm_ms_per_exchange": {"generator": 0.61234567, "discriminator": 0.123456789},
            "rccl": {"nranks": world, "large_allreduce": {"bytes": 67108864, "algo": "Ring", "proto": "Simple",
                                                          "channels": 32}, "sample_lines": ["x" * 200] * 6}}
    return out


@pytest.mark.parametrize("world", [1, 8])
def test_stdout_line_is_compact_strict_json_with_the_contract_keys(world):
    """VERDICT r04 item 1 / ADVICE r04: the driver keeps a 2000-byte tail of stdout; the ONE stdout line has to be a
    standalone JSON object inside it that carries the contract keys, ``roofline`` and ``cpu_baseline``."""
    import json

    bench = _bench()
    line = bench.compact_line(_canned_record(world))
    assert "\n" not in line and len(line.encode()) < 2000, len(line)
    rec = json.loads(line, parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c}"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "train_steps_per_s", "train_ok",
              "hip_graph"):
        assert k in rec, k
    assert rec["config"]["workload"] and "model" not in rec["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rec["roofline"], k
    assert abs(rec["roofline"]["frac"] - rec["roofline"]["achieved"] / rec["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    assert set(rec["train_steps_per_s"]) == {"c2", "c3", "c4", "c5"}
    # B = 1 latencies, MB-MelGAN / PWG inference and the three CPU baselines in the parsed line
    assert set(rec["lat_ms"]) == {"b1f100", "b1f800"}
    assert set(rec["infer_Msps_b16_b1_err"]) == {"pwg", "mb"} and len(rec["infer_Msps_b16_b1_err"]["mb"]) == 3
    assert rec["cpu_baseline"]["pwg_v1_c0"] > 0 and rec["cpu_baseline"]["mb_melgan_v2"] > 0
    if world > 1:
        d = rec["train"]["dist"]
        assert d["rccl"]["nranks"] == world and d["rccl"]["large_allreduce"]["algo"] == "Ring"
        assert set(d["exposed_comm_ms_per_exchange"]) == {"generator", "discriminator"}


def test_compact_line_sheds_optional_objects_rather_than_exceed_the_tail():
    bench = _bench()
    out = _canned_record(8)
    out["train"]["dist"]["rccl"]["large_allreduce"]["algo"] = "R" * 3000  # a pathological field
    line = bench.compact_line(out)
    assert len(line.encode()) < 2000
    import json

    rec = json.loads(line)
    assert "roofline" in rec and "cpu_baseline" in rec and "train_steps_per_s" in rec
