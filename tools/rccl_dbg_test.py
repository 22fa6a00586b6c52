"""Debug aid: does bench.py's RCCL log capture work in this environment (world of one)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print({k: v for k, v in os.environ.items() if k.startswith("NCCL") or k.startswith("RCCL") or k == "TMPDIR"}, file=sys.stderr)
import torch, torch.distributed as dist
import bench
bench.enable_rccl_log(0)
print("log path", bench.RCCL_LOG, {k: v for k, v in os.environ.items() if k.startswith("NCCL")}, file=sys.stderr)
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29577"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(70_000_000, device="cuda")
dist.all_reduce(t); torch.cuda.synchronize()
print("exists", os.path.exists(bench.RCCL_LOG["path"]), file=sys.stderr)
print(bench.rccl_log_summary(), file=sys.stderr)
dist.destroy_process_group()
