import os, sys, importlib.util, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
spec = importlib.util.spec_from_file_location("tel", os.path.join(root, "simple-sfod_amd", "telemetry.py")); tel = importlib.util.module_from_spec(spec); spec.loader.exec_module(tel)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
log = tel.rccl_debug_setup(0)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.ones(48_000_000, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    x.mul_(3.0)
    w = dist.all_reduce(x[:30_000_000], async_op=True)
    y = x[30_000_000:] * 2
    w.wait()
    dist.all_reduce(x[30_000_000:])
    dist.broadcast(x[:8], 0)
    outs = [torch.zeros(4, device="cuda")]
    dist.all_gather(outs, torch.arange(4.0, device="cuda"))
torch.cuda.synchronize(); dist.barrier()
print("sum", x.sum().item(), outs)
dist.destroy_process_group()
print("log", log, os.path.getsize(log) if log and os.path.exists(log) else None)
txt = open(log, errors="replace").read() if log and os.path.exists(log) else ""
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
open(os.path.join(root, "gpurun_out", "rccl_one_rank.log"), "w").write(txt)
print("\n".join(l for l in txt.splitlines() if "WARN" not in l and l.strip())[:5000])
print("SUMMARY", tel.rccl_summary(log))
