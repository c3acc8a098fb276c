import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29555", RANK="0", WORLD_SIZE="1")
import radix_sorting_amd as rsa
from radix_sorting_amd import multi
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t0 = time.perf_counter()
e = multi.HipEngine(rsa.U32)
s = e.overlap_stream(None, True)
torch.cuda.synchronize()
print("calibration %.3f s -> %s" % (time.perf_counter() - t0, s))
t0 = time.perf_counter()
s = multi.HipEngine(rsa.U32).overlap_stream(None, True)
print("again %.6f s" % (time.perf_counter() - t0))
dist.destroy_process_group()
