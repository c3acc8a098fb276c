import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa
if os.environ.get("RSX_LIBV"):
    rsa.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "bisect", "librsx_%s.so" % os.environ["RSX_LIBV"])
rsa.require_gpu()
n = int(sys.argv[1])
b = torch.empty(n, dtype=torch.int32, device="cuda"); aux = torch.empty_like(b)
rsa.fill_splitmix(b, 5)
_, info = rsa.radix_sort(b, aux, rsa.U32)
torch.cuda.synchronize()
r = b if not info.result_in_aux else aux
print(n, os.environ.get("RSX_LIBV"), os.environ.get("RSX_NO_AUX_SLOTS"), "route", info.hybrid, "sorted", bool((r[1:].view(torch.int32).to(torch.int64) & 0xFFFFFFFF >= r[:-1].to(torch.int64) & 0xFFFFFFFF).all()))
