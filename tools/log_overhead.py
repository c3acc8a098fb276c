"""What an input that is NOT route 6's pays for being asked: u64 keys the sort without a histogram refuses (half of them share their
top two bytes) and route 6's sample refuses too, RSX_NO_LOG=1 against the default, one process.  python tools/log_overhead.py [Mi ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402


def main():
    rsa.require_gpu()
    for mi in [int(x) for x in sys.argv[1:]] or [24, 32, 64, 128, 256]:
        n = mi << 20
        src0 = torch.empty(n, dtype=torch.int64, device="cuda")
        rsa.fill_splitmix(src0, seed=9)
        src0[::2] &= 0x0000FFFFFFFFFFFF          # half of the keys: top two bytes zero -> no slot scheme by bytes, no log digits either
        src, aux = torch.empty_like(src0), torch.empty_like(src0)
        out = []
        for no_log in ("1", "0", "1", "0"):
            os.environ["RSX_NO_LOG"] = no_log
            rsa.reload_env()
            ts = []
            for i in range(9):
                src.copy_(src0)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                res, info = rsa.radix_sort(src, aux, dtype=rsa.U64)
                b.record()
                torch.cuda.synchronize()
                if i >= 3:
                    ts.append(a.elapsed_time(b))
            ts.sort()
            out.append("RSX_NO_LOG=%s route %d %.3f ms" % (no_log, info.hybrid, ts[len(ts) // 2]))
        print("%4d Mi u64 keys: %s" % (mi, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
