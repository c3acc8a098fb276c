"""Route 6 (rsx_logroute.hpp) against the route the same keys take without it, inside one process: Zipf-like u64 keys
(SURVEY.md 8d cfg 3 (iv)) at a range of sizes, RSX_NO_LOG=1 against the default.  python tools/log_sizes.py [bmax=40] [sizes in Mi ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402


def zipf(n, seed, bmax):
    r = torch.empty(n, dtype=torch.int64, device="cuda")
    rsa.fill_splitmix(r, seed=seed)
    b = 1 + (((r >> 58) & 63) % bmax)
    one = torch.ones_like(r)
    return (one << (b - 1)) + (r & ((one << (b - 1)) - 1))


def main():
    rsa.require_gpu()
    bmax = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    sizes = [int(x) for x in sys.argv[2:]] or [16, 24, 32, 48, 64, 96, 128, 192, 256]
    os.environ["RSX_LOG_MIN_LOG2"] = "20"
    for mi in sizes:
        n = mi << 20
        src0 = zipf(n, 33, bmax)
        src, aux = torch.empty_like(src0), torch.empty_like(src0)
        row = []
        for no_log in ("1", "0"):
            os.environ["RSX_NO_LOG"] = no_log
            rsa.reload_env()
            ts = []
            for i in range(7):
                src.copy_(src0)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                res, info = rsa.radix_sort(src, aux, dtype=rsa.U64)
                b.record()
                torch.cuda.synchronize()
                if i >= 2:
                    ts.append(a.elapsed_time(b))
            ts.sort()
            row.append((info.hybrid, ts[len(ts) // 2]))
        print("%4d Mi keys (bmax %d): without route 6: route %d %.3f ms | with: route %d %.3f ms | %.2fx" %
              (mi, bmax, row[0][0], row[0][1], row[1][0], row[1][1], row[0][1] / row[1][1]), flush=True)
        del src0, src, aux
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
