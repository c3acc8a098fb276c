"""The local sort of a distributed sort sees keys that an MSD split has already ordered by their top byte, piece by piece: how long
does rsx_sort_inplace_async take on such an array, against the same keys shuffled?  One GPU stands in for rank 0 of G:
2^29 u32 keys below 2^32 / G in G pieces (one per source rank), each piece in stable order of its top byte.
python tools/presplit_probe.py [G = 8] [log2 n = 29]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402


def med(f, reps=7):
    ts = []
    for i in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f(True)
        torch.cuda.synchronize()
        a.record()
        f(False)
        b.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    rsa.require_gpu()
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 29
    n = 1 << log2n
    r = torch.empty(n, dtype=torch.int64, device="cuda")
    rsa.fill_splitmix(r, seed=77)
    keys = (r & ((1 << 32) // G - 1)).to(torch.int32)        # rank 0's share of the key range (values below 2^31: no sign trouble)
    del r
    pieces = []
    for p in range(G):
        piece = keys[p * (n // G):(p + 1) * (n // G)]
        order = torch.sort((piece >> 24) & 0xFF, stable=True).indices
        pieces.append(piece[order])
    presplit = torch.cat(pieces)
    del pieces
    buf, aux = torch.empty_like(keys), torch.empty_like(keys)
    for name, data in (("shuffled", keys), ("pre-split", presplit)):
        for sw in ("", "RSX_PROBE"):
            if sw:
                os.environ[sw] = "4"
            rsa.reload_env()

            def run(prepare):
                if prepare:
                    buf.copy_(data)
                else:
                    rsa.radix_sort_inplace_async(buf, aux, dtype=rsa.U32)
            t = med(run)
            route = rsa.async_route()
            f = buf ^ torch.tensor(-2 ** 31, dtype=torch.int32, device="cuda")
            ok = bool((f[1:] >= f[:-1]).all().item())
            print("G=%d 2^%d keys %-9s %-14s: %.3f ms, route %d, sorted %s" % (G, log2n, name, sw or "default", t, route, ok), flush=True)
            if sw:
                del os.environ[sw]
    rsa.reload_env()


if __name__ == "__main__":
    main()
