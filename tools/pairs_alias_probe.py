"""Key + payload sorts of 2^28 / 2^26 (f32, u32) pairs with the caller's four buffers at chosen distances from one another: adjacent
(n x 4 bytes = a power of two apart, what four torch.empty calls give) or with odd multiples of `pad` bytes between them -- the
level-1 slots lie in the second key / payload buffers (default) or all in scratch memory (RSX_NO_AUX_SLOTS=1).
python tools/pairs_alias_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402


def main():
    rsa.require_gpu()
    for n in (1 << 28, 1 << 26):
        room = 16 * n + (64 << 20)
        base = torch.empty(room, dtype=torch.uint8, device="cuda")
        keys = torch.empty(n, dtype=torch.int32, device="cuda")
        rsa.fill_splitmix(keys, seed=6)
        vals = torch.arange(n, dtype=torch.int32, device="cuda")
        for pad in (0, 64 << 10, 1 << 20, (1 << 20) + (64 << 10), 4 << 10):
            off = [i * 4 * n + (2 * i + 1) * pad * (1 if i else 0) for i in range(4)]
            kin, ka, vin, va = (base[o:o + 4 * n].view(torch.int32) for o in off)
            for name, envs in (("slots in spare buffers", {}), ("RSX_NO_AUX_SLOTS=1", {"RSX_NO_AUX_SLOTS": "1"})):
                os.environ.pop("RSX_NO_AUX_SLOTS", None)
                os.environ.update(envs)
                rsa.reload_env()
                best = 1e9
                for r in range(7):
                    kin.copy_(keys)
                    vin.copy_(vals)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    _, _, info = rsa.radix_sort_pairs(kin, ka, vin, va, dtype=rsa.F32)
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        best = min(best, e0.elapsed_time(e1))
                print("n = 2^%d  buffers %4d KiB (odd multiples) off their adjacent places  %-24s %.3f ms (route %d)" %
                      (n.bit_length() - 1, pad >> 10, name, best, info.hybrid), flush=True)
        del base, keys, vals
        torch.cuda.empty_cache()
    os.environ.pop("RSX_NO_AUX_SLOTS", None)


if __name__ == "__main__":
    main()
