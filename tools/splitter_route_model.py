#!/usr/bin/env python3
"""CPU model of the route DESIGN.md 8.2 proposes for skewed keys (not built): how evenly would two passes fill their slots if
the level-1 digit were a RANGE number -- 256 ranges of near-equal population cut from the histogram of the keys' top sixteen
bits -- and the level-2 digit (key - lo_b) >> shift_b with the range's own origin and shift?

For BASELINE.json's skewed inputs at 2^24 keys (the distributions scale): the level-1 slots' fill against the mean, the
(range, digit) buckets' fill against the mean of n / 65536 and against a slot of slot_cap_for(mean) keys, how many ranges are a
single top-16 value (keys equal in their top half: a stable level-1 pass leaves their low halves to be sorted, nothing else), and
how many bits a leaf of each range has to sort by (16 or fewer: two-byte slots as today; more: four-byte slots).
numpy only; test infrastructure / design note, not part of the product."""
import numpy as np

N = 1 << 24
rng = np.random.default_rng(7)

def splitmix(n, seed):
    x = (np.arange(n, dtype=np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))

def f32_kdf(bits):                      # radix_sort_basic_kdf.hpp: floats sort as their bits with the sign handled
    b = bits.astype(np.uint32)
    neg = (b >> np.uint32(31)).astype(bool)
    return np.where(neg, ~b, b ^ np.uint32(0x80000000)).astype(np.uint32)

def slot_cap_for(mean):
    need = max(mean + mean // 4, mean + 7 * (int(np.sqrt(mean)) + 1) + 8)
    return (need + 255) // 256 * 256

def model(name, keys, bits):
    keys = keys.astype(np.uint64)
    top16 = (keys >> np.uint64(bits - 16)).astype(np.int64)
    hist = np.bincount(top16, minlength=65536)
    target = N / 256.0
    # greedy cut: a range ends when adding the next top-16 value would pass the target; a value heavier than the target is alone
    bounds, acc = [0], 0
    for v in range(65536):
        h = hist[v]
        if acc and acc + h > target * 1.0:
            bounds.append(v); acc = 0
        acc += h
        if h > target and acc == h:
            bounds.append(v + 1); acc = 0
    if bounds[-1] != 65536:
        bounds.append(65536)
    bounds = sorted(set(bounds))
    nr = len(bounds) - 1
    rng_of = np.zeros(65536, dtype=np.int64)
    for r in range(nr):
        rng_of[bounds[r]:bounds[r + 1]] = r
    r1 = rng_of[top16]
    size1 = np.bincount(r1, minlength=nr)
    single = sum(1 for r in range(nr) if bounds[r + 1] - bounds[r] == 1)
    heavy_single = [(r, size1[r]) for r in range(nr) if bounds[r + 1] - bounds[r] == 1 and size1[r] > 2 * target]
    # level 2: per range, the smallest shift that brings its span under 256 digits
    span_top = np.array([bounds[r + 1] - bounds[r] for r in range(nr)], dtype=np.int64)
    span_bits = np.ceil(np.log2(np.maximum(span_top, 1))).astype(np.int64) + (bits - 16)      # bits the range's keys vary in
    shift = np.maximum(span_bits - 8, 0)
    lo = np.array(bounds[:-1], dtype=np.uint64) << np.uint64(bits - 16)
    d2 = ((keys - lo[r1]) >> shift[r1].astype(np.uint64)).astype(np.int64)
    assert d2.max() < 256
    b2 = np.bincount(r1 * 256 + d2, minlength=nr * 256)
    mean2 = N // 65536
    cap2 = slot_cap_for(mean2)
    over = b2[b2 > cap2]
    print("%s" % name)
    print("  ranges: %d (of them one top-16 value each: %d; heavier than twice the mean: %s)" % (nr, single, [(int(a), int(b)) for a, b in heavy_single][:6]))
    print("  level-1 slots: mean %.0f keys, largest %.2f x the mean (without the single-value ranges: %.2f x)" % (
        target, size1.max() / target, max([size1[r] for r in range(nr) if bounds[r + 1] - bounds[r] > 1] or [0]) / target))
    print("  (range, digit) buckets: mean %d, slot %d; non-empty %d of %d; largest %d keys; %d buckets (%.2f %% of the keys) above a slot" % (
        mean2, cap2, int((b2 > 0).sum()), nr * 256, int(b2.max()), len(over), 100.0 * over.sum() / N))
    lb = shift
    w = size1 / N
    print("  bits left to a leaf, by share of the keys: <= 8: %.1f %%, 9-16: %.1f %%, 17-24: %.1f %%, more: %.1f %%" % (
        100 * w[lb <= 8].sum(), 100 * w[(lb > 8) & (lb <= 16)].sum(), 100 * w[(lb > 16) & (lb <= 24)].sum(), 100 * w[lb > 24].sum()))

r = splitmix(N, 21)
# cfg 4 (ii): f32 uniform in [-1, 1)   (tools/bench_configs.py, mk_pm1)
f = ((((r >> np.uint64(40)) & np.uint64(0xFFFFFF)).astype(np.int64) - (1 << 23)).astype(np.float32) * np.float32(2.0 ** -23))
model("cfg 4 (ii): f32 uniform in [-1, 1)", f32_kdf(f.view(np.uint32)), 32)
# cfg 4 (iii): f32 bit patterns & 0xFFF000FF
model("cfg 4 (iii): f32 & 0xFFF000FF", f32_kdf((r & np.uint64(0xFFF000FF)).astype(np.uint32)), 32)
# cfg 3 (iv): u64 Zipf-like: key = 2^(b-1) + low bits, b = 1 + (r >> 58) % 40
b = (np.uint64(1) + ((r >> np.uint64(58)) & np.uint64(63)) % np.uint64(40))
one = np.uint64(1)
z = (one << (b - one)) + (splitmix(N, 22) & ((one << (b - one)) - one))
model("cfg 3 (iv): u64 Zipf-like (40 magnitudes)", z, 64)
# for comparison: uniform u32
model("uniform u32", splitmix(N, 23) & np.uint64(0xFFFFFFFF), 32)
