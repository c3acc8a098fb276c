#!/usr/bin/env python3
"""bench.py -- headline benchmark: Gkeys/s of the MI355X-native LSD radix sort.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full sort (histogram + plan + every scatter pass) of one batch
of synthetic keys that is already resident in HBM when the timed region starts.

  N = 1   BASELINE.json configs[1]: 2^28 uniform-random u32 keys (splitmix64,
          seed 1 for the first batch), 4 x 8-bit LSD passes, keys only.
  N > 1   configs[4]: each rank holds 2^29 u32 keys (N = 8 -> 2^32 keys in all),
          one MSD-digit split pass + RCCL all-to-all-v (in sub-ranges, overlapped
          with the local LSD sorts of the sub-ranges already received) per step
          (radix_sorting_amd/multi.py); launched by torch.distributed.run, one
          rank per GPU.  Weak scaling: per-GPU work is fixed as N grows.

Every step sorts a different, still-unsorted batch (K + W batches are generated
on the device up front: at 1 GiB each they fit easily in 288 GB), so no step
takes the reference's pre-sorted early exit and nothing is copied inside the
timed region.

The one JSON line printed by rank 0 carries, besides the contract's fields,
  roofline      the scatter kernel (dominant: P of the 2P+1 array sweeps): algorithmic
                bytes per launch = n * 2 * sizeof(key) over the kernel's
                average duration measured with HIP events on the launch stream inside
                the timed region (rsx_profile_begin/end), against the 8 TB/s HBM3E peak;
  cpu_baseline  the real reference (oracle/_ref/libref.so, kind "reference") or, when
                that is absent, the C restatement (kind "port"), timed on one host core
                on a bounded sample of the same key stream.  N = 1, rank 0 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s HBM3E spec peak


def load_baseline_metric():
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "Gkeys/s sorting 2^28 u32 (1 GPU) + HBM GB/s vs roofline; 1/2/4/8-GPU scaling"


def pmc_traffic_per_launch():
    """HBM bytes per scatter launch from the committed rocprofv3 --pmc passes, if any (profiles/pmc_scatter.json)."""
    path = os.path.join(ROOT, "profiles", "pmc_scatter.json")
    try:
        with open(path) as f:
            return json.load(f)["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(sample_log2=27, reps=3):
    """Single-core CPU sort of the first 2^sample_log2 keys of the N=1 workload's first batch (default: half of it; 3 sorts,
    a few seconds of host time on the GPU box, about 20 s on a slow host)."""
    import numpy as np
    import oracle_lib as ol
    n = 1 << sample_log2
    keys = ol.splitmix_fill(n, ol.U32, 1)
    ref = ol.ref()
    kind = "reference" if ref is not None else "port"
    times = []
    for _ in range(reps):
        src = keys.copy()
        aux = np.empty_like(src)
        aux.fill(0)                       # pre-faulted
        t0 = time.perf_counter()
        if ref is not None:
            ref.ref_sort(ol.ptr(src), ol.ptr(aux), n, ol.U32, 0)
        else:
            ol.oracle().rso_sort(ol.ptr(src), ol.ptr(aux), n, ol.U32, 0, None)
        times.append(time.perf_counter() - t0)
    assert np.all(src[:-1] <= src[1:])
    best = sorted(times)[len(times) // 2]
    return {"value": n / best / 1e9, "unit": "Gkeys/s", "cores": 1, "kind": kind,
            "sample": "first 2^%d keys of batch 0 (splitmix64 seed 1, u32), median of %d fresh-copy sorts, "
                      "%.0f ms each, 1 thread of %d host cores" % (sample_log2, reps, best * 1e3, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=None, help="keys per GPU (default 28 at N=1, 29 at N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-exchange", action="store_true",
                    help="test switch: take the N>1 code path (process group, partition, all-to-all-v) whatever N is")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import radix_sorting_amd as rsa
    from radix_sorting_amd import multi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    rsa.require_gpu()
    sharded = world > 1 or args.force_exchange
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    K, W = args.steps, args.warmup
    log2n = args.log2n if args.log2n is not None else (29 if sharded else 28)
    n = 1 << log2n
    dev = torch.device("cuda", local_rank)
    nbatches = K + W
    # distinct unsorted batch per step; batch b of rank r continues the splitmix stream of seed 1 + b
    # at global index r * n, so the N-rank job sorts exactly the sequence one rank would generate
    batches = []
    for b in range(nbatches):
        t = torch.empty(n, dtype=torch.int32, device=dev)
        rsa.fill_splitmix(t, seed=1 + b, first_index=rank * n)
        batches.append(t)
    engine = multi.HipEngine(rsa.U32)
    cap = n + n // 4 if sharded else n
    scratch = {"aux": torch.empty(cap, dtype=torch.int32, device=dev)}
    if sharded:
        scratch["part"] = torch.empty(n, dtype=torch.int32, device=dev)
        scratch["recv"] = torch.empty(cap, dtype=torch.int32, device=dev)

    def step(i):
        if not sharded:
            return rsa.radix_sort(batches[i], scratch["aux"], dtype=rsa.U32)
        return multi.distributed_sort(batches[i], engine, scratch=scratch, force_exchange=args.force_exchange)

    def fence():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(W):
        step(i)
    fence()
    rsa.profile_begin()
    t0 = time.perf_counter()
    last = None
    for i in range(W, W + K):
        last = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    prof = rsa.profile_end()

    if sharded:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # sanity: the last step's output is sorted (unsigned order == signed order of bits ^ 0x80000000)
    res = last[0]
    flipped = res ^ torch.tensor(-2 ** 31, dtype=torch.int32, device=dev)
    ok = bool((flipped[1:] >= flipped[:-1]).all().item()) if res.numel() > 1 else True
    if sharded:
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())
    if not ok:
        raise SystemExit("bench: output of the last step is not sorted")

    if sharded:
        # RCCL writes its version banner to the C stdout (NCCL_DEBUG=VERSION on the GPU boxes): push it out on every rank
        # before rank 0 prints, so that the JSON line is the last line of the job's output
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        dist.barrier()
        torch.cuda.synchronize()
    if rank == 0:
        total_keys = float(K) * n * world
        launches = max(int(prof.scatter_launches), 1)
        avg_ms = prof.scatter_ms / launches
        bytes_per_launch = prof.scatter_bytes / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": load_baseline_metric(),
            "value": total_keys / elapsed / 1e9,
            "unit": "Gkeys/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": ("2^%d uniform-random u32 keys, 4x8-bit LSD passes, keys only (BASELINE.json configs[1])" % log2n)
                if not sharded else
                ("%d x 2^%d u32 keys sharded by MSD digit, RCCL all-to-all-v, local LSD (BASELINE.json configs[4])"
                 % (world, log2n)),
                "keys_per_gpu": n, "total_keys": n * world, "generator": "splitmix64 seed 1+batch",
                "parallelism": "msd%d" % world if sharded else "1 gpu", "output_sorted": ok,
            },
            "roofline": {
                "kernel": "rsx_scatter2_kernel<u32,NoVal,u32>",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic_per_launch(),
                "bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms, "launches": int(prof.scatter_launches),
            },
            "kernels": {
                "scatter_ms_per_step": prof.scatter_ms / K,
                "histogram_ms_per_step": prof.hist_ms / K,
                "histogram_GBps": (prof.hist_bytes / max(prof.hist_ms, 1e-9) / 1e6) if prof.hist_ms > 0 else None,
                "sort_algorithmic_GBps": None if sharded else total_keys / world * 36 / elapsed / 1e9,
            },
        }
        if world == 1 and not sharded and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
