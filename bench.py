#!/usr/bin/env python3
"""bench.py -- headline benchmark: Gkeys/s of the MI355X-native LSD radix sort.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full sort of one batch of synthetic keys that is already resident
in HBM when the timed region starts -- whatever route the library takes for it
(`config.passes` names it from rsx_info.hybrid).  For the N = 1 workload that is,
since round 3: a sample kernel that proves the input unsorted and every column kept,
two MSB scatter passes into per-bucket slots (the second writes two bytes per key)
and the leaves; the same run then times the reference's own loop --
histogram + one pass per kept column, radix_sort.hpp:82-90 -- for comparison
(`kernels.lsd_only_*`, never `value`).

  N = 1   BASELINE.json configs[1]: 2^28 uniform-random u32 keys (splitmix64,
          seed 1 for the first batch), 4 kept 8-bit columns, keys only.
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
  roofline      the kernel with the LARGEST time per step (the library times its kernels by
                class with HIP events on the launch stream inside the timed region,
                rsx_profile_begin/end): algorithmic bytes per launch (what that
                launch must read and write) over its average duration, against the 8 TB/s
                HBM3E peak; `per_kernel` lists every class that ran the same way, `whole_sort`
                the sum of their algorithmic bytes over the step time; `traffic` = the
                dominant kernel's HBM bytes per launch from the committed rocprofv3 --pmc
                passes (profiles/pmc_kernels.json, which names the commit it was measured at);
  cpu_baseline  the real reference (oracle/_ref/libref*.so, kind "reference") or, when
                that is absent, the C restatement (kind "port"), timed on ONE pinned host core
                on the whole 2^28-key batch 0 (median of 5) and on 40 M keys (single shot),
                CPU model and core count beside it.  N = 1, rank 0 only.

`python bench.py --gpus N` (N > 1) without a launcher starts the ranks itself
(launch_ranks): the parent never touches the GPU, runs
`python -m torch.distributed.run --nproc-per-node N ... bench.py ...` as a child and
relays rank 0's JSON line as the last line of its own stdout.
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


def pmc_traffic_per_launch(kernel_marks):
    """HBM bytes per launch of the kernel whose profiled name contains every string of `kernel_marks`, from the committed
    rocprofv3 --pmc passes, if any (profiles/pmc_kernels.json), and the commit those counters were measured at (PMC counters
    need rocprofv3 around the process: not a per-run reading)."""
    path = os.path.join(ROOT, "profiles", "pmc_kernels.json")
    try:
        with open(path) as f:
            d = json.load(f)
        for name, row in d["kernels"].items():
            if all(m in name for m in kernel_marks):
                return row["hbm_bytes_per_launch"], d.get("measured_at_commit")
    except Exception:
        pass
    return None, None


# The classes the library's profile separates (rsx_profile: ProfScope kinds 0 .. 3), per route of the N = 1 workload: the
# kernel behind each and the marks its rocprofv3 name carries (profiles/pmc_kernels.json)
def kernel_classes(how):
    two_level = how in (2, 4, 5)
    return {
        "hist": ("rsx_hist_kernel<u32> (all columns' counts + the pre-sorted test, radix_sort.hpp:47-58)", ["rsx_hist_kernel<u32"]),
        "scatter": (("rsx_pass32a_kernel<u32> (level-1 pass: whole keys into 256 slots, whole 64-byte atoms)", ["rsx_pass32a_kernel<u32"])
                    if how == 5 else
                    ("rsx_scatter2_kernel<u32,NoVal,u32> (one stable pass by an 8-bit column, radix_sort.hpp:82-90)", ["rsx_scatter2_kernel<u32, NoVal", "u32, false>"])),
        "narrow": ("rsx_pass16a_kernel<u32> (level-2 pass: two bytes per key into 65536 slots, whole 64-byte atoms)",
                   ["rsx_pass16a_kernel<u32"]),
        "leaf": (("rsx_leaf16_kernel<u32,Leaf16Cfg<256,5120,8,12>> (two-byte slots in, sorted keys out; + the list launch of rsx_leaf_sort_kernel)",
                  ["rsx_leaf16_kernel<u32"]) if how == 5 else
                 ("rsx_leaf_sort_kernel<u32> (%s)" % ("65536 buckets" if two_level else "256 buckets"), ["rsx_leaf_sort_kernel<u32"])),
    }


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def _pick_reference_build():
    """oracle/_ref/libref_native.so (-O3 -march=native as the reference's Makefile:1, built in the build container) when this
    host's CPU can run it -- tried in a child process, an illegal instruction must not take bench.py down -- else
    oracle/_ref/libref.so (-march=x86-64-v3), else None (the C restatement is timed, kind "port")."""
    import subprocess
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    probe = ("import ctypes, sys; import numpy as np; l = ctypes.CDLL(sys.argv[1]); a = np.arange(70000, 0, -1, dtype=np.uint32); "
             "b = np.zeros_like(a); l.ref_sort.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]; "
             "r = l.ref_sort(a.ctypes.data, b.ctypes.data, a.size, 2, 0); res = b if r else a; assert res[0] == 1 and res[-1] == 70000")
    for name, flags in (("libref_native.so", "-O3 -march=native (build container's ISA)"), ("libref.so", "-O3 -march=x86-64-v3")):
        path = os.path.join(ref_dir, name)
        if not os.path.exists(path):
            continue
        try:
            if subprocess.run([sys.executable, "-c", probe, path], timeout=120).returncode == 0:
                return path, flags
        except Exception:
            pass
    return None, None


def cpu_baseline(log2n=28, reps=5):
    """SURVEY.md 8d "CPU baseline beside it": the reference's radix_sort (the real headers compiled in place,
    oracle/_ref) on ONE pinned host core, the full 2^28-key batch 0 of the N = 1 workload, median of `reps` fresh-copy runs,
    plus the single-shot 40 M-key sort of configs[0]; CPU model and core count reported.  About 10-15 s of host time."""
    import ctypes as C
    import numpy as np
    import oracle_lib as ol
    pinned = None
    try:
        cores = sorted(os.sched_getaffinity(0))
        pinned = cores[len(cores) // 2]
        os.sched_setaffinity(0, {pinned})
    except (AttributeError, OSError):
        cores = list(range(os.cpu_count() or 1))
    try:
        path, flags = _pick_reference_build()
        if path is not None:
            lib = C.CDLL(path)
            lib.ref_sort.restype = C.c_int
            lib.ref_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
            kind = "reference"

            def sort(src, aux):
                return lib.ref_sort(ol.ptr(src), ol.ptr(aux), src.size, ol.U32, 0)
        else:
            kind, flags = "port", "oracle/rs_oracle.c -O3 -march=x86-64-v3"

            def sort(src, aux):
                return ol.oracle().rso_sort(ol.ptr(src), ol.ptr(aux), src.size, ol.U32, 0, None)

        def timed(keys, repeat):
            times = []
            for _ in range(repeat):
                src = keys.copy()
                aux = np.zeros_like(src)              # pre-faulted
                t0 = time.perf_counter()
                r = sort(src, aux)
                times.append(time.perf_counter() - t0)
            res = aux if r else src
            assert np.all(res[:-1] <= res[1:])
            return sorted(times)[len(times) // 2]

        n = 1 << log2n
        t_full = timed(ol.splitmix_fill(n, ol.U32, 1), reps)
        n40 = 40000000
        t_40m = timed(ol.splitmix_fill(n40, ol.U32, 40), 1)
    finally:
        if pinned is not None:
            os.sched_setaffinity(0, set(cores))
    return {"value": n / t_full / 1e9, "unit": "Gkeys/s", "cores": 1, "kind": kind,
            "cpu": cpu_model(), "host_cores": os.cpu_count(), "build": flags,
            "ms_2p%d" % log2n: t_full * 1e3, "ms_40M_single_shot": t_40m * 1e3, "Gkeys_per_s_40M": n40 / t_40m / 1e9,
            "sample": "the whole batch 0 of the workload (2^%d u32 keys, splitmix64 seed 1): median of %d fresh-copy sorts, "
                      "%.0f ms each; 40 M keys (configs[0], splitmix64 seed 40) single shot %.0f ms; one thread pinned to "
                      "core %s of %d (%s)" % (log2n, reps, t_full * 1e3, t_40m * 1e3, pinned, os.cpu_count(), cpu_model())}


def launch_ranks(args_list, gpus, script=None):
    """`python bench.py --gpus N` without a launcher: this process stays free of the GPU, starts the N ranks as
    `python -m torch.distributed.run ... bench.py <same arguments>`, relays their output and ends with rank 0's JSON line
    as its own last line of stdout and with their exit status.

    The result does not travel through the pipe all ranks write to (a 5 KB line is several writes; another rank's text can
    land inside it): rank 0 writes it to the file named by RSX_BENCH_RESULT, and the parent prints that file after the
    ranks have exited.  Lines of the relayed output that look like a result line are dropped, so nothing glued to one
    can reach the parent's stdout either."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    fd, result_path = tempfile.mkstemp(prefix="rsx_bench_", suffix=".json")
    os.close(fd)
    env["RSX_BENCH_RESULT"] = result_path
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + args_list
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
        for line in proc.stdout:
            text = line.rstrip("\n")
            if '"metric"' in text and "{" in text:
                continue                         # a rank's copy of the result (possibly with another rank's text glued on)
            print(text, flush=True)
        rc = proc.wait()
        with open(result_path) as f:
            line_json = f.read().strip()
    finally:
        try:
            os.unlink(result_path)
        except OSError:
            pass
    if line_json:
        json.loads(line_json)                    # (a torn file would be a bug here: fail loudly rather than print it)
        sys.stdout.write(line_json + "\n")
        sys.stdout.flush()
    elif rc == 0:
        rc = 1
        print("bench: the ranks wrote no result line", file=sys.stderr)
    raise SystemExit(rc)


def emit_result(out, rank):
    """Rank 0's one JSON line: to the file the launching parent named (RSX_BENCH_RESULT, written whole and renamed), and
    to stdout as ONE write after everything else of this process has been flushed."""
    if rank != 0:
        return
    text = json.dumps(out)
    path = os.environ.get("RSX_BENCH_RESULT")
    if path:
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            f.write(text + "\n")
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    data = (text + "\n").encode()
    while data:                                   # (a pipe takes 64 KiB at once; the loop is for a short write)
        data = data[os.write(1, data):]


def quiet_this_rank():
    """A rank other than 0 has nothing more to say once its checks are through: what its teardown (RCCL, the runtime)
    still prints must not land inside or behind rank 0's result line on the shared stdout / stderr."""
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    sys.stderr.flush()
    null = os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 1)
    os.dup2(null, 2)
    os.close(null)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=None, help="keys per GPU (default 28 at N=1, 29 at N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-exchange", action="store_true",
                    help="test switch: take the N>1 code path (process group, MSD split, all-to-all-v) whatever N is")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(sys.argv[1:], args.gpus)     # (before anything touches the GPU; does not return)

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
    engine.host_marks = {}
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

    def checksums(t):
        """order-independent sums of a batch (wrap-around int64): the keys' sum and the sum of key * (key >> 7)"""
        return torch.stack([t.sum(dtype=torch.int64), (t * (t >> 7)).sum(dtype=torch.int64)])

    # (the last step's input, summed before anything sorts it: its output must be the same keys -- a sorted array of OTHER keys
    # passes a sortedness test, and round 5's one-rank forced exchange of 2^31 bytes delivered just that)
    sums_in = checksums(batches[W + K - 1])
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
    sums_out = checksums(res)
    if sharded:
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())
        dist.all_reduce(sums_in, op=dist.ReduceOp.SUM)      # (int64, wrap-around: the ranks' keys move between ranks)
        dist.all_reduce(sums_out, op=dist.ReduceOp.SUM)
    if not ok:
        raise SystemExit("bench: output of the last step is not sorted")
    if not bool((sums_in == sums_out).all().item()):
        raise SystemExit("bench: the output of the last step is sorted but does not hold the input's keys (checksums differ)")
    # per-phase GPU time of the LAST step, max over ranks (split / exchange / sort; the chunk pipeline overlaps the last two):
    # a first run on several GPUs should explain itself
    phase_max = None
    if sharded and isinstance(last[1], dict) and callable(last[1].get("phases")):
        ph = last[1]["phases"]() or {}
        names = ["split_ms", "exchange_ms", "sort_ms", "exchange_and_sort_ms", "total_ms"]
        vec = torch.tensor([float(ph.get(k, -1.0)) for k in names], dtype=torch.float64, device=dev)
        dist.all_reduce(vec, op=dist.ReduceOp.MAX)
        phase_max = {k: float(v) for k, v in zip(names, vec.tolist()) if v >= 0}

    if sharded:
        # RCCL writes its version banner to the C stdout (NCCL_DEBUG=VERSION on the GPU boxes): push it out on every rank
        # before rank 0 prints, so that the JSON line is the last line of the job's output
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        dist.barrier()
        torch.cuda.synchronize()
    # how the last step went through its columns (rsx_info.hybrid): 0 one pass per kept column, 1 / 2 one / two MSB passes
    # and LDS leaves (README.md:647-650), 3 MSB pass + LSB passes inside its buckets
    how = int(last[1].hybrid) if (not sharded and hasattr(last[1], "hybrid")) else 0
    lsd_only = None
    if not sharded and how != 0 and rank == 0:
        # the same workload with one pass per kept column (RSX_NO_HYBRID=1), a few steps, for comparison; not `value`
        os.environ["RSX_NO_HYBRID"] = "1"
        rsa.reload_env()
        reps = min(K, 5)
        for b in range(1 + reps):      # (the timed steps have sorted their batches in place: fresh ones)
            rsa.fill_splitmix(batches[b], seed=1001 + b, first_index=rank * n)
        step(0)
        fence()
        t1 = time.perf_counter()
        for i in range(1, 1 + reps):
            step(i)
        fence()
        lsd_only = (time.perf_counter() - t1) / reps
        del os.environ["RSX_NO_HYBRID"]
        rsa.reload_env()
    out = None
    if rank == 0:
        total_keys = float(K) * n * world
        classes = kernel_classes(how)
        per_kernel = {}
        for cls, (ms, launches, nbytes) in (("hist", (prof.hist_ms, prof.hist_launches, prof.hist_bytes)),
                                            ("scatter", (prof.scatter_ms, prof.scatter_launches, prof.scatter_bytes)),
                                            ("narrow", (prof.narrow_ms, prof.narrow_launches, prof.narrow_bytes)),
                                            ("leaf", (prof.leaf_ms, prof.leaf_launches, prof.leaf_bytes))):
            if not launches or ms <= 0:
                continue
            # (a class may be several launches per step -- four passes, or the leaves and their list launch: what is compared
            # with the peak is the class's bytes over the class's time, i.e. the average over its launches)
            gbps = nbytes / (ms * 1e-3) / 1e9
            per_kernel[cls] = {"kernel": classes[cls][0], "ms_per_step": ms / K, "launches_per_step": launches / K,
                               "bytes_per_launch": nbytes / launches, "avg_launch_ms": ms / launches,
                               "achieved": gbps, "frac": gbps / HBM_PEAK_GBS}
        if not per_kernel:   # (nothing was timed: cannot happen on a working library, but the line must still print)
            per_kernel["scatter"] = {"kernel": classes["scatter"][0], "ms_per_step": 0.0, "launches_per_step": 0.0, "bytes_per_launch": 0.0,
                                     "avg_launch_ms": 0.0, "achieved": 0.0, "frac": 0.0}
        dominant = max(per_kernel, key=lambda c: per_kernel[c]["ms_per_step"])
        dom = per_kernel[dominant]
        traffic, traffic_commit = pmc_traffic_per_launch(classes[dominant][1])
        all_bytes = prof.hist_bytes + prof.scatter_bytes + prof.leaf_bytes + prof.narrow_bytes
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
                "workload": ("2^%d uniform-random u32 keys, 4 kept 8-bit columns, keys only (BASELINE.json configs[1])" % log2n)
                if not sharded else
                ("%d x 2^%d u32 keys sharded by MSD digit, RCCL all-to-all-v, local LSD (BASELINE.json configs[4])"
                 % (world, log2n)),
                "keys_per_gpu": n, "total_keys": n * world, "generator": "splitmix64 seed 1+batch",
                "parallelism": "msd%d" % world if sharded else "1 gpu", "output_sorted": ok, "output_checksums_match_input": True,
                "passes": {0: "one scatter pass per kept column, LSB first (radix_sort.hpp:82-90)",
                           1: "one MSB scatter pass, then the other kept columns per bucket in LDS (README.md:647-650)",
                           2: "two MSB scatter passes, then the other kept columns per bucket in LDS (README.md:647-650)",
                           3: "one MSB scatter pass, then one pass per remaining column inside its buckets",
                           4: "two MSB scatter passes (the second into per-bucket slots, no second count), then the other kept "
                              "columns per bucket in LDS (README.md:647-650)",
                           5: "no histogram (a sample proves all columns kept and the input unsorted): two MSB scatter passes, both "
                              "into per-bucket slots with the bucket sizes read off the look-back chains (the second writes only "
                              "the two bytes the leaves sort by), then the other kept columns per bucket in LDS (README.md:647-650)"}[how],
            },
            "roofline": {
                "kernel": dom["kernel"], "kernel_class": dominant,
                "dominant_by": "largest time per step among the kernel classes of the step (HIP events): %.3f of %.3f ms" % (dom["ms_per_step"], elapsed / K * 1e3),
                "bound": "hbm", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["frac"],
                "traffic": traffic, "traffic_measured_at_commit": traffic_commit,
                "bytes_per_launch": dom["bytes_per_launch"], "avg_launch_ms": dom["avg_launch_ms"],
                "launches": int(round(dom["launches_per_step"] * K)),
                "per_kernel": per_kernel,
                # every kernel's algorithmic bytes of a step over the whole step (launch gaps and the small kernels included)
                "whole_sort": None if sharded else {"bytes_per_key": all_bytes / total_keys, "achieved": all_bytes / elapsed / 1e9,
                                                    "frac": all_bytes / elapsed / 1e9 / HBM_PEAK_GBS},
            },
            "kernels": {
                "scatter_ms_per_step": prof.scatter_ms / K,
                "histogram_ms_per_step": prof.hist_ms / K,
                "histogram_GBps": (prof.hist_bytes / max(prof.hist_ms, 1e-9) / 1e6) if prof.hist_ms > 0 else None,
                "leaf_ms_per_step": prof.leaf_ms / K,
                "leaf_GBps": (prof.leaf_bytes / max(prof.leaf_ms, 1e-9) / 1e6) if prof.leaf_ms > 0 else None,
                # the level-2 pass of a sort without a histogram writes two bytes per 4-byte key (another instantiation of the
                # pass kernel, 6 algorithmic bytes per key; `roofline` above is the full-width one's)
                "narrow_pass_ms_per_step": prof.narrow_ms / K,
                "narrow_pass_GBps": (prof.narrow_bytes / max(prof.narrow_ms, 1e-9) / 1e6) if prof.narrow_ms > 0 else None,
                # algorithmic bytes per key of the whole sort, summed over the kernels that ran: 4 (histogram) +
                # 4 passes x 8 = 36 with one pass per kept column; two MSB passes + leaves: 4 + 2 x 8 + 8 = 28 (+ 4 when the
                # second pass needs per-bucket counts first)
                "sort_algorithmic_bytes_per_key": None if sharded else all_bytes / total_keys,
                "sort_algorithmic_GBps": None if sharded else all_bytes / elapsed / 1e9,
                "lsd_only_ms_per_step": None if lsd_only is None else lsd_only * 1e3,
                "lsd_only_Gkeys_per_s": None if lsd_only is None else n / lsd_only / 1e9,
            },
        }
        if sharded and isinstance(last[1], dict) and "host_ms_submit_exchange_and_sorts" in last[1]:
            # the host's share of a step (multi.py is Python): how long rank 0 took to get the counts back (includes the split
            # pass on the device) and to submit the exchange and the local sorts (nothing of it waits for the device)
            out["host"] = {"ms_split_and_counts_last_step": last[1]["host_ms_split_and_counts"],
                           "ms_submit_exchange_and_sorts_last_step": last[1]["host_ms_submit_exchange_and_sorts"],
                           "chunks": last[1].get("chunks"), "heavy_digits": last[1].get("heavy_digits"),
                           "split_and_counts_parts_ms": dict(engine.host_marks)}
        if sharded and isinstance(last[1], dict):
            out["multi"] = {"phases_last_step_max_over_ranks_ms": phase_max, "safe_mode": bool(last[1].get("safe_mode")),
                            "safe_why": last[1].get("safe_why"), "overlap_stream": last[1].get("overlap_stream"),
                            "imbalance": last[1].get("imbalance")}
        if world == 1 and not sharded and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if sharded:
        # the result line is the LAST thing the job prints: the other ranks go quiet, the process group is torn down
        # (RCCL's own chatter comes out here), and only then does rank 0 write its line
        if rank != 0:
            quiet_this_rank()
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
    emit_result(out if rank == 0 else None, rank)


if __name__ == "__main__":
    main()
