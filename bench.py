#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Stenos block codec.

Metric (BASELINE.json): encode+decode GB/s of input bytes at level 1 on int32, with the compression
ratio.  One "step" = one level-1 encode pass + one decode pass over the synthetic typed array that is
already resident in HBM (device-pointer entry points of libstenos.so; nothing crosses PCIe except the
8-byte frame size).  The decoder is handed the frame and nothing else (no index from the encoder: the superblock
chain is walked on the device inside the timed region).  value = input bytes of all ranks / (encode + decode time).

Workload at N=1: BASELINE.json configs[1], 8 GiB int32 (2^31 elements, bytesoftype 4, level 1 block
codec only), variant (b) of SURVEY.md section 8d: uniform 12-bit values `u & 0xFFF` from splitmix64 seed 42
-- the variant in which the codec does real work (ratio ~2.52).  The literal full-entropy variant (a),
where every superblock falls back to COPY, is measured too and reported under "full_entropy" with a roofline of
its own.  --config int16 / double run one shard of configs[3] / configs[2] at level 1 the same way.
With --gpus N every rank owns its own 8 GiB superblock range of an 8*N GiB array (weak scaling, no
data-path collective in the timed region); the gather of the compressed segments to rank 0 and the sharded
decode are run and timed afterwards ("sharded_exchange").
Outside the timed region at N=1: the reference's CPU path on a 1 GiB prefix ("cpu_baseline"), a byte-for-byte
comparison of the WHOLE GPU frame with the reference's frames of its 1 GiB slices ("parity_bytes"), the host-pointer
ABI end to end ("host_pointer"), the other configurations of BASELINE.json ("other_configs"), and the rate of the
runtime's own device-to-device copy on the box ("roofline.device_copy": context for the fractions of the nominal peak).

Launch: python bench.py [--gpus N --steps K --warmup W].  For N > 1 either through torch.distributed.run (one rank per GPU,
RANK / WORLD_SIZE in the environment) or plainly: without RANK the script starts the N ranks itself as a child process and
relays rank 0's line.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--gib", type=float, default=8.0, help="input GiB per GPU (default: the 8 GiB of configs[1])")
    p.add_argument("--config", choices=("int32", "int16", "double"), default="int32", help="BASELINE.json config: int32 = configs[1] (the headline), "
                   "int16 = one shard of configs[3], double = configs[2] at level 1")
    p.add_argument("--kind", default=None, help="another datagen kind for the chosen element size (experiments)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-full-entropy", action="store_true", help="skip the extra full-entropy (all COPY) measurement")
    p.add_argument("--no-other-configs", action="store_true", help="skip the int16 / double / level 2 / level 3 entries")
    p.add_argument("--cpu-sample-mib", type=int, default=1024)
    return p.parse_args()


def cpu_share():
    """CPUs granted by the container's cgroup quota (cpu.max), or None."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(float(q) / float(p) + 0.5))
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            return max(1, int(q / p + 0.5)) if q > 0 else None
        except Exception:
            return None


def cpu_baseline(sample, T, sample_desc, level=1, reps=5):
    """The reference's own CPU path (oracle/_ref, unmodified sources, AVX2/BMI2 build) timed on this
    host's cores on a bounded prefix of the same workload, at the same level; falls back to the scalar C oracle ("port")."""
    import numpy as np

    from _libs import load_oracle, load_ref, np_ptr

    cores = os.cpu_count() or 1
    share = cpu_share()  # CPUs a container's quota really grants (None: no quota)
    ref = load_ref(det=False)
    nb = sample.nbytes
    if ref is not None:
        out = np.zeros(ref.stenos_bound(nb), dtype=np.uint8)
        back = np.zeros(nb, dtype=np.uint8)
        res = {}
        for threads in sorted({1, cores} | ({min(cores, share + share // 2)} if share else set())):
            ctx = ref.stenos_make_context()
            ref.stenos_set_level(ctx, level)
            ref.stenos_set_threads(ctx, threads)
            best_e = best_d = 1e30
            r = 0
            for _ in range(reps if threads > 1 else max(1, reps // 2)):  # best of 5, as benchs/bench_to_csv.cpp:113-126
                t = time.perf_counter()
                r = ref.stenos_compress_generic(ctx, np_ptr(sample), T, nb, np_ptr(out), out.nbytes)
                best_e = min(best_e, time.perf_counter() - t)
                t = time.perf_counter()
                d = ref.stenos_decompress_generic(ctx, np_ptr(out), T, r, np_ptr(back), nb)
                best_d = min(best_d, time.perf_counter() - t)
                assert d == nb
            ref.stenos_destroy_context(ctx)
            res[threads] = (nb / (best_e + best_d) / 1e9, nb / best_e / 1e9, nb / best_d / 1e9, nb / r)
        use = max(res, key=lambda k: res[k][0])
        return {"value": round(res[use][0], 3), "unit": "GB/s", "cores": use, "kind": "reference", "sample": sample_desc,
                "encode_gbps": round(res[use][1], 3), "decode_gbps": round(res[use][2], 3), "ratio": round(res[use][3], 4),
                "single_thread_value": round(res[1][0], 3), "host_cores": cores, "cpu_quota": share,
                "threads_tried": {str(k): round(v[0], 3) for k, v in res.items()}}
    lib = load_oracle()
    small = sample[: 64 << 20]
    nb = small.nbytes
    out = np.zeros(lib.so_bound(nb), dtype=np.uint8)
    back = np.zeros(nb, dtype=np.uint8)
    t = time.perf_counter()
    r = lib.so_compress(np_ptr(small), T, nb, np_ptr(out), out.nbytes, level)
    te = time.perf_counter() - t
    t = time.perf_counter()
    lib.so_decompress(np_ptr(out), T, r, np_ptr(back), nb, 1)
    td = time.perf_counter() - t
    return {"value": round(nb / (te + td) / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port", "sample": "first 64 MiB of the workload",
            "encode_gbps": round(nb / te / 1e9, 3), "decode_gbps": round(nb / td / 1e9, 3), "ratio": round(nb / r, 4)}


def run_workload(st, torch, src, T, steps, warmup, dist, world, blocking=False):
    """Returns wall seconds for `steps` round trips (max over ranks), per-direction seconds, kernel ms, csize."""
    nbytes = src.numel()
    dst = torch.empty(st.bound(nbytes), dtype=torch.uint8, device=src.device)
    back = torch.empty_like(src)
    csize = 0

    def enc():
        if blocking:  # levels >= 2: the host finishes the frame (zstd) inside the call
            return st.compress(src, T, dst)
        st.compress(src, T, dst, wait=False)
        return st.finish()  # the frame size is a host value the decoder needs

    def dec(csize, indexed=False):
        # The decoder gets the FRAME and nothing else: the superblock chain is found on the device (csrc/walk.h) inside the
        # timed region.  indexed=True (an extra leg behind the timed region) hands over the index the encoder left behind,
        # which a decoder in the same process could use.
        idx, nsb = st.last_index() if indexed else (None, 0)
        if blocking:
            st.decompress(dst, T, csize, back, index_ptr=idx if nsb else None)
        else:
            st.decompress(dst, T, csize, back, index_ptr=idx if nsb else None, wait=False)
            st.finish()

    def step():
        nonlocal csize
        csize = enc()
        dec(csize)

    for _ in range(warmup):
        step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    enc_s = dec_s = 0.0
    kenc = kdec = 0.0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ev[0].record()
        csize = enc()
        ev[1].record()
        dec(csize)
        ev[2].record()
        ev[2].synchronize()
        enc_s += ev[0].elapsed_time(ev[1]) / 1e3
        dec_s += ev[1].elapsed_time(ev[2]) / 1e3
        kenc += st.kernel_ms(0)
        kdec += st.kernel_ms(1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=src.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ok = bool(torch.equal(back, src))
    # extra leg, outside the timed region: the same decode with the encoder's index handed over (no walk)
    back.zero_()
    ev[0].record()
    for _ in range(steps):
        dec(csize, indexed=True)
    ev[1].record()
    ev[1].synchronize()
    dec_indexed_s = ev[0].elapsed_time(ev[1]) / 1e3
    ok = ok and bool(torch.equal(back, src))
    return dict(wall=wall, enc_s=enc_s, dec_s=dec_s, dec_indexed_s=dec_indexed_s, kenc_ms=kenc / max(steps, 1), kdec_ms=kdec / max(steps, 1), csize=csize, ok=ok)


CONFIGS = {
    # name: (datagen kind, bytesoftype, description)
    "int32": ("rand12", 4, "configs[1]: {gib:g} GiB int32 per GPU, bytesof=4, level 1, uniform 12-bit values (splitmix64 seed 42, u & 0xFFF)"),
    "int16": ("walk", 2, "configs[3] shard: {gib:g} GiB int16 random walk per GPU (x += u % 17 - 8), bytesof=2, level 1; every rank draws a walk of its own "
              "(splitmix64 seed 7 + rank): N independent series of {gib:g} GiB, not one series of N x {gib:g} GiB cut in N"),
    "double": ("sine", 8, "configs[2] at level 1: {gib:g} GiB double sin(i * 0.001) per GPU, bytesof=8"),
}


# The other BASELINE.json configurations, one bench entry each under "other_configs" at N=1 (the driver times them with the
# headline): name -> (datagen kind, bytesoftype, level, GiB, steps, MiB of the CPU sample, description)
OTHER_CONFIGS = {
    "int16_walk_level1": ("walk", 2, 1, 8.0, 3, 512, "configs[3], one shard: 8 GiB int16 random walk (x += u % 17 - 8, splitmix64 seed 7), bytesof=2, level 1"),
    "double_sine_level1": ("sine", 8, 1, 8.0, 3, 512, "configs[2] at level 1: 8 GiB double sin(i * 0.001), bytesof=8 (the level-2 frame is the same stream)"),
    "double_sine_level2": ("sine", 8, 2, 8.0, 1, 256, "configs[2]: 8 GiB double sin(i * 0.001), bytesof=8, level 2 (block codec on the GPU, LZ4-dry estimate and zstd attempts on the host)"),
    "bytes_smooth_level3": ("smooth8", 1, 3, 8.0, 1, 256, "configs[4]: 8 GiB bytes, bytesof=1, level 3 (256 KiB superblocks; block stream + zstd, code 5), smooth signal 128 + 100 sin(0.01 i) +- 2"),
}


def measure_other(name, torch, dist, dev, with_cpu):
    import numpy as np

    from stenos_amd.api import Stenos
    from stenos_amd.datagen import generate, generate_torch

    kind, T, level, gib, steps, cpu_mib, desc = OTHER_CONFIGS[name]
    n = int(gib * (1 << 30)) // T
    src = generate_torch(kind, T, n, seed={"walk": 7, "smooth8": 9}.get(kind, 42), device=dev)
    st = Stenos(level=level)
    st.set_profiling(True)
    hosted = level >= 2 or T == 1
    if hosted:  # (one untimed call brings the context's staging buffers up; then the stage clocks start)
        warm = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device=dev)
        st.compress(src, T, warm)
        del warm
        st.stage_ms(reset=True)
    r = run_workload(st, torch, src, T, steps, 0 if hosted else 1, dist, 1, blocking=hosted)
    assert r["ok"], name
    nb = src.numel()
    e = {"workload": desc, "bytesoftype": T, "level": level, "value": round(nb * steps / r["wall"] / 1e9, 3), "unit": "GB/s", "steps": steps,
         "compression_ratio": round(nb / r["csize"], 4), "encode_gbps": round(nb * steps / r["enc_s"] / 1e9, 3), "decode_gbps": round(nb * steps / r["dec_s"] / 1e9, 3),
         "decode_indexed_gbps": round(nb * steps / r["dec_indexed_s"] / 1e9, 3), "decode_input": "the frame alone (header chain walked on the device inside the timed region)"}
    if level == 1 and T > 1:
        algo = nb + r["csize"]
        key = {"walk": "int16", "sine": "double"}.get(kind) if gib == 8.0 else None
        t_enc, src_enc = profiled_traffic(key)
        t_dec, _ = profiled_traffic(key, "decode_superblocks")
        roof = roofline("encode_superblocks", algo, r["kenc_ms"], t_enc, src_enc)
        roof["decode_superblocks"] = {k: v for k, v in roofline("decode_superblocks", algo, r["kdec_ms"], t_dec).items() if k in ("achieved", "frac", "kernel_ms", "traffic")}
        e["roofline"] = roof
    else:
        e["roofline"] = None
        stages = st.stage_ms(reset=True)
        calls = 2 * steps  # (the timed decode and the indexed decode leg both count into the decode stages)
        e["stage_ms_per_encode"] = {k: round(stages[k] / steps, 2) for k in ("gpu_pass", "estimates", "wait_block_streams", "zstd", "layout", "wait_upload")}
        e["stage_ms_per_decode"] = {k: round(stages[k] / calls, 2) for k in ("inflate", "device_decode")}
        zs = stages["zstd"] / steps
        e["zstd_only_gbps"] = round(nb / (zs / 1e3) / 1e9, 3) if zs > 0 else None
        e["note"] = ("host-bound: one LZ4-dry estimate and up to two zstd calls per superblock run on the host's cores; the transfers of the block streams and of "
                     "the frame run beside zstd (what is left of them shows as wait_*); zstd_only_gbps is the input rate of the zstd stage alone")
    if with_cpu:
        nbytes = min((cpu_mib << 20) + 4000 * T, nb)
        sample = src[:nbytes].cpu().numpy()
        e["cpu_baseline"] = cpu_baseline(sample, T, f"first {cpu_mib} MiB + {4000 * T} B of the same workload, same level, best of 3", level=level, reps=3)
        if level == 1 and T > 1:  # the whole frame against the reference's, slice by slice (outside the timed region)
            e["parity_bytes"] = reference_frame_parity(st, torch, src, T)
            e["parity_frame_bytes"] = int(r["csize"])
    st.close()
    del src
    torch.cuda.empty_cache()
    return e


# Counters cannot be collected inside a bench run: the HBM traffic of a configuration is the figure of the committed profile
# of the same workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, tools/pmc_config.sh + tools/pmc_summary.py,
# with the FETCH x 2 correction of the guide), quoted with its origin.
TRAFFIC_PROFILES = {"int32": "pmc_traffic.json", "int32_rand": "r05_pmc_int32_rand.json", "int16": "r05_pmc_int16.json", "double": "r05_pmc_double.json"}


def profiled_traffic(key, kernel="encode_superblocks"):
    path = os.path.join(ROOT, "profiles", TRAFFIC_PROFILES.get(key, ""))
    if key not in TRAFFIC_PROFILES or not os.path.exists(path):
        return None, None
    try:
        with open(path) as f:
            j = json.load(f)
        t = j.get(f"{kernel}_hbm_bytes_per_launch")
        src = f"profiles/{TRAFFIC_PROFILES[key]} ({j.get('profile', '?')}, build {j.get('build', '?')}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this workload, separate passes"
        return t, (src if t else None)
    except Exception:
        return None, None


def roofline(kernel, algo_bytes, kernel_ms, traffic=None, traffic_source=None):
    achieved = algo_bytes / (kernel_ms / 1e3) / 1e9 if kernel_ms > 0 else 0.0
    return {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": round(kernel_ms, 4)}


def device_copy_rate(torch, dev, nbytes=2 << 30, reps=5):
    """What the runtime's own device-to-device copy reaches on this box (read + written bytes per second): the practical
    ceiling of a read + write stream, for context next to the fractions of the nominal 8 TB/s.  Never part of `value`."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    a.random_(0, 255)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    torch.cuda.empty_cache()
    return {"achieved": round(2 * nbytes / (ms / 1e3) / 1e9, 2), "unit": "GB/s", "frac": round(2 * nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 4),
            "note": f"torch copy_ of {nbytes >> 30} GiB between two device buffers (the runtime's copy kernel), read + written bytes: the practical ceiling of a read + write stream here"}


def reference_frame_parity(st, torch, src, T, slice_bytes=1 << 30):
    """Byte-for-byte check of the WHOLE frame at benchmark scale, outside the timed region: superblocks are coded independently,
    so the reference's frame of a slice of whole superblocks must equal the same superblocks of the GPU's frame of the whole
    array.  The array is taken in slices of 1 GiB (host memory stays bounded; the reference needs about half a second per
    slice with all cores).  Returns the compared frame bytes (the frame minus its 8-byte header), or None without the
    compiled reference."""
    import numpy as np

    from _libs import load_ref, np_ptr

    ref = load_ref(det=False)
    if ref is None:
        return None
    nbytes = src.numel()
    sb = 131072 // (256 * T) * 256 * T
    slice_bytes = max(sb, slice_bytes // sb * sb)
    dst = torch.empty(st.bound(nbytes), dtype=torch.uint8, device=src.device)
    csize = st.compress(src, T, dst)
    idx_ptr, nsb = st.last_index()
    assert idx_ptr and nsb == (nbytes + sb - 1) // sb
    index = torch.empty(nsb + 1, dtype=torch.int64)
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(ctypes.c_void_p(index.data_ptr()), ctypes.c_void_p(idx_ptr), ctypes.c_size_t((nsb + 1) * 8), 2) == 0
    ctx = ref.stenos_make_context()
    ref.stenos_set_level(ctx, 1)
    ref.stenos_set_threads(ctx, os.cpu_count() or 1)
    exp = np.zeros(ref.stenos_bound(slice_bytes), dtype=np.uint8)
    compared = 0
    for begin in range(0, nbytes, slice_bytes):
        n = min(slice_bytes, nbytes - begin)
        sample = src[begin:begin + n].cpu().numpy()
        r = ref.stenos_compress_generic(ctx, np_ptr(sample), T, n, np_ptr(exp), exp.nbytes)
        assert r < (1 << 63), "the reference failed on a slice"
        s0, s1 = begin // sb, (begin + n + sb - 1) // sb
        o0, o1 = int(index[s0]), int(index[s1])
        assert o1 - o0 == r - 8, f"slice at {begin}: {o1 - o0} frame bytes on the GPU, {r - 8} from the reference"
        got = dst[o0:o1].cpu().numpy()
        assert np.array_equal(got, exp[8:r]), f"GPU frame differs from the reference's frame in the slice at {begin}"
        compared += r - 8
    ref.stenos_destroy_context(ctx)
    assert compared == csize - 8
    return int(compared)


def host_pointer_rate(T, sample):
    """The frozen ABI with host pointers (stenos_compress_generic / stenos_decompress_generic): PCIe both ways included.
    Twice: on pageable memory (what a malloc'ing caller gets: staged copies) and on page-locked memory (the kernels read and
    write the caller's buffers through the link, both directions at once)."""
    import numpy as np
    import torch

    from _libs import np_ptr
    from stenos_amd.api import load_library

    lib = load_library()
    nb = sample.nbytes

    def rate(src, out, back):
        ctx = lib.stenos_make_context()
        best_e = best_d = 1e30
        r = 0
        for _ in range(3):
            t = time.perf_counter()
            r = lib.stenos_compress_generic(ctx, np_ptr(src), T, nb, np_ptr(out), out.nbytes)
            best_e = min(best_e, time.perf_counter() - t)
            t = time.perf_counter()
            d = lib.stenos_decompress_generic(ctx, np_ptr(out), T, r, np_ptr(back), nb)
            best_d = min(best_d, time.perf_counter() - t)
            assert d == nb
        lib.stenos_destroy_context(ctx)
        assert np.array_equal(back, src), "host-pointer round trip mismatch"  # (outside the timed loop)
        return round(nb / best_e / 1e9, 2), round(nb / best_d / 1e9, 2), int(r)

    cap = lib.stenos_bound(nb)
    e, d, r = rate(sample, np.zeros(cap, dtype=np.uint8), np.zeros(nb, dtype=np.uint8))
    res = {"encode_gbps": e, "decode_gbps": d, "sample_bytes": nb, "frame_bytes": r,
           "note": "host pointers through the C ABI, pageable memory, PCIe both ways included; never the headline value"}
    try:
        keep = [torch.empty(n, dtype=torch.uint8, pin_memory=True) for n in (nb, cap, nb)]
        src, out, back = (t.numpy() for t in keep)
        src[:] = sample
        e, d, r2 = rate(src, out, back)
        assert r2 == r
        res["pinned"] = {"encode_gbps": e, "decode_gbps": d, "note": "the same calls on page-locked buffers: no staging copy, the kernels go through the link"}
    except RuntimeError as ex:  # no page-locked memory to be had
        res["pinned"] = {"error": repr(ex)[:200]}
    return res


def launch_ranks(args):
    """python bench.py --gpus N without an outer torch.distributed.run: start the N ranks as a CHILD process (this process has
    not touched the GPU -- nothing is imported that would -- and never execs), relay rank 0's JSON line, return its exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:  # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if p.returncode != 0 or line is None:
        sys.stderr.write(p.stdout[-4000:])
        sys.stderr.write(f"\nbench.py: the {args.gpus}-rank child exited with {p.returncode}" + ("" if line else " and printed no result line") + "\n")
        return p.returncode or 1
    print(line, flush=True)
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    import torch.distributed as dist

    # one process per GPU; STENOS_BENCH_ONE_DEVICE=1 (test rigs with a single GPU) puts every rank on cuda:0
    # and uses gloo for the collectives (two ranks cannot share a GPU under RCCL)
    one_device = os.environ.get("STENOS_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if one_device else "nccl", rank=rank, world_size=world)

    from stenos_amd.api import Stenos
    from stenos_amd.datagen import generate_torch

    kind, T, desc = CONFIGS[args.config]
    if args.kind:
        kind, desc = args.kind, "{gib:g} GiB per GPU, bytesof=%d, kind=%s" % (T, args.kind)
    n = int(args.gib * (1 << 30)) // T
    nbytes = n * T
    dev = f"cuda:{local}"
    st = Stenos(level=1)
    st.set_profiling(True)

    if kind == "walk":  # a series of its own per shard
        src = generate_torch(kind, T, n, seed=7 + rank, device=dev)
    else:
        src = generate_torch(kind, T, n, seed=42, device=dev, start=rank * n)
    torch.cuda.synchronize()
    r = run_workload(st, torch, src, T, args.steps, args.warmup, dist, world)
    assert r["ok"], "round trip mismatch"
    total_in = world * nbytes * args.steps
    value = total_in / r["wall"] / 1e9
    ratio = nbytes / r["csize"]

    out = None
    if rank == 0:
        # roofline of the dominant kernel (encode_superblocks, the fused encoder): algorithmic bytes = N read + C written per launch
        algo = nbytes + r["csize"]
        key = {"int32": "int32", "int16": "int16", "double": "double"}[args.config] if not args.kind and args.gib == 8.0 else None
        traffic, source = profiled_traffic(key)
        t_dec, _ = profiled_traffic(key, "decode_superblocks")
        roof = roofline("encode_superblocks", algo, r["kenc_ms"], traffic, source)
        roof["decode_superblocks"] = {k: v for k, v in roofline("decode_superblocks", algo, r["kdec_ms"], t_dec).items() if k in ("achieved", "frac", "kernel_ms", "traffic")}
        if world == 1:
            try:
                roof["device_copy"] = device_copy_rate(torch, dev)
            except Exception as ex:  # (context only: never in the way of the line)
                roof["device_copy"] = {"error": repr(ex)[:200]}
        out = {
            "metric": "encode+decode GB/s (input bytes) at level 1, int32" if T == 4 else f"encode+decode GB/s (input bytes) at level 1, bytesof={T}",
            "value": round(value, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(r["wall"] / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": desc.format(gib=args.gib), "bytesoftype": T, "level": 1, "superblock_bytes": 131072 // (256 * T) * 256 * T,
                       "sharding": f"{world} x contiguous superblock ranges"},
            "compression_ratio": round(ratio, 4),
            "encode_gbps": round(nbytes * args.steps / r["enc_s"] / 1e9, 3),
            "decode_gbps": round(nbytes * args.steps / r["dec_s"] / 1e9, 3),
            "decode_input": "the frame alone: the superblock chain is walked on the device inside the timed region (csrc/walk.h); decode_indexed_gbps is the "
                            "same decode with the encoder's index handed over (a decoder in the encoder's process), measured behind the timed region",
            "decode_indexed_gbps": round(nbytes * args.steps / r["dec_indexed_s"] / 1e9, 3),
            "value_with_indexed_decode": round(nbytes * args.steps / (r["enc_s"] + r["dec_indexed_s"]) / 1e9, 3),
            "roofline": roof,
        }
    # the literal "random int32" variant: every superblock becomes COPY (reported next to the headline, with its own roofline)
    if world == 1 and T == 4 and not args.kind and not args.no_full_entropy:
        src2 = generate_torch("rand", T, n, seed=42, device=dev)
        k = max(1, min(3, args.steps))
        r2 = run_workload(st, torch, src2, T, k, 1, dist, world)
        assert r2["ok"]
        algo2 = nbytes + r2["csize"]
        t2, s2 = profiled_traffic("int32_rand" if args.gib == 8.0 else None)
        t2d, _ = profiled_traffic("int32_rand" if args.gib == 8.0 else None, "decode_superblocks")
        roof2 = roofline("encode_superblocks", algo2, r2["kenc_ms"], t2, s2)
        roof2["decode_superblocks"] = {kk: v for kk, v in roofline("decode_superblocks", algo2, r2["kdec_ms"], t2d).items() if kk in ("achieved", "frac", "kernel_ms", "traffic")}
        out["full_entropy"] = {"workload": "the literal configs[1]: uniform 32-bit values (splitmix64 seed 42), every superblock stored as a copy",
                               "value": round(nbytes * k / r2["wall"] / 1e9, 3), "compression_ratio": round(nbytes / r2["csize"], 5),
                               "encode_gbps": round(nbytes * k / r2["enc_s"] / 1e9, 3), "decode_gbps": round(nbytes * k / r2["dec_s"] / 1e9, 3),
                               "decode_indexed_gbps": round(nbytes * k / r2["dec_indexed_s"] / 1e9, 3), "roofline": roof2}
        del src2
    if world == 1 and not args.no_cpu_baseline:
        mib = min(args.cpu_sample_mib, nbytes >> 20)
        sample_bytes = (mib << 20) + 4000 if (mib << 20) + 4000 <= nbytes else nbytes  # not a superblock multiple: the reference decoder rejects those
        sample = src[:sample_bytes].cpu().numpy()
        out["cpu_baseline"] = cpu_baseline(sample, T, f"first {mib} MiB + 4000 B of the same workload, best of 5")
        # the whole frame against the reference's, slice by slice (parity_prefix_bytes: the name of rounds 2-4, when it was a prefix)
        out["parity_bytes"] = out["parity_prefix_bytes"] = reference_frame_parity(st, torch, src, T)
        out["parity_frame_bytes"] = int(r["csize"])
        out["host_pointer"] = host_pointer_rate(T, sample)
        del sample
    if world == 1 and args.config == "int32" and not args.kind and not args.no_other_configs:
        del src
        torch.cuda.empty_cache()
        out["other_configs"] = {name: measure_other(name, torch, dist, dev, not args.no_cpu_baseline) for name in OTHER_CONFIGS}
        src = None
    if world > 1:
        # The one exchange of the sharded form, after the timed region and timed on its own: the compressed segments go to
        # rank 0 (RCCL over xGMI with the nccl backend), rank 0 cuts the assembled frame again and every rank decodes its
        # segment; each rank checks its slice against its input.
        from stenos_amd.sharded import decompress_sharded, gather_frames

        # Every stage that can fail on one rank alone (allocation, a codec call) is followed by an all_reduce of an ok flag:
        # either all ranks enter the next collective or none does, so a local failure is reported instead of hanging the job.
        extra = {}
        cdev = dev if dist.get_backend() == "nccl" else "cpu"

        def all_ok(ok):
            t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item() == 1.0)

        stage, err = "compress", None
        csize, frame, index, back = 0, None, None, None
        try:
            dst = torch.empty(st.bound(nbytes), dtype=torch.uint8, device=dev)
            csize = st.compress(src, T, dst)
            local = dst[:csize]
            back = torch.empty_like(src)
        except Exception as e:
            err = repr(e)[:300]
        if all_ok(err is None):
            stage = "gather"
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            frame = gather_frames(local, world * nbytes)  # (collectives only: a failure here is a failure of the job)
            torch.cuda.synchronize()
            dist.barrier()
            gather_s = time.perf_counter() - t0
            sizes = torch.tensor([float(csize)], dtype=torch.float64, device=cdev)
            dist.all_reduce(sizes)
            moved = float(sizes.item()) - csize if rank == 0 else 0.0
            try:
                index = st.frame_index(frame, T, frame.numel()) if rank == 0 else None
            except Exception as e:
                err = repr(e)[:300]
            if all_ok(err is None):
                stage = "scatter_decode"
                derr = []

                def decode(seg, nb):
                    try:
                        st.decompress(seg, T, seg.numel(), back[:nb])
                    except Exception as e:  # reported after the collectives of decompress_sharded are through
                        derr.append(repr(e)[:300])
                    return back[:nb]

                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                part, o0, o1 = decompress_sharded(decode, frame, index, world * nbytes, T, dev)
                torch.cuda.synchronize()
                dist.barrier()
                scatter_decode_s = time.perf_counter() - t0
                good = not derr and o1 - o0 == nbytes and o0 == rank * nbytes and bool(torch.equal(part, src))
                extra = {"gather_ms": round(gather_s * 1e3, 3), "gather_bytes_to_rank0": int(moved), "gather_gbps": round(moved / gather_s / 1e9, 2) if gather_s > 0 else None,
                         "scatter_decode_ms": round(scatter_decode_s * 1e3, 3), "sharded_roundtrip_ok": all_ok(good), "backend": dist.get_backend()}
        if not extra:
            extra = {"gather_error": f"stage {stage}: " + (err or "another rank failed")}
        if rank == 0:
            out["sharded_exchange"] = extra
    if rank == 0:
        print(json.dumps(out), flush=True)
    st.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
