"""Device-resident path (include/stenos_hip.h) at sizes the oracle cannot reach in seconds: size
independent properties -- encode -> decode round trips on torch CUDA tensors, indexed and walked
decoding agree, frames equal the host-ABI frames, compressed size is additive over superblocks."""
import os

import numpy as np
import pytest

from stenos_amd.api import Stenos, StenosError
from stenos_amd.datagen import generate, generate_torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def st():
    import torch

    assert torch.cuda.is_available()
    s = Stenos(level=1)
    yield s
    s.close()


@pytest.mark.parametrize("kind,T,n", [("rand12", 4, (256 << 20) // 4), ("walk", 2, (128 << 20) // 2 + 77), ("sine", 8, (128 << 20) // 8 + 5),
                                      ("rand", 4, (64 << 20) // 4 + 1), ("sorted_i32", 4, 50_000_000)])
def test_device_roundtrip_large(st, kind, T, n):
    import torch

    src = generate_torch(kind, T, n, 42)
    dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
    csize = st.compress(src, T, dst)
    assert 8 < csize <= st.bound(src.numel())
    idx, nsb = st.last_index()
    sb = 131072 // (256 * T) * 256 * T
    assert nsb == (src.numel() + sb - 1) // sb
    # indexed decode
    back = torch.zeros_like(src)
    assert st.decompress(dst, T, csize, back, index_ptr=idx) == src.numel()
    assert torch.equal(back, src)
    # frame-only decode (device walks the superblock chain); the index buffer is reused, so copy the frame first
    frame = dst[:csize].clone()
    back.zero_()
    assert st.decompress(frame, T, csize, back, index_ptr=None) == src.numel()
    assert torch.equal(back, src)
    # idempotence: same input, same frame
    dst2 = torch.empty_like(dst)
    assert st.compress(src, T, dst2) == csize
    assert torch.equal(dst2[:csize], dst[:csize])


@pytest.mark.parametrize("T,kind", [(4, "mixed"), (2, "walk"), (4, "lzmix"), (8, "sine")])
def test_device_pointers_of_any_alignment(st, oracle, T, kind):
    """Source, destination and decoded buffers that start at odd device addresses: the encoder's register path needs
    16-byte aligned blocks and must fall back, the stores must not touch a byte outside their ranges."""
    import torch
    from _libs import oracle_compress

    n = 600_003
    data = generate(kind, T, n, 9)
    r, ref = oracle_compress(oracle, data, T, 1)
    for so, do in ((1, 0), (4, 3), (8, 5), (13, 16)):
        buf = torch.zeros(data.nbytes + 64, dtype=torch.uint8, device="cuda")
        buf[so:so + data.nbytes] = torch.from_numpy(data).cuda()
        src = buf[so:so + data.nbytes]
        out = torch.full((st.bound(data.nbytes) + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        dst = out[do:do + st.bound(data.nbytes)]
        csize = st.compress(src, T, dst)
        assert csize == r, (so, do)
        got = out.cpu().numpy()
        assert np.array_equal(got[do:do + csize], ref), (so, do)
        assert (got[:do] == 0xA5).all() and (got[do + st.bound(data.nbytes):] == 0xA5).all(), (so, do)
        back_buf = torch.full((data.nbytes + 64,), 0x5A, dtype=torch.uint8, device="cuda")
        back = back_buf[do:do + data.nbytes]
        assert st.decompress(dst, T, csize, back, index_ptr=None) == data.nbytes
        b = back_buf.cpu().numpy()
        assert np.array_equal(b[do:do + data.nbytes], data), (so, do)
        assert (b[:do] == 0x5A).all() and (b[do + data.nbytes:] == 0x5A).all(), (so, do)


def test_device_frame_equals_host_abi_frame(st):
    import torch

    data = generate("burst", 4, 3_000_001, 5)
    src = torch.from_numpy(data).cuda()
    dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
    csize = st.compress(src, 4, dst)
    out = np.zeros(st.bound(data.nbytes), dtype=np.uint8)
    r = st.lib.stenos_compress(data.ctypes.data, 4, data.nbytes, out.ctypes.data, out.nbytes, 1)
    assert r == csize
    assert np.array_equal(dst[:csize].cpu().numpy(), out[:r])


def test_superblock_additivity(st):
    """Superblocks are independent units (stenos.cpp:893-904): compressing two halves cut at a superblock
    boundary gives the same superblock bytes as compressing the whole."""
    import torch

    T, sb = 4, 131072
    src = generate_torch("rand12", T, 64 * sb // T + 333, 3)
    dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
    c_all = st.compress(src, T, dst)
    whole = dst[:c_all].cpu().numpy()
    cut = 40 * sb
    a = torch.empty(st.bound(cut), dtype=torch.uint8, device="cuda")
    ca = st.compress(src[:cut].contiguous(), T, a)
    b = torch.empty(st.bound(src.numel() - cut), dtype=torch.uint8, device="cuda")
    cb = st.compress(src[cut:].contiguous(), T, b)
    assert c_all == ca + cb - 8
    assert np.array_equal(whole[8:ca], a[8:ca].cpu().numpy())
    assert np.array_equal(whole[ca:], b[8:cb].cpu().numpy())


def test_dst_too_small_on_device(st):
    import torch

    src = generate_torch("rand", 4, 100_000, 1)
    small = torch.full((1000,), 7, dtype=torch.uint8, device="cuda")
    with pytest.raises(StenosError):
        st.compress(src, 4, small)
    assert bool((small == 7).all())


def test_async_compress_then_finish(st):
    import torch

    src = generate_torch("walk", 2, 5_000_000, 2)
    dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
    assert st.compress(src, 2, dst, wait=False) == 0
    csize = st.finish()
    back = torch.zeros_like(src)
    idx, _ = st.last_index()
    assert st.decompress(dst, 2, csize, back, index_ptr=idx, wait=False) == 0
    assert st.finish() == src.numel()
    assert torch.equal(back, src)


def test_fused_launch_that_gives_up_is_redone_without_the_fused_kernel(hooks_lib):
    """The fused encoder bounds its waits for frame offsets; a launch that gives up (injected here: it has never been
    observed) must end in the same frame, produced by encode_blocks / plan / scan / pack, not in an error."""
    import torch

    for kind, T, n in (("rand12", 4, 3_000_003), ("walk", 2, 2_000_001), ("sine", 8, 700_001)):
        src = generate_torch(kind, T, n, 5)
        s = Stenos(level=1, lib=hooks_lib)
        dst = torch.zeros(s.bound(src.numel()), dtype=torch.uint8, device="cuda")
        want = s.compress(src, T, dst)
        ref = dst[:want].clone()
        assert s.lib.stenos_hip_fused_fallbacks(s.ctx) == 0
        s.lib.stenos_hip_test_fused_timeouts(s.ctx, 2)
        for wait in (True, False):
            dst.zero_()
            got = s.compress(src, T, dst) if wait else (s.compress(src, T, dst, wait=False), s.finish())[1]
            assert got == want and torch.equal(dst[:got], ref)
        assert s.lib.stenos_hip_fused_fallbacks(s.ctx) == 2
        dst.zero_()
        assert s.compress(src, T, dst) == want and torch.equal(dst[:want], ref)  # and the fused kernel again afterwards
        assert s.lib.stenos_hip_fused_fallbacks(s.ctx) == 2
        back = torch.zeros_like(src)
        assert s.decompress(dst, T, want, back) == src.numel() and torch.equal(back, src)
        s.close()


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """bench.py's N > 1 control flow (rank sharding, barrier, max over ranks, one JSON line from rank 0)
    exercised with two processes on the single GPU of the test box (gloo for the two collectives)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STENOS_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gib", "0.25"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and abs(d["compression_ratio"] - 2.52) < 0.01
    x = d["sharded_exchange"]  # the gather of the segments to rank 0 and the sharded decode, run after the timed region
    assert "gather_error" not in x, x
    assert x["sharded_roundtrip_ok"] is True and x["gather_ms"] > 0 and x["gather_bytes_to_rank0"] > 0


SHARDED_WORKER = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from stenos_amd.api import Stenos
from stenos_amd.sharded import compress_sharded, decompress_sharded
from stenos_amd.datagen import generate_torch
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
torch.cuda.set_device(0)
st = Stenos(level=1)
for kind, T, n in (("rand12", 4, 7 * 32768 + 1234), ("walk", 2, 11 * 65536 + 99), ("rand", 4, 5 * 32768 + 7)):
    data = generate_torch(kind, T, n, 42, device="cuda:0")   # every rank holds the whole array, as compress_sharded expects
    def compress(chunk):
        dst = torch.empty(st.bound(chunk.numel()), dtype=torch.uint8, device="cuda:0")
        return dst[:st.compress(chunk, T, dst)].clone()
    frame = compress_sharded(compress, data, T)
    rank0 = dist.get_rank() == 0
    if rank0:
        assert torch.equal(frame, compress(data)), "sharded frame differs from the single-GPU frame"
    back = torch.empty_like(data)
    def decompress(seg, nb):
        out = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
        assert st.decompress(seg, T, seg.numel(), out) == nb
        return out
    whole, o0, o1 = decompress_sharded(decompress, frame if rank0 else None, st.frame_index(frame, T, frame.numel()) if rank0 else None, data.numel(), T,
                                       "cuda:0", gather_output=True)
    if rank0:
        assert torch.equal(whole, data), "sharded decode differs from the input"
if dist.get_rank() == 0:
    print("SHARDED_GPU_OK")
dist.barrier()
dist.destroy_process_group()
"""


def test_sharded_codec_two_ranks_on_one_gpu(tmp_path):
    """SURVEY 8e end to end through the GPU codec at level 1: two ranks (on the one GPU of the test box, gloo moving the
    segments) compress their superblock ranges, rank 0 assembles the frame -- equal to the single-process frame -- cuts it
    again with the device-side header walk, and both ranks decode their segments."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(SHARDED_WORKER.format(root=root))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert "SHARDED_GPU_OK" in p.stdout


@pytest.mark.gpu
def test_library_before_torch_in_fresh_process():
    """libstenos.so loaded before torch must still share one HIP runtime with it."""
    import subprocess
    import sys

    code = (
        "from stenos_amd.api import Stenos\n"
        "st = Stenos(level=1)\n"
        "import torch\n"
        "src = (torch.arange(1 << 20, dtype=torch.int32, device='cuda') & 0xFFF).view(torch.uint8)\n"
        "dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device='cuda')\n"
        "back = torch.empty_like(src)\n"
        "c = st.compress(src, 4, dst)\n"
        "st.decompress(dst, 4, c, back)\n"
        "assert torch.equal(src, back)\n"
        "print('ok', c)\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stderr[-2000:]
