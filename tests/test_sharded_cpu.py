"""Multi-rank path on CPU: world_size 2 and 8 over gloo.  Level 0 frames are produced by the library's host
framing (no GPU needed), so the sharding / gather / assembly logic is tested for real here; the same
code runs with RCCL on the GPU node."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from _libs import ROOT
from stenos_amd.sharded import shard_ranges, superblock_bytes


def test_shard_ranges_cover_and_align():
    for T in (2, 3, 4, 8):
        sb = superblock_bytes(T)
        for total in (0, 1, sb - 1, sb, sb + 1, 7 * sb + 5, 64 * sb):
            for world in (1, 2, 3, 8):
                r = shard_ranges(total, T, world)
                assert r[0][0] == 0 and r[-1][1] == total
                for (a, b), (c, d) in zip(r, r[1:] + [(total, total)]):
                    assert b == c and a <= b and (a % sb == 0 or a == b == total)
                counts = [(b - a + sb - 1) // sb for a, b in r]
                assert max(counts) - min(counts) <= 1


WORKER = textwrap.dedent("""
    import os, sys, numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    from stenos_amd.api import load_library
    from stenos_amd.sharded import compress_sharded, decompress_sharded, walk_frame_host
    from stenos_amd.datagen import generate
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    lib = load_library()
    T, level = 4, 0
    data = generate("walk", T, {nsb_full} * 32768 + 1234, 3)   # whole superblocks and a short one
    def compress(chunk):
        a = chunk.numpy()
        out = np.zeros(lib.stenos_bound(a.nbytes), dtype=np.uint8)
        r = lib.stenos_compress(a.ctypes.data, T, a.nbytes, out.ctypes.data, out.nbytes, level)
        assert r < (1 << 63)
        return torch.from_numpy(out[:r].copy())
    frame = compress_sharded(compress, torch.from_numpy(data), T)
    # ... and with every rank holding only its own range, as the ranks of a large job do
    from stenos_amd.sharded import shard_ranges
    b, e = shard_ranges(data.nbytes, T, dist.get_world_size())[dist.get_rank()]
    frame2 = compress_sharded(compress, torch.from_numpy(data.view(np.uint8).ravel()[b:e].copy()), T, total_bytes=data.nbytes)
    if dist.get_rank() == 0:
        whole = compress(torch.from_numpy(data))
        assert torch.equal(frame, whole), "sharded frame differs from the single-process frame"
        assert torch.equal(frame2, whole), "frame of per-rank slices differs from the single-process frame"
        back = np.zeros(data.nbytes, dtype=np.uint8)
        f = frame.numpy()
        assert lib.stenos_decompress(f.ctypes.data, T, f.nbytes, back.ctypes.data, back.nbytes) == data.nbytes
        assert np.array_equal(back, data)
    # the decode half: rank 0 cuts the frame at the range boundaries, every rank decodes its own segment
    def decompress(local, nbytes):
        f = local.numpy()
        out = np.zeros(nbytes, dtype=np.uint8)
        assert lib.stenos_decompress(f.ctypes.data, T, f.nbytes, out.ctypes.data, nbytes) == nbytes
        return torch.from_numpy(out)
    rank0 = dist.get_rank() == 0
    whole, o0, o1 = decompress_sharded(decompress, frame if rank0 else None, walk_frame_host(frame, T) if rank0 else None, data.nbytes, T, "cpu",
                                       gather_output=True)
    assert 0 <= o0 <= o1 <= data.nbytes and (o0 % 131072 == 0 or o0 == o1)  # (a rank beyond the last superblock has an empty range)
    if rank0:
        assert np.array_equal(whole.numpy(), data), "sharded decode differs from the input"
        print("SHARDED_OK")
    dist.barrier()
    dist.destroy_process_group()
""")


import pytest


@pytest.mark.parametrize("world,nsb_full,port", [(2, 5, 29517), (8, 5, 29518), (8, 19, 29519)])
def test_ranks_gloo(tmp_path, world, nsb_full, port):
    """world 8 with 6 superblocks: ranks without a superblock of their own (empty segments in both directions); with 20:
    the partition of the real node size, two or three superblocks per rank."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, nsb_full=nsb_full))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "SHARDED_OK" in p.stdout
