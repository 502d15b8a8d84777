"""bench.py as the driver starts it, and the collectives of the sharded form on real backends.

* `python bench.py --gpus N` WITHOUT an outer torch.distributed.run: the script starts its ranks itself as a child process
  (nothing that has touched the GPU ever execs) and relays rank 0's line.  One GPU here: STENOS_BENCH_ONE_DEVICE=1 puts every
  rank on cuda:0 and moves the collectives to gloo.  With 2 ranks and with 4: this pool lets a job have six processes on a
  card at once, the test runner being one of them, so the 8-way partition, its segment table and its empty ranges run on the
  CPU instead (tests/test_sharded_cpu.py, world size 8 over gloo).
* gather_frames / decompress_sharded with the "nccl" backend (RCCL), world size 1, on cuda:0: RCCL loads and the
  device-tensor collectives and point-to-point calls of stenos_amd/sharded.py execute.
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from _libs import ROOT

pytestmark = pytest.mark.gpu


def _bench(gpus, gib):
    env = dict(os.environ, STENOS_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--gib", str(gib), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("gpus,gib", [(2, 0.25), (4, 0.125)])
def test_bench_starts_its_own_ranks(gpus, gib):
    out = _bench(gpus, gib)
    assert out["n_gpus"] == gpus and out["scaling"] == "weak" and out["value"] > 0
    ex = out["sharded_exchange"]
    assert ex.get("sharded_roundtrip_ok") is True, ex
    assert ex["backend"] == "gloo" and ex["gather_bytes_to_rank0"] > 0


NCCL_WORKER = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    from stenos_amd.api import Stenos
    from stenos_amd.datagen import generate_torch
    from stenos_amd.sharded import compress_sharded, decompress_sharded
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    T = 4
    src = generate_torch("rand12", T, (24 << 20) // T + 333, 42, device="cuda:0")
    st = Stenos(1)
    def compress(chunk):
        dst = torch.empty(st.bound(chunk.numel()), dtype=torch.uint8, device="cuda:0")
        return dst[:st.compress(chunk, T, dst)]
    t = torch.ones(4, device="cuda:0"); dist.all_reduce(t); assert float(t.sum()) == 4.0   # RCCL runs a collective on device memory
    frame = compress_sharded(compress, src, T)
    assert frame.is_cuda and torch.equal(frame, compress(src)), "sharded frame differs from the single-GPU frame"
    index = st.frame_index(frame, T, frame.numel())
    def decompress(local, nbytes):
        out = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
        assert st.decompress(local, T, local.numel(), out) == nbytes
        return out
    whole, o0, o1 = decompress_sharded(decompress, frame, index, src.numel(), T, "cuda:0", gather_output=True)
    assert (o0, o1) == (0, src.numel()) and torch.equal(whole, src)
    dist.barrier(); dist.destroy_process_group()
    print("NCCL_OK")
""")


def test_sharded_collectives_on_rccl(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(NCCL_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "NCCL_OK" in p.stdout
