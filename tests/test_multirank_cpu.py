"""The N > 1 path rehearsed on CPU: two gloo ranks shard one read stream exactly as bench.py
does (contiguous ranges, per-rank generation by global index, no data-path collective), trim
their shard (with the oracle standing in for the GPU, which is absent here) and only
reduce counters.  Rank 0 checks the union against a single-process run."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from cutseq_amd import plan as planmod, shard, synth  # noqa: E402
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig  # noqa: E402

N_TOTAL = 30_001  # odd on purpose: uneven shards


def _plan():
    st = planmod.CutadaptConfig()
    st.trim_polyA = True
    return planmod.compile_paired(BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"]), st)


def _trim(batch, tp):
    import oracle
    a1, n1, a2, n2 = tp.pack()
    p = tp.params()
    r1, _, s1 = oracle.trim_mate(a1, n1, p, batch.seq1, batch.qual1, batch.len1)
    r2, _, s2 = oracle.trim_mate(a2, n2, p, batch.seq2, batch.qual2, batch.len2)
    return r1, r2, s1, s2


def _worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tp = _plan()
    lo, hi = shard.shard_bounds(N_TOTAL, rank, world)
    batch = synth.generate_pairs(hi - lo, 150, first_index=lo, threads=2)
    r1, r2, s1, s2 = _trim(batch, tp)
    np.save(Path(tmpdir) / f"r1_{rank}.npy", r1)
    np.save(Path(tmpdir) / f"r2_{rank}.npy", r2)
    # the only collective of the job: counters (here via all_reduce, in the product on the host)
    vec = torch.tensor([s1.n_reads, s1.out_bp, s1.op_matched[1], s2.n_reads, s2.out_bp, s2.qualtrim_bp],
                       dtype=torch.int64)
    dist.all_reduce(vec)
    if rank == 0:
        np.save(Path(tmpdir) / "reduced.npy", vec.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 64, 1000, N_TOTAL):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_two_ranks_equal_one_process(tmp_path):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    tp = _plan()
    whole = synth.generate_pairs(N_TOTAL, 150, first_index=0, threads=2)
    w1, w2, s1, s2 = _trim(whole, tp)
    got1 = np.concatenate([np.load(tmp_path / f"r1_{r}.npy") for r in range(world)])
    got2 = np.concatenate([np.load(tmp_path / f"r2_{r}.npy") for r in range(world)])
    assert np.array_equal(got1, w1) and np.array_equal(got2, w2)
    red = np.load(tmp_path / "reduced.npy")
    assert list(red) == [s1.n_reads, s1.out_bp, s1.op_matched[1], s2.n_reads, s2.out_bp, s2.qualtrim_bp]
    merged = shard.merge_stats([s1.as_dict(), s1.as_dict()])
    assert merged["n_reads"] == 2 * s1.n_reads and merged["op_matched"][1] == 2 * s1.op_matched[1]


def test_bench_self_launch_spawns_ranks_and_relays_failure():
    """``python bench.py --gpus 2`` with no launcher around it: the parent spawns one child per rank with the
    torch.distributed.run environment and exits non-zero when a child does.  Without a GPU every child stops at
    "bench.py needs an MI355X" -- which is exactly what shows that two ranks were started with WORLD_SIZE=2."""
    import subprocess

    if torch.cuda.is_available():
        pytest.skip("covered on the GPU by tests/test_gpu_cli.py::test_bench_two_ranks_rehearsal_on_one_gpu")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--pairs", "1000"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert out.stderr.count("needs an MI355X") >= 1, out.stderr[-1000:]
    assert "does not match --gpus" not in out.stderr
