import sys, os, time, json, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pathlib import Path
from concurrent.futures import ThreadPoolExecutor
from cutseq_amd import workloads, run as cli
from tools import tiers
n = 4_000_000
batch = workloads.make_batch("config3", n)
for piece in (262_144, 1 << 30, 65_536 * 4 * 4):
    work = Path("/dev/shm/cutseq_single"); shutil.rmtree(work, ignore_errors=True); work.mkdir()
    with ThreadPoolExecutor(16) as pool:
        tiers.write_inputs(work, batch, n, pool, piece_records=min(piece, n if piece > n else piece))
    for rep in range(3):
        for f in work.glob("out_*"): f.unlink()
        t0 = time.perf_counter()
        try:
            cli.main(["-A", "TAKARAV3", "--trim-polyA", str(work / "single_R1.fastq.gz"), str(work / "single_R2.fastq.gz"), "-O", str(work / "out")])
        except SystemExit as e:
            if e.code: raise
        dt = time.perf_counter() - t0
        print(f"pieces of {piece}: run {rep}: {dt:.3f} s = {n / dt / 1e6:.2f} M pairs/s", flush=True)
    shutil.rmtree(work, ignore_errors=True)
