#!/usr/bin/env python3
"""Where does tier E spend its time?  One synthetic data set, the CLI in-process several times with the text path's
per-thread profile (CUTSEQ_PROFILE=1) and a few settings (block size, engines per GPU, host path for comparison).
    python tools/e2e_text_probe.py [pairs] [workdir]"""
import json
import os
import shutil
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ["CUTSEQ_PROFILE"] = "1"
os.environ.setdefault("CUTSEQ_CHUNK_READS", "262144")  # the steady-state regime (see tools/e2e_bench.py)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    work = Path(sys.argv[2] if len(sys.argv) > 2 else "/dev/shm/cutseq_e2e")
    work.mkdir(parents=True, exist_ok=True)
    subprocess.run([sys.executable, str(ROOT / "tools" / "make_fastq.py"), str(n), str(work / "syn")], check=True)
    from cutseq_amd import codec, fastq, run as cli
    for m in (1, 2):
        src = codec.GzipSource(str(work / f"syn_R{m}.fastq.gz"), None, fastq.ARENA.take, fastq.ARENA.give)
        with open(work / f"plain_R{m}.fastq", "wb") as dst:
            for arr, nbytes in src.blocks():
                dst.write(memoryview(arr)[:nbytes])
                fastq.ARENA.give(arr)
        src.close()
    plain_in = [str(work / "plain_R1.fastq"), str(work / "plain_R2.fastq")]
    plain_out = ["-o", str(work / "o1.fastq"), str(work / "o2.fastq"), "-s", str(work / "s1.fastq"), str(work / "s2.fastq")]
    gz_in = [str(work / "syn_R1.fastq.gz"), str(work / "syn_R2.fastq.gz")]

    def run(tag, inputs, outputs, **env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update({k: str(v) for k, v in env.items()})
        for stale in work.glob("o[12].fastq"), work.glob("s[12].fastq"), work.glob("gzout_*"):
            for f in stale:  # a run writes NEW files: freeing the previous run's tmpfs pages is not its job
                f.unlink()
        t0 = time.perf_counter()
        try:
            cli.main(["-A", "TAKARAV3", "--trim-polyA"] + inputs + outputs)
        except SystemExit as exc:
            if exc.code:
                raise
        dt = time.perf_counter() - t0
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        print(json.dumps({"run": tag, "seconds": round(dt, 3), "M_pairs_per_s": round(n / dt / 1e6, 2)}), flush=True)

    run("plain warm-up", plain_in, plain_out)
    run("plain", plain_in, plain_out)
    run("plain again", plain_in, plain_out)
    if "--more" in sys.argv:
        run("plain 2 engines", plain_in, plain_out, CUTSEQ_DEVICES="0,0")
        run("plain blocks of 131072", plain_in, plain_out, CUTSEQ_CHUNK_READS=131072)
        run("plain blocks of 524288", plain_in, plain_out, CUTSEQ_CHUNK_READS=524288)
        run("plain host path", plain_in, plain_out, CUTSEQ_TEXT_PATH=0)
    run("gz->gz", gz_in, ["-O", str(work / "gzout")])
    run("gz->gz 2 engines", gz_in, ["-O", str(work / "gzout")], CUTSEQ_DEVICES="0,0")
    run("gz->plain", gz_in, plain_out)
    run("plain->gz", plain_in, ["-O", str(work / "gzout")])
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
