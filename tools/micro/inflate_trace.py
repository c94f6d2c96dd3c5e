#!/usr/bin/env python3
"""Where do the pool's threads spend a gzip read?  Every pool task of GzipSource.blocks() with its thread, kind, start and
end; printed: busy share of the pool over the run, per-kind totals, the longest gaps per thread.
    python3 tools/micro/inflate_trace.py single|syn [pairs] [files]"""
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from cutseq_amd import codec, fastq  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "single"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
n_files = int(sys.argv[3]) if len(sys.argv) > 3 else 2
work = Path("/dev/shm/cutseq_inflate_trace")
work.mkdir(exist_ok=True)
subprocess.run([sys.executable, str(ROOT / "tools" / "make_fastq.py"), str(n), str(work / "syn")], check=True)
if kind == "single":
    for m in (1, 2):
        with open(work / f"single_R{m}.fastq.gz", "wb") as out:
            p1 = subprocess.Popen(["gzip", "-dc", str(work / f"syn_R{m}.fastq.gz")], stdout=subprocess.PIPE)
            p2 = subprocess.Popen(["gzip", "-1"], stdin=p1.stdout, stdout=out)
            p2.wait()
            p1.wait()

pool = fastq._pool()
events = []
real_submit = pool.submit


def submit(fn, *args, **kw):
    t_sub = time.perf_counter()

    def run():
        t0 = time.perf_counter()
        try:
            return fn(*args, **kw)
        finally:
            events.append((threading.get_ident(), getattr(fn, "__name__", str(fn)), t_sub, t0, time.perf_counter()))
    return real_submit(run)


pool.submit = submit


marks = []


def drain(path, box):
    m = [time.perf_counter()]
    src = codec.GzipSource(str(path), pool, fastq.ARENA.take, fastq.ARENA.give)
    m.append(time.perf_counter())  # opened
    tot = 0
    first = None
    for arr, nbytes in src.blocks():
        if first is None:
            first = time.perf_counter()
        tot += nbytes
        fastq.ARENA.give(arr)
    m.append(first)                 # first block
    m.append(time.perf_counter())   # last block
    box.append(tot)
    src.close()
    m.append(time.perf_counter())   # closed
    marks.append(m)


files = [work / f"{kind}_R{m}.fastq.gz" for m in (1, 2)][:n_files]
for rep in range(2):
    events.clear()
    marks.clear()
    box = []
    t0 = time.perf_counter()
    ts = [threading.Thread(target=drain, args=(f, box)) for f in files]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    t1 = time.perf_counter()
    dt = t1 - t0
    workers = fastq.pool_size()
    busy = sum(e[4] - e[3] for e in events)
    print(f"{kind} {len(files)} file(s) rep {rep}: {sum(box) / dt / 1e9:.2f} GB/s, wall {dt:.3f} s, pool busy {busy:.3f} s = "
          f"{100 * busy / (dt * workers):.0f} % of {workers} threads, {len(events)} tasks")
    kinds = {}
    for _, k, ts_, a, b in events:
        c = kinds.setdefault(k, [0, 0.0, 0.0])
        c[0] += 1
        c[1] += b - a
        c[2] += a - ts_
    for k, (c, s, q) in sorted(kinds.items()):
        print(f"   {k:24s} {c:5d} tasks, {s:7.3f} s run ({1e3 * s / c:6.2f} ms each), {1e3 * q / c:7.2f} ms queued each")
    # busy threads over time, 10 slices
    slices = 10
    line = []
    for i in range(slices):
        lo, hi = t0 + dt * i / slices, t0 + dt * (i + 1) / slices
        b = sum(max(0.0, min(e[4], hi) - max(e[3], lo)) for e in events)
        line.append(f"{b / (hi - lo):4.1f}")
    print("   busy threads per tenth of the run:", " ".join(line))
    first_task = min(e[3] for e in events) - t0
    last_task = max(e[4] for e in events) - t0
    for m in marks:
        print("   reader: opened %.1f ms, first block %.1f ms, last block %.1f ms, closed %.1f ms; first pool task %.1f ms, last ends %.1f ms"
              % (tuple(1e3 * (x - t0) for x in m[1:]) + (1e3 * first_task, 1e3 * last_task)))
import shutil
shutil.rmtree(work, ignore_errors=True)
