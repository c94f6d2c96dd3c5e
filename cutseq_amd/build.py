"""In-tree build of the HIP library:  python -m cutseq_amd.build  (hipcc cross-compiles gfx950 without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
SRC = HERE / "csrc" / "cutseq_hip.hip"
DEPS = [SRC, HERE / "csrc" / "trim_kernel.hip.inc", HERE / "csrc" / "long_kernel.hip.inc",
        HERE / "csrc" / "text_kernels.hip.inc", HERE / "csrc" / "deflate_kernels.hip.inc",
        HERE.parent / "include" / "cutseq_hip.h"]
OUT = HERE / "libcutseq_hip.so"


def kernel_source_hash() -> str:
    """sha256 over the device sources: counter files under profiles/ carry it, bench.py replays a file only when
    it was collected on the kernels it is running (tools/summarize_pmc.py)."""
    import hashlib
    h = hashlib.sha256()
    for path in (HERE / "csrc" / "trim_kernel.hip.inc", SRC):
        # (the text kernels are not what the counters measure)
        h.update(path.read_bytes())
    return h.hexdigest()


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found (ROCm toolchain required)")
    return exe


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and OUT.exists() and OUT.stat().st_mtime >= max(p.stat().st_mtime for p in DEPS):
        return OUT
    cmd = [hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-o", str(OUT), str(SRC)]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    env = dict(os.environ)
    subprocess.run(cmd, check=True, cwd=str(SRC.parent), env=env)
    return OUT


HOST_SRC = HERE / "csrc" / "cutseq_host.c"
HOST_SRCS = (HOST_SRC, HERE / "csrc" / "pinflate.c")
HOST_DEPS = HOST_SRCS + (HERE / "csrc" / "pinflate_loop.h",)
HOST_OUT = HERE / "libcutseq_host.so"


def build_host(force: bool = False) -> Path:
    """Host-only helpers (synthetic generator): plain gcc, no GPU toolchain involved."""
    if not force and HOST_OUT.exists() and HOST_OUT.stat().st_mtime >= max(p.stat().st_mtime for p in HOST_DEPS):
        return HOST_OUT
    cmd = ["gcc", "-O3", "-std=gnu11", "-fPIC", "-shared", "-Wall", "-o", str(HOST_OUT)] + [str(p) for p in HOST_SRCS] + ["-lpthread"]
    subprocess.run(cmd, check=True)
    return HOST_OUT


SYNTH_SRC = HERE / "csrc" / "synth_device.hip"
SYNTH_DEPS = (SYNTH_SRC, HERE.parent / "include" / "cutseq_synth.h")
SYNTH_OUT = HERE / "libcutseq_synth.so"


def build_synth(force: bool = False) -> Path:
    """The synthetic generator's device form (bench / test infrastructure, its own library: the trimming library
    carries no generator)."""
    if not force and SYNTH_OUT.exists() and SYNTH_OUT.stat().st_mtime >= max(p.stat().st_mtime for p in SYNTH_DEPS):
        return SYNTH_OUT
    cmd = [hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", str(SYNTH_OUT), str(SYNTH_SRC)]
    subprocess.run(cmd, check=True, cwd=str(SYNTH_SRC.parent))
    return SYNTH_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
    print(build_host(force="--force" in sys.argv))
    print(build_synth(force="--force" in sys.argv))
