#!/usr/bin/env python3
"""Write a synthetic paired FASTQ data set (gz) with the seeded generator: tools/make_fastq.py N out_prefix"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np

from cutseq_amd import abi, fastq, plan as planmod, synth
from cutseq_amd.common import BUILDIN_ADAPTERS, BarcodeConfig


def main():
    n = int(sys.argv[1])
    prefix = sys.argv[2]
    level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    tp = planmod.compile_paired(BarcodeConfig(BUILDIN_ADAPTERS["TAKARAV3"]), planmod.CutadaptConfig())
    tp.has_umi = False
    tp.r1.name_suffixes = ()
    tp.r2.name_suffixes = ()
    outs = [fastq.OutputFile(f"{prefix}_R1.fastq.gz", level), fastq.OutputFile(f"{prefix}_R2.fastq.gz", level)]
    step = fastq.CHUNK_READS  # one gzip member per block, like the CLI's own output
    for lo in range(0, n, step):
        m = min(step, n - lo)
        b = synth.generate_pairs(m, 150, first_index=lo)
        names1 = "".join(f"SIM:{lo + i} 1:N:0:IDX\n" for i in range(m)).encode()
        names2 = names1.replace(b" 1:N", b" 2:N")
        lens = np.array([len(x) for x in names1.split(b"\n")[:-1]], dtype=np.int32)
        offs = np.concatenate([[0], np.cumsum(lens + 1)[:-1]]).astype(np.int64)
        res = np.zeros(m, dtype=abi.RESULT_DTYPE)
        res["stop"] = 150
        chunk = fastq.Chunk(m, b.stride, names1, offs, lens, b.seq1, b.qual1, b.len1, names2, offs, lens, b.seq2,
                            b.qual2, b.len2)
        data, _ = fastq.format_chunk(chunk, tp, res, None, res.copy())
        outs[0].write(data[0][0])
        outs[1].write(data[0][1])
    for o in outs:
        o.close()


if __name__ == "__main__":
    main()
