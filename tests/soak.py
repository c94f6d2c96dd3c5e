#!/usr/bin/env python3
"""Parity soak on a GPU box (not collected by pytest): the seeded GPU parity tests again, with seeds nobody
looked at -- chains of every preset / flag set on fresh synthetic batches with ragged lengths, read lengths from
30 to 300, adversarial single-adapter reads -- until the time budget is spent.  Every comparison is bit-exact
against the CPU oracle (tests/test_gpu_parity.run_both).

    python tests/soak.py [seconds] [first_seed]
"""
import random
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np  # noqa: E402

import test_gpu_parity as tg  # noqa: E402
import util  # noqa: E402
from cutseq_amd import plan as planmod, synth  # noqa: E402
from cutseq_amd.common import BUILDIN_ADAPTERS  # noqa: E402
from test_oracle import CHAIN_CASES  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
    t0 = time.time()
    done = {"chains": 0, "fuzz": 0, "stress": 0}
    while time.time() - t0 < budget:
        rng = random.Random(seed)
        name, flags, paired = CHAIN_CASES[rng.randrange(len(CHAIN_CASES))]
        scheme = BUILDIN_ADAPTERS.get(name, name)
        st = planmod.CutadaptConfig()
        for k, v in flags.items():
            setattr(st, k, v)
        st.select_rule = rng.randrange(2)
        read_len = rng.choice([30, 75, 100, 150, 151, 250, 300])
        batch = synth.generate_pairs(rng.choice([1, 63, 64, 65, 1000, 20_000]), read_len, scheme, seed=seed,
                                     chunk_index=rng.randrange(1000), single_end=not paired,
                                     adapter_fraction=rng.choice([0.0, 0.35, 0.9]), poly_fraction=rng.choice([0.0, 0.02, 0.3]),
                                     art5_fraction=rng.choice([0.001, 0.2]), indel_frac=rng.choice([0.03, 0.3]),
                                     sub_rate=rng.choice([0.0, 0.01, 0.08]))
        nrng = np.random.default_rng(seed)
        for lens in (batch.len1, batch.len2):
            if lens is not None and rng.random() < 0.5:
                cut = nrng.random(batch.n) < 0.25
                lens[cut] = nrng.integers(0, read_len, size=int(cut.sum())).astype(np.uint16)
        if rng.random() < 0.5:
            util.soft_mask(batch, 0.2, seed=seed)
        tg.run_both(util.compile_plan(scheme, st, paired), batch)
        done["chains"] += 1
        if seed % 5 == 0:
            tg.test_fuzz_odd_alphabets_qualities_and_lengths(seed)
            done["fuzz"] += 1
        if seed % 7 == 0:
            tg.test_hits_that_settle_in_the_filter_stress(seed)
            done["stress"] += 1
        seed += 1
        if seed % 20 == 0:
            print(f"{time.time() - t0:6.0f} s  seed {seed}  {done}", flush=True)
    print(f"soak ok: {done}, seeds up to {seed - 1}, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
