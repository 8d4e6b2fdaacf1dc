#!/usr/bin/env python3
"""Extended randomised sweep: the bodies of tests/test_gpu_parity.py::test_random_shapes_against_the_oracle (small batches, every
forward path and backward form) and ::test_random_shapes_large_batches (>= 2^18 samples, fp32 / fp16) over many more seeds than the
suite runs.   usage: fuzz_shapes.py [small_seeds] [large_seeds] [first_seed] [planned_seeds]   -> one line per failure, a summary line"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_parity as T
import test_gpu_plan as P
from shacira_amd import _lib

n_small = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_large = int(sys.argv[2]) if len(sys.argv) > 2 else 30
first = int(sys.argv[3]) if len(sys.argv) > 3 else 100
n_plan = int(sys.argv[4]) if len(sys.argv) > 4 else n_large
dev = torch.device("cuda:0")
fails = 0
for name, fn, count in (("small", T.test_random_shapes_against_the_oracle, n_small),
                        ("large", T.test_random_shapes_large_batches, n_large),
                        ("planned", P.test_random_shapes_through_the_plan, n_plan)):
    # (the planned sweep alternates the automatic brick rule with "wherever the shape allows", as tests/test_gpu_plan.py's
    # fixture sets it)
    for seed in range(first, first + count):
        _lib.set_option("bwd_brick", 1 if (name == "planned" and seed % 2 == 0) else -1)
        try:
            fn(dev, seed)
        except Exception as exc:      # noqa: BLE001 -- report and go on
            fails += 1
            print(f"FAIL {name} seed {seed}: {type(exc).__name__}: {str(exc)[:300]}", flush=True)
            traceback.print_exc(limit=2)
_lib.set_option("bwd_brick", -1)
print(f"fuzz: {n_small} small + {n_large} large + {n_plan} planned shapes from seed {first}: {fails} failures", flush=True)
sys.exit(1 if fails else 0)
