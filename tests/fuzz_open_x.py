"""Exploratory parity fuzz for the inflow / outflow branch (GPU box): random row lengths / closures / floors / stretchings / routes, three to
nine substeps of the device against the C oracle on seeded random fields (tests/test_gpu_open_x.py open_x_vs_oracle).  Prints every case
that exceeds 1e-9 or raises.  Test infrastructure (uses oracle/).
    python tests/fuzz_open_x.py [ncases] [seed]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "u-dales_amd")]
from test_gpu_open_x import open_x_vs_oracle  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for idx in range(n):
        nx = int(rng.choice([8, 10, 12, 14, 16, 20, 24, 30, 32, 36, 40, 48, 62, 64, 66, 96, 126, 128]))
        ny = int(rng.choice([4, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64]))
        nz = int(rng.choice([3, 4, 5, 6, 8, 10, 12, 16, 17, 24, 32, 40]))
        sgs = int(rng.choice([0, 1, 2]))
        stretch = float(rng.choice([1.0, 1.0, 1.03, 1.08]))
        floor = bool(rng.integers(0, 2))
        nsub = int(rng.choice([3, 6, 9]))
        route = str(rng.choice(["fused", "fused", "routine", "deferred"]))
        desc = f"#{idx} {nx}x{ny}x{nz} sgs={sgs} stretch={stretch} floor={floor} nsub={nsub} route={route}"
        try:
            err = open_x_vs_oracle((nx, ny, nz), sgs, stretch, floor, nsub, idx * 13 + 5, route)
            worst = max(err.values())
            if not worst <= 1e-9:
                bad += 1
                print("MISMATCH", desc, {k: f"{v:.2e}" for k, v in err.items()}, flush=True)
        except Exception:
            bad += 1
            print("EXCEPTION", desc, flush=True)
            traceback.print_exc()
    print(f"fuzz_open_x: {n} cases, {bad} bad")


if __name__ == "__main__":
    main()
