"""sha256 (first 16 hex digits) over the kernel sources (mivos_amd/csrc/*.hip, *.h, include/mivos_hip.h) - what a committed profile
record is valid for.  The PMC summaries (scripts/pmc_traffic.py, scripts/pmc_mfma_util.py) store it under "_meta"; bench.py refuses a
record whose fingerprint differs from the tree it runs in (the kernels changed after the counters were read).

    python scripts/csrc_fingerprint.py        # prints the fingerprint of this tree"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_fingerprint(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "mivos_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "mivos_amd", "csrc", "*.h")) +
                   [os.path.join(root, "include", "mivos_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_fingerprint())
