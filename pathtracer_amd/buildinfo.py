"""Identity of the kernel sources a measurement belongs to: profiles/pmc_traffic.json is stamped with it by
tools/summarize_profile.py and bench.py refuses counter data taken from other sources."""
import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_source_hash():
    """sha256 over csrc/*.hpp (every kernel lives there) and the Makefile (the compiler flags), names and bytes,
    sorted; first 16 hex digits. The .hip files hold host code only (the C-ABI, launch sequences)."""
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(_CSRC, "*.hpp")) + [os.path.join(_CSRC, "Makefile")]):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]
