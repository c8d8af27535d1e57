"""Identity of the kernel sources a measurement belongs to.

Counter summaries under profiles/ are collected in profiler passes of their own, not in the run that prints a
benchmark line; `csrc_hash()` -- a SHA-256 over the library's sources -- is stored with each summary and recomputed by
whoever pairs a summary with a fresh timing, so that counters of one kernel are never silently paired with the time of
another."""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_EXTS = (".hip", ".h", ".cpp")


def csrc_files():
    d = os.path.join(_HERE, "csrc")
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(_EXTS) or f == "Makefile")


def csrc_hash():
    """First 16 hex digits of SHA-256 over (file name, contents) of every source under pathtracer_amd/csrc, in name order."""
    h = hashlib.sha256()
    for path in csrc_files():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())
