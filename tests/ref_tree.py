"""Where the reference tree lives: /root/reference in the build container, oracle/_ref/ (staged by tools/stage_reference.sh,
git-ignored) on a builder-side gpurun box, nowhere in the driver's round-end run.  Test infrastructure only."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_root():
    for p in ("/root/reference", os.path.join(REPO, "oracle", "_ref")):
        if os.path.isdir(os.path.join(p, "auto_round")):
            return p
    return None


def import_reference():
    """Put the reference (and the `cpuinfo` shim it needs at import time) on sys.path; never write bytecode into it."""
    root = reference_root()
    if root is None:
        raise ImportError("reference tree not present")
    sys.dont_write_bytecode = True
    shim = os.path.join(REPO, "oracle", "ref_shim")
    for p in (shim, root):
        if p not in sys.path:
            sys.path.insert(0, p)
    import auto_round  # noqa: F401

    return root
