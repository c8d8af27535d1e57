"""The host builder cuts large inputs across worker threads (parallel bins and exchanges at the top of
the tree, whole subtrees below); the tree must not depend on how many there are: node for node and
primitive for primitive the one a single thread builds (which the golden fixtures pin to the reference's)."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = """
import sys, hashlib
sys.path.insert(0, %r)
from pathtracer_amd import capi, synth
verts, idx = synth.soup(150000, key=91, half=0.3)
b = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
print(hashlib.sha1(b.nodes().tobytes() + b.order().tobytes()).hexdigest(), len(b.nodes()), b.stats())
""" % ROOT


def build_digest(threads):
    env = dict(os.environ, PTX_BUILD_THREADS=str(threads))
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return out.stdout.strip().splitlines()[-1]


def test_tree_does_not_depend_on_thread_count():
    one = build_digest(1)
    assert build_digest(3) == one
    assert build_digest(8) == one
