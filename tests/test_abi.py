"""The C ABI library loads, exports every function include/ptx.h declares, and its host-only
entry points (no GPU needed) agree with the CPU checker."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from pathtracer_amd import abi, capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ptx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(ptx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 25
    L = capi.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/ptx.h but not exported: {missing}"
    # and nothing is exported under the ptx_ prefix without a declaration
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r" T (ptx_[a-z0-9_]+)$", out, flags=re.M)))
    assert set(exported) <= set(names), f"exported but not declared: {sorted(set(exported) - set(names))}"


def test_header_compiles_as_c(tmp_path):
    src = tmp_path / "use.c"
    src.write_text('#include "ptx.h"\nint main(void) { PtxRay r; PtxHit h; (void)r; (void)h; return sizeof(PtxRay) == 32 && sizeof(PtxHit) == 8 && sizeof(PtxBvhNode) == 40 ? 0 : 1; }\n')
    exe = tmp_path / "use"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_struct_sizes_match_python_mirror():
    assert C.sizeof(abi.PtxRay) == 32 and C.sizeof(abi.PtxHit) == 8 and C.sizeof(abi.PtxBvhNode) == 40
    assert C.sizeof(abi.PtxTile) == 28 and C.sizeof(abi.PtxCamera) == 24 + 64


def test_version_and_error_reporting():
    L = capi.lib()
    assert L.ptx_version().decode()
    h = C.c_void_p()
    rc = L.ptx_bvh_build_triangles(None, 0, None, 0, 0, C.byref(h))     # max_leaf_size 0
    assert rc != abi.PTX_OK
    assert L.ptx_last_error().decode()


def test_make_tiles_matches_checker(oracle):
    """Pathtracer::render's tile list (pathtracer.cpp:522-563): sizes, seeds, centre-out order."""
    for w, h, spp, seed in ((256, 256, 16, 4660), (1024, 1024, 256, 7), (130, 90, 51, 1), (99, 100, 50, 3), (1, 1, 1, 9)):
        a, b = capi.make_tiles(w, h, spp, seed), oracle.make_tiles(w, h, spp, seed)
        assert a.tobytes() == b.tobytes()
        n = ((w + 99) // 100) * ((h + 99) // 100) * ((spp + 49) // 50)
        assert len(a) == n
        # every (pixel, sample) is covered exactly once
        cover = np.zeros((h, w), dtype=np.int64)
        for t in a:
            cover[t["y_begin"]:t["y_end"], t["x_begin"]:t["x_end"]] += int(t["s_end"]) - int(t["s_begin"])
        assert (cover == spp).all()


def test_accumulator_to_image_and_tonemap_match_checker(oracle):
    """accumulator_to_image (pathtracer.cpp:429-442), HDR_Image::tonemap_to (hdr_image.cpp:201-225)."""
    g = np.random.Generator(np.random.Philox(key=8))
    n = 5000
    acc = (g.random(n * 3) * (1 << 30)).astype(np.int64)
    acc[::17] = -acc[::17]
    smp = g.integers(0, 300, n).astype(np.uint32)
    smp[::13] = 0
    a, b = capi.accumulator_to_image(acc, smp), oracle.accumulator_to_image(acc, smp)
    assert a.tobytes() == b.tobytes()
    rgb = (g.random((n, 3), dtype=np.float32) * 4).astype(np.float32)
    for exposure in (1.0, 0.5, 2.5):
        ta, tb = capi.tonemap(rgb, exposure), oracle.tonemap(rgb, exposure)
        assert np.abs(ta.astype(np.int32) - tb.astype(np.int32)).max() <= 1    # exp/pow last-bit differences at a rounding boundary
        assert (ta != tb).mean() < 1e-3


def test_host_builder_statistics_and_guards():
    """The bin index is clamped where the reference would index out of its 8 bins (bvh.cpp:104-115)."""
    verts, idx = synth.soup(5000)
    b = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
    nodes, order, stats = b.nodes(), b.order(), b.stats()
    assert sorted(order.tolist()) == list(range(5000))
    assert int(nodes[0]["size"]) + int(stats[2]) == 5000          # root covers everything that was not dropped
    leaves = nodes[nodes["l"] == nodes["r"]]
    assert int(leaves["size"].max()) <= 4 or int(stats[2]) > 0
    b.close()
    with pytest.raises(capi.PtxError):
        capi.Bvh(verts=verts, indices=np.array([[0, 1, 999999]], dtype=np.uint32))    # index out of range


def _build_facade_check(tmp_path):
    exe = str(tmp_path / "facade_check")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "facade_check.cpp"), "-o", exe,
                           "-L" + libdir, "-lptx_amd", "-pthread", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def test_cpp_facade_host_side(tmp_path):
    """PT::BVH<Triangle> / PT::Ray of the C++ facade compile against the C ABI and behave like the reference's classes."""
    out = subprocess.check_output([_build_facade_check(tmp_path)], text=True)
    assert "host checks passed" in out


@pytest.mark.gpu
def test_cpp_facade_on_gpu(tmp_path):
    exe = _build_facade_check(tmp_path)
    out = subprocess.check_output([exe, "gpu"], text=True)
    assert "gpu checks passed" in out
    # ptx_frame_reduce's path for films on other GPUs (staging films, one copy stream per part), forced on one GPU
    out = subprocess.check_output([exe, "gpu"], text=True, env=dict(os.environ, PTX_FRAME_REDUCE_STAGE="1"))
    assert "gpu checks passed" in out


def test_cli_reports_errors_like_the_reference(tmp_path):
    """Exit code 1 and the reference's messages for a missing scene / unknown option (main.cpp:96-134)."""
    cli = os.path.join(ROOT, "pathtracer_amd", "bin", "ptx_render")
    assert os.path.exists(cli), "build it with make -C pathtracer_amd/csrc"
    r = subprocess.run([cli, "--trace"], capture_output=True, text=True)
    assert r.returncode == 1 and "must specify a scene file" in r.stderr
    r = subprocess.run([cli, "--trace", "-s", str(tmp_path / "none.js3d"), "-c", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to load scene" in r.stderr
    r = subprocess.run([cli, "-s", "x"], capture_output=True, text=True)
    assert r.returncode == 1
