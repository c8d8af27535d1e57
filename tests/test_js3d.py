"""The scene-file reader (ptx_js3d_*): a scene written by hand in the test, the committed scene
fixtures, and -- where the reference tree and its build are present -- the reference's own loader
(scene/io.cpp:15-70, util/to_json.cpp:301-502, geometry/indexed.cpp:10-34) on every bundled scene."""
import base64
import json
import os

import numpy as np
import pytest

from pathtracer_amd import abi, capi

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MEDIA = "/root/reference/media/js3d"


def blob(prefix, arr):
    return prefix + base64.b64encode(np.ascontiguousarray(arr).tobytes()).decode().rstrip("=")   # the format carries no padding


def bool_blob(bits):
    """Packed MSB first, then a 1 bit and zeros to the byte boundary (util/to_json.cpp:588-614)."""
    bits = list(bits)
    if bits:
        bits = bits + [1]
        bits += [0] * (-len(bits) % 8)
    packed = np.packbits(np.array(bits, dtype=np.uint8)) if bits else np.zeros(0, np.uint8)
    return blob("bool:", packed)


def halfedge_quad_and_triangle():
    """A quad (0,1,2,3) and a triangle (0,3,4) sharing edge 0-3, plus one boundary face around them.
    Half-edges come in twin pairs (i, i^1)."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [-1, 1, 0]], dtype=np.float32)
    # interior half-edges: quad 0->1 (h0), 1->2 (h2), 2->3 (h4), 3->0 (h6); triangle 0->3 (h7 = twin of h6), 3->4 (h8), 4->0 (h10)
    # boundary half-edges (twins): 1->0 (h1), 2->1 (h3), 3->2 (h5), 4->3 (h9), 0->4 (h11)
    nxt = np.zeros(12, dtype=np.uint32)
    nxt[0], nxt[2], nxt[4], nxt[6] = 2, 4, 6, 0
    nxt[7], nxt[8], nxt[10] = 8, 10, 7
    # boundary loop: 1->0 (h1) then 0->4 (h11) then 4->3 (h9) then 3->2 (h5) then 2->1 (h3) then back to h1
    nxt[1], nxt[11], nxt[9], nxt[5], nxt[3] = 11, 9, 5, 3, 1
    origin = {0: 0, 2: 1, 4: 2, 6: 3, 7: 0, 8: 3, 10: 4, 1: 1, 3: 2, 5: 3, 9: 4, 11: 0}
    vertex_halfedges = np.array([0, 2, 4, 6, 10], dtype=np.uint32)
    normals = np.zeros((12, 3), dtype=np.float32)
    normals[:, 2] = 1
    normals[7] = (0, 0.6, 0.8)
    uvs = np.array([[h / 16.0, 1 - h / 16.0] for h in range(12)], dtype=np.float32)
    face_halfedges = np.array([0, 7, 1], dtype=np.uint32)
    mesh = {
        "FORMAT": "s3d-hm-1",
        "halfedge_nexts": blob("uint32:", nxt), "halfedge_corner_uvs": blob("vec2:", uvs), "halfedge_corner_normals": blob("vec3:", normals),
        "bone_weight_bones": "uint32:", "bone_weight_weights": "float:",
        "vertex_halfedges": blob("uint32:", vertex_halfedges), "vertex_positions": blob("vec3:", pos),
        "vertex_bone_weight_ends": blob("uint32:", np.zeros(5, np.uint32)),
        "edge_sharps": bool_blob([0] * 6), "face_halfedges": blob("uint32:", face_halfedges), "face_boundaries": bool_blob([0, 0, 1]),
    }
    return mesh, pos, origin, normals, uvs


def tr(parent=None, t=(0, 0, 0), q=(0, 0, 0, 1), s=(1, 1, 1)):
    return {"parent": parent, "translation": list(t), "rotation": list(q), "scale": list(s)}


@pytest.fixture
def scene_file(tmp_path):
    mesh, pos, origin, normals, uvs = halfedge_quad_and_triangle()
    img = np.arange(2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3) / 18
    hdr = np.concatenate([np.frombuffer(b"rawf", np.uint8), np.array([3, 2], np.uint32).view(np.uint8), img.view(np.uint8).reshape(-1)])
    half = float(np.sqrt(0.5))
    scene = {
        "FORMAT": "js3d-v1",
        "transforms": {
            "root": tr(t=(1, 2, 3), q=(0, half, 0, half)),
            "child": tr(parent="root", t=(0, 0, -1), s=(2, 1, 0.5)),
            "cam_t": tr(t=(0, 1, 5)),
        },
        "cameras": {"Cam": {"vertical_fov": 45, "aspect_ratio": 1.5, "near_plane": 0.1, "film.width": 300, "film.height": 200,
                            "film.samples": 7, "film.max_ray_depth": 3, "film.sample_pattern": "Center"}},
        "meshes": {"M": mesh},
        "skinned_meshes": {}, "particles": {},
        "shapes": {"Ball": {"type": "Sphere", "radius": 0.75}},
        "textures": {"a_const": {"type": "Constant", "color": [0.1, 0.2, 0.3], "scale": 2},
                     "b_image": {"type": "Image", "sampler": "nearest", "image": blob("hdr64:", hdr)}},
        "materials": {"glass": {"type": "Glass", "transmittance": "a_const", "reflectance": "b_image", "ior": 1.25},
                      "lamb": {"type": "Lambertian", "albedo": "a_const"},
                      "lamp": {"type": "Emissive", "emissive": "b_image"}},
        "delta_lights": {"Spot": {"type": "Spot", "color": [1, 0.5, 0.25], "intensity": 3, "inner_angle": 20, "outer_angle": 50}},
        "env_lights": {"Sky": {"type": "Hemisphere", "intensity": 1, "radiance": "a_const"}},
        "instances": {
            "cameras": {"Cam_instance": {"transform": "cam_t", "camera": "Cam"}},
            "meshes": {"b_inst": {"transform": "child", "mesh": "M", "material": "glass", "visible": True, "collides": True, "style": "Correct"},
                       "a_hidden": {"transform": "root", "mesh": "M", "material": "lamb", "visible": False, "collides": True, "style": "Correct"},
                       "c_default_material": {"transform": "root", "mesh": "M", "material": None, "visible": True, "collides": True, "style": "Correct"}},
            "skinned_meshes": {}, "particles": {},
            "shapes": {"ball_inst": {"transform": "root", "shape": "Ball", "material": "lamp", "visible": True, "collides": True, "style": "Correct"}},
            "delta_lights": {"spot_inst": {"transform": "child", "light": "Spot", "visible": True}},
            "env_lights": {},
        },
    }
    p = tmp_path / "hand.js3d"
    p.write_text(json.dumps({"scene": scene, "animator": {"FORMAT": "js3d-v1", "splines": {}}}))
    return str(p), pos, origin, normals, uvs, img


def quat_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w],
                     [2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w],
                     [2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y]])


def test_hand_written_scene(scene_file):
    path, pos, origin, normals, uvs, img = scene_file
    j = capi.Js3d(path)
    d = j.desc
    assert j.mesh_names() == ["M"] and j.camera_names() == ["Cam_instance"]
    verts, idx = j.mesh_arrays(0)
    # quad -> 4 corners + fan of 2, triangle -> 3 corners + 1; the boundary face is skipped
    assert len(verts) == 7 and idx.tolist() == [[0, 1, 2], [0, 2, 3], [4, 5, 6]]
    for corner, h in enumerate([0, 2, 4, 6, 7, 8, 10]):
        assert (verts[corner, :3] == pos[origin[h]]).all()
        assert (verts[corner, 3:6] == normals[h]).all() and (verts[corner, 6:] == uvs[h]).all()
    # visible instances in name order, meshes before shapes
    assert j.instance_names() == ["b_inst", "c_default_material", "ball_inst"]
    inst = d.instances
    assert (inst[0].mesh, inst[1].mesh, inst[2].mesh) == (0, 0, -1)
    assert inst[1].material == -1 and inst[2].sphere_radius == 0.75
    mats = {i: (d.materials[i].type, d.materials[i].tex_a, d.materials[i].tex_b, d.materials[i].ior) for i in range(d.n_materials)}
    assert mats[inst[0].material] == (abi.PTX_MAT_GLASS, 1, 0, 1.25)
    assert mats[inst[2].material][0] == abi.PTX_MAT_EMISSIVE
    # transforms: child = root * local, column major
    half = np.sqrt(0.5)
    R = np.eye(4)
    R[:3, :3] = quat_matrix((0, half, 0, half))
    R[:3, 3] = (1, 2, 3)
    L = np.diag([2, 1, 0.5, 1.0])
    L[:3, 3] = (0, 0, -1)
    np.testing.assert_allclose(np.array(inst[1].T[:]).reshape(4, 4).T, R, atol=1e-6)
    np.testing.assert_allclose(np.array(inst[0].T[:]).reshape(4, 4).T, R @ L, atol=1e-6)
    # textures, lights, camera
    t = d.textures
    assert (t[0].type, t[0].scale) == (abi.PTX_TEX_CONSTANT, 2.0) and abs(t[0].color[1] - 0.2) < 1e-7
    assert (t[1].type, t[1].sampler, t[1].w, t[1].h) == (abi.PTX_TEX_IMAGE, abi.PTX_SAMPLER_NEAREST, 3, 2)
    assert (np.ctypeslib.as_array(t[1].pixels, shape=(2, 3, 3)) == img).all()
    assert d.n_delta_lights == 1 and d.delta_lights[0].type == abi.PTX_LIGHT_SPOT and d.delta_lights[0].outer_angle == 50
    np.testing.assert_allclose(np.array(d.delta_lights[0].T[:]).reshape(4, 4).T, R @ L, atol=1e-6)
    assert d.n_env_lights == 1 and d.env_lights[0].type == abi.PTX_ENV_HEMISPHERE and d.env_lights[0].tex == 0
    cam = j.camera("Cam_instance")
    assert (cam.vertical_fov, cam.aspect_ratio, cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth) == (45, 1.5, 300, 200, 7, 3)
    assert list(cam.camera_to_world[12:15]) == [0, 1, 5]
    with pytest.raises(capi.PtxError):
        j.camera("nope")
    # --no_bvh reaches every mesh and the instance level
    j.set_modes(False, abi.PTX_FRESNEL_FLOAT_ABS)
    assert j.desc.use_bvh == 0 and j.desc.meshes[0].use_bvh == 0 and j.desc.fresnel_mode == abi.PTX_FRESNEL_FLOAT_ABS
    j.close()


def test_reader_errors(tmp_path):
    with pytest.raises(capi.PtxError):
        capi.Js3d(str(tmp_path / "missing.js3d"))
    bad = tmp_path / "bad.js3d"
    bad.write_text('{"scene": {"meshes": {"M": {"halfedge_nexts": "uint32:AAAA"}}}}')     # 3 bytes: not a multiple of 4
    with pytest.raises(capi.PtxError):
        capi.Js3d(str(bad))
    bad.write_text('{"scene": ')
    with pytest.raises(capi.PtxError):
        capi.Js3d(str(bad))
    bad.write_text('{"animator": {}}')
    with pytest.raises(capi.PtxError):
        capi.Js3d(str(bad))


def test_png_writer(tmp_path):
    import zlib
    w, h = 5, 3
    rgba = (np.arange(w * h * 4) * 7 % 256).astype(np.uint8).reshape(h, w, 4)
    p = tmp_path / "o.png"
    capi.write_png(str(p), rgba, w, h, flip_rows=True)
    data = p.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(data):
        n = int.from_bytes(data[pos:pos + 4], "big")
        kind, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        assert zlib.crc32(kind + body) == int.from_bytes(data[pos + 8 + n:pos + 12 + n], "big")
        chunks[kind] = body
        pos += 12 + n
    assert chunks[b"IHDR"] == w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([8, 6, 0, 0, 0])
    raw = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), np.uint8).reshape(h, 1 + w * 4)
    assert (raw[:, 0] == 0).all()
    assert (raw[:, 1:].reshape(h, w, 4) == rgba[::-1]).all()      # image row 0 is the bottom row


@pytest.mark.skipif(not os.path.isdir(MEDIA), reason="reference tree not present")
@pytest.mark.parametrize("name", ["cbox", "refract_cow_cbox"])
def test_scene_fixture_is_what_the_reader_produces(name):
    j = capi.Js3d(f"{MEDIA}/{name}.js3d")
    a = abi.desc_to_arrays(j.desc, {n: j.camera(n) for n in j.camera_names()})
    g = np.load(os.path.join(G, f"scene_{name}.npz"), allow_pickle=False)
    assert set(a) == set(g.files)
    for k in a:
        assert np.array_equal(a[k], g[k]), k


@pytest.mark.skipif(not os.path.isdir(MEDIA), reason="reference tree not present")
def test_reader_equals_reference_loader_on_bundled_scenes(ref):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    for f in ("cbox.js3d", "refract_cow_cbox.js3d", "A3-cbox-spheres.js3d", "A1T6-samplers.js3d", "dodecahedron-lit.js3d", "cow-simple.js3d"):
        mg.check_reader_against_reference(ref, f"{MEDIA}/{f}").close()


def test_fixture_round_trip():
    g = np.load(os.path.join(G, "scene_cbox.npz"), allow_pickle=False)
    sb, cams = abi.builder_from_arrays(g)
    assert "Camera_instance" in cams
    a = abi.desc_to_arrays(sb.desc(), cams)
    for k in g.files:
        assert np.array_equal(a[k], g[k]), k


@pytest.mark.gpu
def test_cli_renders_a_scene_file(scene_file, tmp_path):
    """ptx_render with the reference's flags (main.cpp:55-71): scene file in, tone-mapped PNG out."""
    import subprocess
    import zlib
    path = scene_file[0]
    cli = os.path.join(os.path.dirname(G), "..", "pathtracer_amd", "bin", "ptx_render")
    out = tmp_path / "frame.png"
    r = subprocess.run([cli, "--trace", "-s", path, "-c", "Cam_instance", "-o", str(out), "--film-width", "64", "--film-height", "48",
                        "--film-samples", "8", "--film-max-ray-depth", "3", "--seed", "4660", "--exposure", "1.5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Set film size to [64x48]" in r.stdout and "Wrote result to" in r.stdout
    data = out.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    assert int.from_bytes(data[16:20], "big") == 64 and int.from_bytes(data[20:24], "big") == 48
    n = int.from_bytes(data[33:37], "big")
    assert data[37:41] == b"IDAT"
    raw = np.frombuffer(zlib.decompress(data[41:41 + n]), np.uint8).reshape(48, 1 + 64 * 4)[:, 1:].reshape(48, 64, 4)
    assert (raw[..., 3] == 255).all()
    assert raw[..., :3].max() > 0            # the hemisphere light is visible above the horizon
    # same flags, --no_bvh: List<Instance> / List<Triangle> give the same picture (same random stream)
    out2 = tmp_path / "frame_list.png"
    r = subprocess.run([cli, "--trace", "-s", path, "-c", "Cam_instance", "-o", str(out2), "--film-width", "64", "--film-height", "48",
                        "--film-samples", "8", "--film-max-ray-depth", "3", "--seed", "4660", "--exposure", "1.5", "--no_bvh"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    raw2 = np.frombuffer(zlib.decompress(out2.read_bytes()[41:41 + n]), np.uint8).reshape(48, 1 + 64 * 4)[:, 1:].reshape(48, 64, 4)
    assert (np.abs(raw.astype(int) - raw2.astype(int)) > 1).mean() < 0.01
