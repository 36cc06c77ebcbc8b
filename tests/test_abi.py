"""The C-ABI shared library on a machine without a GPU: it loads, exports every symbol the public header
declares, the ctypes mirrors match the compiled struct sizes, and the host-only entry points behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raymarcher_amd import abi, lib
from raymarcher_amd._lib import LIB_PATH, SIGNATURES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "raymarcher_amd.h")).read()


def declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(rm_[a-z0-9_]+)\s*\(", body)))


def test_library_exports_every_declared_symbol():
    L = C.CDLL(LIB_PATH)
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/raymarcher_amd.h but not exported"
    assert set(names) == set(SIGNATURES), set(names) ^ set(SIGNATURES)


def test_struct_sizes_match_the_compiled_library():
    L = lib()
    structs = [abi.RmObject, abi.RmLight, abi.RmCamera, abi.RmGlobals, abi.RmSettings, abi.RmCounters,
               abi.RmHostSettings, abi.RmCameraData, abi.RmTexture, abi.RmPostSettings]
    for i, s in enumerate(structs):
        assert L.rm_abi_sizeof(i) == C.sizeof(s), s.__name__
    assert L.rm_abi_sizeof(99) == -1
    assert L.rm_abi_version() == abi.RM_ABI_VERSION
    # sizes quoted in SURVEY §8a: RayMarchObject 176 B, LightSource 116 B
    assert C.sizeof(abi.RmObject) == 176 and C.sizeof(abi.RmLight) == 116


def test_defaults_and_status_strings():
    L = lib()
    s = abi.RmSettings()
    L.rm_settings_default(C.byref(s))
    assert (s.maxSteps, s.fractalIters, s.mengerLevels, s.numReflection) == (256, 20, 4, 1)
    assert s.features == abi.RM_FEAT_REFERENCE_DEFAULT and s.enableSoftShadow == 0
    d = abi.default_settings()
    assert bytes(d) == bytes(s)
    hs = abi.RmHostSettings()
    L.rm_host_settings_default(C.byref(hs))
    assert (hs.screenWidth, hs.screenHeight, hs.power) == (1024, 768, 8.0)
    assert abs(hs.nearPlane - 0.1) < 1e-7 and hs.farPlane == 100.0
    assert L.rm_status_string(0) == b"RM_OK" and L.rm_status_string(3) == b"RM_ERR_UNSUPPORTED"
    assert L.rm_status_string(12345) == b"RM_ERR_UNKNOWN"


def test_enum_values_follow_the_reference_order():
    # scenedata.h:18-33 == frag:53-68, scenedata.h:10-15 == frag:72-75
    for name, val in (("RM_CUBE", 0), ("RM_SPHERE", 3), ("RM_RECTANGLE", 8), ("RM_MANDELBULB", 10), ("RM_CUSTOM", 13),
                      ("RM_LIGHT_POINT", 0), ("RM_LIGHT_DIRECTIONAL", 1), ("RM_LIGHT_SPOT", 2), ("RM_LIGHT_AREA", 3)):
        assert getattr(abi, name) == val
        assert re.search(rf"\b{name}\s*=\s*{val}\b", HEADER)
    assert abi.RM_MAX_OBJECTS == 30 and abi.RM_MAX_LIGHTS == 10


@pytest.mark.parametrize("H,T,N", [(2160, 8, 1), (2160, 8, 8), (2160, 8, 4), (50, 8, 3), (7, 8, 4), (64, 16, 5), (4320, 8, 8)])
def test_row_tile_partition_is_a_bijection(H, T, N):
    L = lib()
    seen = []
    counts = []
    for k in range(N):
        n = L.rm_shard_rows(H, T, k, N)
        counts.append(n)
        rows = [L.rm_shard_row_to_frame(H, T, k, N, r) for r in range(n)]
        assert rows == sorted(rows)
        for r in rows:
            assert (r // T) % N == k
        seen += rows
        assert L.rm_shard_row_to_frame(H, T, k, N, n) == -1
    assert sorted(seen) == list(range(H))
    assert counts[0] == max(counts)  # equal-size gather slots are sized by shard 0
    assert max(counts) - min(counts) <= T
    assert L.rm_shard_rows(H, T, N, N) == -1 and L.rm_shard_rows(H, 0, 0, N) == -1


def test_png_writer_roundtrip(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (13, 17, 4), dtype=np.uint8)
    p = tmp_path / "x.png"
    assert lib().rm_write_png(str(p).encode(), img.ctypes.data_as(C.c_void_p), 17, 13) == 0
    assert (np.asarray(Image.open(p)) == img).all()
    assert lib().rm_write_png(b"/nonexistent_dir/x.png", img.ctypes.data_as(C.c_void_p), 17, 13) == abi.RM_ERR_IO


@pytest.mark.parametrize("mode,bits", [("RGB", 8), ("RGBA", 8), ("L", 8), ("LA", 8), ("P", 8), ("I;16", 16), ("1", 1)])
def test_png_reader_matches_pillow(tmp_path, mode, bits):
    from PIL import Image
    from raymarcher_amd.render import load_image
    rng = np.random.default_rng(5)
    W, H = 29, 17
    if mode == "P":
        im = Image.fromarray(rng.integers(0, 256, (H, W), dtype=np.uint8), "P")
        im.putpalette(list(rng.integers(0, 256, 768)))
    elif mode == "I;16":
        im = Image.fromarray(rng.integers(0, 65536, (H, W)).astype(np.uint16))
    elif mode == "1":
        im = Image.fromarray(rng.integers(0, 2, (H, W)).astype(bool))
    else:
        ch = {"RGB": 3, "RGBA": 4, "L": 1, "LA": 2}[mode]
        a = rng.integers(0, 256, (H, W, ch), dtype=np.uint8)
        im = Image.fromarray(a[..., 0] if ch == 1 else a, mode)
    p = tmp_path / f"t_{bits}.png"
    im.save(p)
    got = load_image(p, flip_vertical=False)
    if mode == "I;16":
        exp16 = np.asarray(Image.open(p)).astype(np.uint16)
        exp = np.stack([exp16 >> 8] * 3 + [np.full_like(exp16, 255)], -1).astype(np.uint8)
    else:
        exp = np.asarray(Image.open(p).convert("RGBA"))
    assert got.shape == (H, W, 4) and (got == exp).all()
    assert (load_image(p, flip_vertical=True) == exp[::-1]).all()


def test_png_reader_errors(tmp_path):
    from raymarcher_amd import RaymarcherError
    from raymarcher_amd.render import load_image
    (tmp_path / "x.bmp").write_bytes(b"BM" + b"0" * 64)
    with pytest.raises(RaymarcherError) as e:
        load_image(tmp_path / "x.bmp")
    assert e.value.status == abi.RM_ERR_UNSUPPORTED
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff\xe0" + b"0" * 64)
    with pytest.raises(RaymarcherError) as e:
        load_image(tmp_path / "x.jpg")
    assert e.value.status == abi.RM_ERR_PARSE
    with pytest.raises(RaymarcherError) as e:
        load_image(tmp_path / "missing.png")
    assert e.value.status == abi.RM_ERR_IO


def test_no_cpu_fallback_in_the_product_path():
    """The renderer refuses to run without a HIP device, and the product sources never reach into oracle/."""
    import torch
    from raymarcher_amd import Renderer
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            Renderer(0)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raymarcher_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "oracle/" not in text, f"{fn} mentions oracle/"
                assert "rm_oracle" not in text and "librm_oracle" not in text, fn


def test_skybox_face_paths():
    """getCubeMapWithType's lists (raymarchscene.cpp:50-86), including the NIGHTSKY order as written there."""
    from raymarcher_amd import lib
    L = lib()
    assert L.rm_skybox_face_path(1, 0) == b"texture_store/cube_map/beach/+x.jpg"
    assert [L.rm_skybox_face_path(2, f).decode()[-6:-4] for f in range(6)] == ["-x", "+x", "-y", "+y", "+z", "-z"]
    assert [L.rm_skybox_face_path(3, f).decode()[-6:-4] for f in range(6)] == ["+x", "-x", "+y", "-y", "+z", "-z"]
    assert L.rm_skybox_face_path(0, 0) is None and L.rm_skybox_face_path(4, 0) is None and L.rm_skybox_face_path(1, 6) is None


@pytest.mark.parametrize("size,subsampling,quality,mode", [
    ((64, 64), 0, 90, "RGB"), ((37, 23), 0, 75, "RGB"), ((64, 48), 1, 85, "RGB"), ((37, 23), 1, 60, "RGB"),
    ((64, 64), 2, 90, "RGB"), ((37, 23), 2, 75, "RGB"), ((259, 380), 2, 92, "RGB"), ((512, 512), 2, 80, "RGB"),
    ((1, 1), 2, 75, "RGB"), ((3, 5), 2, 75, "RGB"), ((5, 3), 1, 75, "RGB"), ((40, 31), 0, 75, "L"), ((17, 1), 2, 30, "RGB")])
def test_jpeg_reader_matches_libjpeg(tmp_path, size, subsampling, quality, mode):
    """rm_image_load on baseline JPEG files (the reference's "Beach" sky box and one texture are JPEGs read by QImage →
    libjpeg): the pixels equal libjpeg-turbo's (Pillow) value for value — islow IDCT, fancy upsampling of 4:2:2 / 4:2:0
    chroma (box replication for planes at most two samples wide), 16-bit fixed-point YCbCr → RGB."""
    from PIL import Image
    from raymarcher_amd.render import load_image
    W, H = size
    rng = np.random.default_rng(W * 1000 + H + subsampling)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 // max(W - 1, 1)), (yy * 255 // max(H - 1, 1)), ((xx * 5 + yy * 9) % 256)], -1).astype(np.int32)
    img = np.clip(img + rng.integers(-40, 41, img.shape), 0, 255).astype(np.uint8)  # noise exercises every AC coefficient
    im = Image.fromarray(img, "RGB").convert(mode)
    path = tmp_path / "t.jpg"
    kw = {} if mode == "L" else {"subsampling": subsampling}
    im.save(path, "JPEG", quality=quality, **kw)
    exp = np.asarray(Image.open(path).convert("RGBA"))
    got = load_image(path, flip_vertical=False)
    assert got.shape == exp.shape
    assert (got == exp).all(), f"max |Δ| {np.abs(got.astype(int) - exp.astype(int)).max()} on {(got != exp).any(-1).mean():.3%} of the pixels"
    assert (load_image(path, flip_vertical=True) == exp[::-1]).all()


def test_jpeg_reader_restart_markers_and_refusals(tmp_path):
    from PIL import Image
    from raymarcher_amd import RaymarcherError
    from raymarcher_amd.render import load_image
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (50, 70, 3), dtype=np.uint8)
    path = tmp_path / "r.jpg"
    try:
        Image.fromarray(img).save(path, "JPEG", quality=80, subsampling=2, restart_marker_blocks=3)
    except TypeError:
        pytest.skip("this Pillow cannot write restart markers")
    assert b"\xff\xdd" in path.read_bytes()
    assert (load_image(path, flip_vertical=False) == np.asarray(Image.open(path).convert("RGBA"))).all()
    prog = tmp_path / "p.jpg"
    Image.fromarray(img).save(prog, "JPEG", progressive=True)
    with pytest.raises(RaymarcherError) as e:
        load_image(prog)
    assert e.value.status == abi.RM_ERR_UNSUPPORTED
    # crafted headers: a frame header that lists one component id twice, and a scan that names one id twice (leaving
    # another component without Huffman tables) are refused, not decoded with unset tables
    base = tmp_path / "b.jpg"
    Image.fromarray(img).save(base, "JPEG", quality=80, subsampling=0)
    data = bytearray(base.read_bytes())
    sof = data.index(b"\xff\xc0")
    ids = [data[sof + 10 + 3 * k] for k in range(3)]
    bad = bytearray(data)
    bad[sof + 10 + 3] = ids[0]  # second component takes the first one's id
    (tmp_path / "dup_sof.jpg").write_bytes(bad)
    sos = data.index(b"\xff\xda")
    bad2 = bytearray(data)
    bad2[sos + 5 + 2] = ids[0]  # second scan component names the first id again
    (tmp_path / "dup_sos.jpg").write_bytes(bad2)
    for name in ("dup_sof.jpg", "dup_sos.jpg"):
        with pytest.raises(RaymarcherError) as e:
            load_image(tmp_path / name)
        assert e.value.status == abi.RM_ERR_PARSE, name


@pytest.mark.parametrize("size,colors,interlace,transparent", [((61, 30), 256, False, False), ((33, 47), 16, False, False),
                                                              ((64, 64), 2, False, False), ((50, 41), 200, True, False),
                                                              ((40, 40), 64, False, True), ((361, 300), 256, False, False)])
def test_gif_reader_matches_pillow(tmp_path, size, colors, interlace, transparent):
    """First frame of a GIF (one reference texture is a single-frame GIF): LZW, colour tables, interlace, transparency."""
    from PIL import Image
    from raymarcher_amd.render import load_image
    W, H = size
    rng = np.random.default_rng(W + H + colors)
    yy, xx = np.mgrid[0:H, 0:W]
    idx = ((xx // 3 + yy // 2 + rng.integers(0, 3, (H, W))) % colors).astype(np.uint8)
    im = Image.fromarray(idx, "P")
    pal = rng.integers(0, 256, 768, dtype=np.uint8)
    im.putpalette(list(pal))
    path = tmp_path / "t.gif"
    kw = {"interlace": interlace}
    if transparent:
        kw["transparency"] = 3
    im.save(path, "GIF", **kw)
    exp = np.asarray(Image.open(path).convert("RGBA")).copy()
    got = load_image(path, flip_vertical=False)
    assert got.shape == exp.shape
    opaque = exp[..., 3] == 255
    assert (got[opaque] == exp[opaque]).all()
    assert (got[~opaque][:, 3] == 0).all() and (~opaque).any() == transparent


def test_header_is_plain_c_and_links(tmp_path):
    """include/raymarcher_amd.h must be usable from C (the reference's maintainers bind it from C++/C):
    compile a C99 program with gcc -pedantic against the header, link the shared library, run the host-side calls."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "use_abi.c"
    scene = os.path.join(ROOT, "tests", "golden", "scenes", "simple", "unit_mandelbulb.json")
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "raymarcher_amd.h"
int main(int argc, char **argv) {
  RmScene *sc = NULL;
  RmCamera cam;
  RmCameraData cd;
  RmSettings s;
  RmResources res;
  float view[16], proj[16];
  memset(&res, 0, sizeof res);
  if (rm_abi_version() != RM_ABI_VERSION) return 2;
  if (rm_abi_sizeof(0) != (int)sizeof(RmObject) || rm_abi_sizeof(10) != (int)sizeof(RmResources)) return 3;
  rm_settings_default(&s);
  if (s.maxSteps != 256 || s.fractalIters != 20) return 4;
  if (rm_scene_load(argv[1], &sc) != RM_OK) { printf("%s\n", rm_last_error()); return 5; }
  if (rm_scene_num_objects(sc) != 1 || rm_scene_objects(sc)[0].type != RM_MANDELBULB || rm_scene_num_lights(sc) != 3) return 6;
  if (rm_scene_camera_data(sc, &cd) != RM_OK || rm_camera_build(&cd, 3840, 2160, 0.1f, 100.0f, view, proj, &cam) != RM_OK) return 7;
  if (rm_shard_rows(2160, 8, 0, 8) != 272 || rm_shard_row_to_frame(2160, 8, 3, 8, 9) != 89) return 8;
  if (rm_scene_load("/nonexistent.json", &sc) == RM_OK) return 9;
  printf("ok %s %.3f\n", rm_status_string(RM_ERR_UNSUPPORTED), cam.invProjView[0]);
  return 0;
}
''')
    exe = tmp_path / "use_abi"
    libdir = os.path.join(ROOT, "raymarcher_amd", "lib")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-lraymarcher_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    r = subprocess.run([str(exe), scene], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok RM_ERR_UNSUPPORTED"), (r.returncode, r.stdout, r.stderr[-500:])


def test_gather_slot_arithmetic():
    """rm_gather_slot_rows (the equal slot of the C-ABI gather) holds every shard of the rm_render_tiles partition, the
    shards partition the frame's rows, and rm_shard_row_to_frame inverts the packing — for ragged sizes too."""
    L = lib()
    for H, T, N in ((2160, 8, 8), (2160, 8, 1), (4320, 8, 8), (54, 8, 3), (7, 8, 4), (1080, 16, 5), (33, 1, 64)):
        slot = L.rm_gather_slot_rows(H, T, N)
        rows = [L.rm_shard_rows(H, T, k, N) for k in range(N)]
        assert sum(rows) == H and slot == rows[0] == max(rows)
        seen = sorted(L.rm_shard_row_to_frame(H, T, k, N, i) for k in range(N) for i in range(rows[k]))
        assert seen == list(range(H))
    assert L.rm_gather_slot_rows(0, 8, 2) == 0 and L.rm_gather_slot_rows(10, 0, 2) == 0
