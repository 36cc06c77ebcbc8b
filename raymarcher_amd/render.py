"""Host-side driver of the C-ABI: scene tables, camera, row-range / row-tile renders into torch tensors.

Mirrors the reference's scene surface — RayMarchScene::initScene/getShapes/getLights/getCamera/
getGlobalData (src/raymarch/raymarchscene.h:12-90) and Realtime::rayMarch / saveViewportImage
(src/realtimerender.cpp:53-87, src/realtime.cpp:284-350) — on top of rm_scene_*, rm_camera_build,
rm_render*, rm_frame_to_rgba8 and rm_write_png.
"""
import ctypes as C
from dataclasses import dataclass

from . import abi
from ._lib import RaymarcherError, check, lib


@dataclass
class SceneTables:
    """The uniform tables one frame needs (what configure*Uniforms upload, realtimerender.cpp:596-811)."""
    camera: abi.RmCamera
    objects: C.Array
    num_objects: int
    lights: C.Array
    num_lights: int
    globals_: abi.RmGlobals
    textures: list = None  # host uint8 arrays (H, W, 4), rows bottom-up, indexed by RmObject.texLoc
    noise: object = None   # uint8 (H, W, 4): the `noise` sampler of NIGHTSKY_BACKGROUND / SEA (noise_texture_1.png, mirrored)
    skybox: list = None    # six uint8 (H, W, 4) cube-map faces +X,-X,+Y,-Y,+Z,-Z as uploaded (mirrored at load)
    ltc1: object = None    # uint8 (64, 64, 4) LTC tables of the area lights (ltc_quantise of the float tables)
    ltc2: object = None

    def args(self, settings):
        return (C.byref(self.camera), self.objects, self.num_objects, self.lights, self.num_lights,
                C.byref(self.globals_), C.byref(settings))


def load_image(path, flip_vertical=True):
    """rm_image_load → numpy uint8 (H, W, 4); flip_vertical=True gives the bottom-up rows the renderer samples
    (QImage::mirrored at load, raymarchscene.cpp:208)."""
    import numpy as np
    px, w, h = C.c_void_p(), C.c_int(), C.c_int()
    check(lib().rm_image_load(str(path).encode(), 1 if flip_vertical else 0, C.byref(px), C.byref(w), C.byref(h)))
    try:
        arr = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 4)).copy()
    finally:
        lib().rm_image_free(px)
    return arr


def ltc_quantise(table):
    """Float RGBA table → the 8-bit texels the reference's glTexImage2D(GL_RGBA, …, GL_FLOAT, LTC) upload leaves
    (realtimerender.cpp:908, 925)."""
    import numpy as np
    t = np.ascontiguousarray(table, dtype=np.float32).reshape(-1, 4)
    out = np.empty(t.shape, dtype=np.uint8)
    lib().rm_ltc_quantise(C.c_void_p(t.ctypes.data), C.c_void_p(out.ctypes.data), t.shape[0])
    return out.reshape(np.shape(table))


def build_camera(pos, look, up, height_angle_rad, W, H, near=0.1, far=100.0):
    """Camera::initializeCamera + configureCameraUniforms via rm_camera_build (camera.cpp:8-133)."""
    cd = abi.RmCameraData()
    for i in range(3):
        cd.pos[i], cd.look[i], cd.up[i] = pos[i], look[i], up[i]
    cd.pos[3], cd.look[3], cd.up[3] = 1.0, 0.0, 0.0
    cd.heightAngle = height_angle_rad
    cam = abi.RmCamera()
    view = (C.c_float * 16)()
    proj = (C.c_float * 16)()
    check(lib().rm_camera_build(C.byref(cd), W, H, near, far, view, proj, C.byref(cam)))
    return cam, list(view), list(proj)


class Scene:
    """A parsed scenefile (SceneParser::parse + RayMarchScene::initScene)."""

    def __init__(self, path=None, text=None):
        self._h = C.c_void_p()
        L = lib()
        if path is not None:
            check(L.rm_scene_load(str(path).encode(), C.byref(self._h)))
        elif text is not None:
            check(L.rm_scene_load_string(text.encode(), C.byref(self._h)))
        else:
            raise ValueError("path or text required")

    def close(self):
        if self._h:
            lib().rm_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_objects(self):
        return lib().rm_scene_num_objects(self._h)

    @property
    def num_lights(self):
        return lib().rm_scene_num_lights(self._h)

    def camera_data(self):
        cd = abi.RmCameraData()
        check(lib().rm_scene_camera_data(self._h, C.byref(cd)))
        return cd

    def texture_of(self, i):
        t = lib().rm_scene_object_texture(self._h, i)
        return t.decode() if t else None

    def tables(self, W, H, near=0.1, far=100.0, host_settings=None, load_textures=True):
        """Copy the tables out (so they outlive the handle) and build the camera for a W×H frame.
        load_textures=False leaves `textures` empty for the caller to fill (slot i = RmObject.texLoc i)."""
        L = lib()
        no, nl = self.num_objects, self.num_lights
        objs = (abi.RmObject * max(no, 1))()
        lights = (abi.RmLight * max(nl, 1))()
        po, pl = L.rm_scene_objects(self._h), L.rm_scene_lights(self._h)
        for i in range(no):
            C.memmove(C.byref(objs[i]), C.byref(po[i]), C.sizeof(abi.RmObject))
        for i in range(nl):
            C.memmove(C.byref(lights[i]), C.byref(pl[i]), C.sizeof(abi.RmLight))
        g = abi.RmGlobals()
        hs = host_settings
        check(L.rm_scene_globals(self._h, C.byref(hs) if hs is not None else None, C.byref(g)))
        cd = self.camera_data()
        cam = abi.RmCamera()
        check(L.rm_camera_build(C.byref(cd), W, H, near, far, None, None, C.byref(cam)))
        # texture slots in texLoc order (configureShapesUniforms binds them in first-use order, realtimerender.cpp:735-806)
        textures = {}
        for i in range(no if load_textures else 0):
            if objs[i].texLoc >= 0 and objs[i].texLoc not in textures:
                try:
                    textures[objs[i].texLoc] = load_image(self.texture_of(i), flip_vertical=True)
                except RaymarcherError as e:
                    if e.status != abi.RM_ERR_IO:
                        raise
                    # a texture file that is not there: the reference prints "Failed to load in image", still creates the GL
                    # texture (initShapesTextures, realtimerender.cpp:266-303) and samples the incomplete texture, which
                    # reads (0,0,0,1) — a 1×1 black texel gives the same samples
                    import numpy as np
                    textures[objs[i].texLoc] = np.array([[[0, 0, 0, 255]]], dtype=np.uint8)
        tex_list = [textures[k] for k in sorted(textures)] if textures else None
        return SceneTables(cam, objs, no, lights, nl, g, tex_list)


class Renderer:
    """Launches the HIP raymarch on one GPU; outputs are torch tensors on that device."""

    def __init__(self, device=0):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("raymarcher_amd.Renderer needs a HIP device; there is no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        check(lib().rm_set_device(device))

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _upload(self, a):
        """Host uint8 array → device tensor, cached per array object."""
        if not hasattr(self, "_tex_cache"):
            self._tex_cache = {}
        key = id(a)
        if key not in self._tex_cache:
            self._tex_cache[key] = (a, self.torch.from_numpy(a).contiguous().to(self.device))
        return self._tex_cache[key][1]

    def _resources(self, tables):
        """RmResources with device pointers for everything `tables` carries; returns (struct, keep-alive list)."""
        res = abi.RmResources()
        keep = []

        def fill(slot, a):
            dev = self._upload(a)
            keep.append(dev)
            slot.pixels = dev.data_ptr()
            slot.height, slot.width = a.shape[0], a.shape[1]

        host = tables.textures or []
        if host:
            arr = (abi.RmTexture * len(host))()
            for i, a in enumerate(host):
                fill(arr[i], a)
            keep.append(arr)
            res.textures = arr
            res.numTextures = len(host)
        if tables.noise is not None:
            fill(res.noise, tables.noise)
        if tables.skybox:
            if len(tables.skybox) != 6:
                raise ValueError("skybox needs six faces (+X,-X,+Y,-Y,+Z,-Z)")
            for f in range(6):
                fill(res.skybox[f], tables.skybox[f])
        for name in ("ltc1", "ltc2"):
            a = getattr(tables, name)
            if a is not None:
                if a.shape != (abi.RM_LTC_SIZE, abi.RM_LTC_SIZE, 4) or str(a.dtype) != "uint8":
                    raise ValueError(f"{name} must be uint8 (64, 64, 4); see ltc_quantise")
                dev = self._upload(a)
                keep.append(dev)
                setattr(res, name, dev.data_ptr())
        return res, keep

    def render(self, tables, settings, W, H, row_begin=0, row_end=None, bright=False, out=None):
        """rm_render: rows [row_begin,row_end) → float32 tensor (rows, W, 4), row 0 = bottom."""
        t = self.torch
        row_end = H if row_end is None else row_end
        n = row_end - row_begin
        if out is None:
            out = t.empty((max(n, 0), W, 4), dtype=t.float32, device=self.device)
        br = t.empty_like(out) if bright else None
        res, _keep = self._resources(tables)
        check(lib().rm_render_res(*tables.args(settings), C.byref(res), W, H, row_begin, row_end, C.c_void_p(out.data_ptr()),
                                  C.c_void_p(br.data_ptr()) if bright else None, self._stream()))
        return (out, br) if bright else out

    def render_counted(self, tables, settings, W, H, mode=abi.RM_COUNT_REFERENCE):
        """rm_render_counted_res: the frame plus its work counters — the reference's work (mode RM_COUNT_REFERENCE) or
        what the production kernel really executes (RM_COUNT_EXECUTED; plain scene classes only)."""
        t = self.torch
        out = t.empty((H, W, 4), dtype=t.float32, device=self.device)
        cnt = abi.RmCounters()
        res, _keep = self._resources(tables)
        t.cuda.synchronize(self.device)
        check(lib().rm_render_counted_res(*tables.args(settings), C.byref(res), W, H, 0, H, C.c_void_p(out.data_ptr()), None, mode,
                                          C.byref(cnt)))
        return out, cnt

    def render_clocked(self, tables, settings, W, H, wave_spans=False):
        """rm_render_clocked: the frame from the stamped diagnostic build and the shader clock (MHz) it ran at; with
        wave_spans=True also an int64 tensor (waves, 2) of every wave's first / last 100 MHz tick."""
        t = self.torch
        out = t.empty((H, W, 4), dtype=t.float32, device=self.device)
        mhz = C.c_double()
        # one (first, last) pair per wave; the kernel indexes tile·(waves per workgroup) + wave with tiles = ceil(W / (8·wpb)) per
        # row, so a row holds at most ceil(W/8) + 3 waves for any RM_WAVES_PER_BLOCK in {1, 2, 4}
        spans = t.zeros(((((W + 7) // 8) + 3) * ((H + 7) // 8), 2), dtype=t.int64, device=self.device) if wave_spans else None
        t.cuda.synchronize(self.device)
        check(lib().rm_render_clocked(*tables.args(settings), W, H, C.c_void_p(out.data_ptr()), C.byref(mhz),
                                      C.c_void_p(spans.data_ptr()) if wave_spans else None))
        return (out, mhz.value, spans) if wave_spans else (out, mhz.value)

    def render_tiles(self, tables, settings, W, H, tile_rows, shard, num_shards, out=None):
        """rm_render_tiles: this shard's interleaved row tiles, packed → (rm_shard_rows, W, 4)."""
        t = self.torch
        n = lib().rm_shard_rows(H, tile_rows, shard, num_shards)
        if out is None:
            out = t.empty((max(n, 0), W, 4), dtype=t.float32, device=self.device)
        res, _keep = self._resources(tables)
        check(lib().rm_render_tiles_res(*tables.args(settings), C.byref(res), W, H, tile_rows, shard, num_shards,
                                        C.c_void_p(out.data_ptr()), None, self._stream()))
        return out

    def deinterleave(self, gathered, W, H, tile_rows, num_shards, shard_stride_rows=0):
        t = self.torch
        frame = t.empty((H, W, 4), dtype=t.float32, device=self.device)
        check(lib().rm_deinterleave(C.c_void_p(gathered.data_ptr()), C.c_void_p(frame.data_ptr()), W, H, tile_rows,
                                    num_shards, shard_stride_rows, self._stream()))
        return frame

    def tiles_to_rgba8(self, tiles, out=None):
        """rm_tiles_to_rgba8: a shard's packed float4 rows → RGBA8, same rows (clamp → ×255 → round, no flip)."""
        t = self.torch
        rows, W = tiles.shape[0], tiles.shape[1]
        if out is None:
            out = t.empty((rows, W, 4), dtype=t.uint8, device=self.device)
        check(lib().rm_tiles_to_rgba8(C.c_void_p(tiles.data_ptr()), C.c_void_p(out.data_ptr()), W, rows, self._stream()))
        return out

    def deinterleave_rgba8(self, gathered8, W, H, tile_rows, num_shards, shard_stride_rows=0, flip=True):
        """rm_deinterleave_rgba8: gathered RGBA8 slots → the frame's image (flip: row 0 = top, as to_rgba8 writes it)."""
        t = self.torch
        img = t.empty((H, W, 4), dtype=t.uint8, device=self.device)
        check(lib().rm_deinterleave_rgba8(C.c_void_p(gathered8.data_ptr()), C.c_void_p(img.data_ptr()), W, H, tile_rows, num_shards,
                                          shard_stride_rows, 1 if flip else 0, self._stream()))
        return img

    def post_process(self, frame, bright, post):
        """rm_post_process: bloom / HDR / gamma / FXAA (applyLightEffects + applyFXAA, realtimerender.cpp:78-165)."""
        t = self.torch
        H, W = frame.shape[0], frame.shape[1]
        out = t.empty((H, W, 4), dtype=t.float32, device=self.device)
        check(lib().rm_post_process(C.c_void_p(frame.data_ptr()), C.c_void_p(bright.data_ptr()) if bright is not None else None,
                                    C.c_void_p(out.data_ptr()), W, H, C.byref(post), self._stream()))
        return out

    def to_rgba8(self, frame):
        """Clamp/quantise + vertical flip (saveViewportImage, realtime.cpp:284-350) → uint8 (H, W, 4)."""
        t = self.torch
        H, W = frame.shape[0], frame.shape[1]
        out = t.empty((H, W, 4), dtype=t.uint8, device=self.device)
        check(lib().rm_frame_to_rgba8(C.c_void_p(frame.data_ptr()), C.c_void_p(out.data_ptr()), W, H, self._stream()))
        return out

    def save_png(self, frame, path):
        img = self.to_rgba8(frame).cpu().contiguous()
        check(lib().rm_write_png(str(path).encode(), C.c_void_p(img.data_ptr()), img.shape[1], img.shape[0]))

    def probe_math(self, fn, x, y=None, z=None):
        t = self.torch
        out = t.empty_like(x)
        check(lib().rm_probe_math(fn, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()) if y is not None else None,
                                  C.c_void_p(z.data_ptr()) if z is not None else None, C.c_void_p(out.data_ptr()),
                                  x.numel(), self._stream()))
        return out

    def probe_sdscene(self, tables, settings, pts):
        t = self.torch
        n = pts.shape[0]
        out = t.empty((n, 4), dtype=t.float32, device=self.device)
        check(lib().rm_probe_sdscene(tables.objects, tables.num_objects, C.byref(tables.globals_), C.byref(settings),
                                     C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()), n, self._stream()))
        return out
