"""Loader of the C-ABI shared library (raymarcher_amd/lib/libraymarcher_amd.so).

The product path has no CPU fallback: if the library is missing this raises, it never routes
anywhere else.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C raymarcher_amd/csrc`.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libraymarcher_amd.so")
_LIB = None

_P = C.POINTER
_SCENE_ARGS = [_P(abi.RmCamera), _P(abi.RmObject), C.c_int, _P(abi.RmLight), C.c_int, _P(abi.RmGlobals),
               _P(abi.RmSettings)]

# name -> (restype, argtypes); every symbol include/raymarcher_amd.h declares.
SIGNATURES = {
    "rm_settings_default": (None, [_P(abi.RmSettings)]),
    "rm_abi_version": (C.c_int, []),
    "rm_status_string": (C.c_char_p, [C.c_int]),
    "rm_last_error": (C.c_char_p, []),
    "rm_device_count": (C.c_int, []),
    "rm_set_device": (C.c_int, [C.c_int]),
    "rm_render": (C.c_int, _SCENE_ARGS + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rm_render_ex": (C.c_int, _SCENE_ARGS + [_P(abi.RmTexture), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "rm_render_res": (C.c_int, _SCENE_ARGS + [_P(abi.RmResources), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    "rm_render_tiles_res": (C.c_int, _SCENE_ARGS + [_P(abi.RmResources), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                    C.c_void_p, C.c_void_p]),
    "rm_skybox_face_path": (C.c_char_p, [C.c_int, C.c_int]),
    "rm_ltc_quantise": (None, [C.c_void_p, C.c_void_p, C.c_int]),
    "rm_image_load": (C.c_int, [C.c_char_p, C.c_int, _P(C.c_void_p), _P(C.c_int), _P(C.c_int)]),
    "rm_image_free": (None, [C.c_void_p]),
    "rm_render_tiles": (C.c_int, _SCENE_ARGS + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p]),
    "rm_shard_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "rm_shard_row_to_frame": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rm_deinterleave": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rm_gather_create": (C.c_int, [_P(C.c_int), C.c_int, _P(C.c_void_p)]),
    "rm_gather_create_ex": (C.c_int, [_P(C.c_int), C.c_int, C.c_uint, _P(C.c_void_p)]),
    "rm_gather_destroy": (None, [C.c_void_p]),
    "rm_tiles_to_rgba8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rm_gather_tiles_rgba8": (C.c_int, [C.c_void_p, _P(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _P(C.c_void_p)]),
    "rm_deinterleave_rgba8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rm_gather_slot_rows": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "rm_gather_tiles": (C.c_int, [C.c_void_p, _P(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _P(C.c_void_p)]),
    "rm_render_counted": (C.c_int, _SCENE_ARGS + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                  _P(abi.RmCounters)]),
    "rm_render_counted_ex": (C.c_int, _SCENE_ARGS + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                     _P(abi.RmCounters)]),
    "rm_render_counted_res": (C.c_int, _SCENE_ARGS + [_P(abi.RmResources), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                      C.c_int, _P(abi.RmCounters)]),
    "rm_render_clocked": (C.c_int, _SCENE_ARGS + [C.c_int, C.c_int, C.c_void_p, _P(C.c_double), C.c_void_p]),
    "rm_set_timing": (C.c_int, [C.c_int]),
    "rm_get_timing": (C.c_int, [_P(C.c_double), _P(C.c_int)]),
    "rm_get_stage_timing": (C.c_int, [_P(C.c_double), _P(C.c_double), _P(C.c_int)]),
    "rm_set_kernel_path": (C.c_int, [C.c_int]),
    "rm_debug_last_path": (C.c_int, []),
    "rm_debug_last_split": (C.c_int, []),
    "rm_debug_set_light_split": (C.c_int, [C.c_int]),
    "rm_debug_set_tile_shape": (C.c_int, [C.c_int]),
    "rm_set_root_relief": (C.c_int, [C.c_int]),
    "rm_get_root_relief": (C.c_int, []),
    "rm_set_workspace_limit": (C.c_int, [C.c_ulonglong]),
    "rm_release_workspaces": (C.c_int, [C.POINTER(C.c_ulonglong)]),
    "rm_set_tile_order": (C.c_int, [C.c_int]),
    "rm_debug_set_tile_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rm_debug_ray_planes": (C.c_int, [_P(abi.RmCamera), _P(C.c_float)]),
    "rm_debug_cull_bounds": (C.c_int, [_P(abi.RmObject), C.c_int, _P(abi.RmGlobals), _P(C.c_float)]),
    "rm_debug_check_math": (C.c_int, [_P(C.c_ulonglong)]),
    "rm_frame_to_rgba8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rm_post_process": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _P(abi.RmPostSettings), C.c_void_p]),
    "rm_probe_math": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "rm_probe_sdscene": (C.c_int, [_P(abi.RmObject), C.c_int, _P(abi.RmGlobals), _P(abi.RmSettings), C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_void_p]),
    "rm_host_settings_default": (None, [_P(abi.RmHostSettings)]),
    "rm_camera_build": (C.c_int, [_P(abi.RmCameraData), C.c_int, C.c_int, C.c_float, C.c_float, _P(C.c_float),
                                  _P(C.c_float), _P(abi.RmCamera)]),
    "rm_scene_load": (C.c_int, [C.c_char_p, _P(C.c_void_p)]),
    "rm_scene_load_string": (C.c_int, [C.c_char_p, _P(C.c_void_p)]),
    "rm_scene_free": (None, [C.c_void_p]),
    "rm_scene_num_objects": (C.c_int, [C.c_void_p]),
    "rm_scene_num_lights": (C.c_int, [C.c_void_p]),
    "rm_scene_objects": (_P(abi.RmObject), [C.c_void_p]),
    "rm_scene_lights": (_P(abi.RmLight), [C.c_void_p]),
    "rm_scene_globals": (C.c_int, [C.c_void_p, _P(abi.RmHostSettings), _P(abi.RmGlobals)]),
    "rm_scene_camera_data": (C.c_int, [C.c_void_p, _P(abi.RmCameraData)]),
    "rm_scene_object_texture": (C.c_char_p, [C.c_void_p, C.c_int]),
    "rm_write_png": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
    "rm_abi_sizeof": (C.c_int, [C.c_int]),
}


class RaymarcherError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"{text} (status {status})")
        self.status = status


def _share_torchs_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 (soname libamdhip64.so.7, loaded by path).  If this
    library were loaded first it would bring in the system copy, a later `import torch` would add its own, and the process
    would hold two HIP runtimes — the second one finds no device.  So the bundled copies are loaded (globally) before
    this library whenever torch is installed, whatever the import order; no torch in the environment: nothing to do."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return  # e.g. a CPU-only torch: fall back to the system runtime


def lib():
    """The loaded library with argtypes set; raises if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built and there is no fallback path. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'`.")
        _share_torchs_hip_runtime()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError = header/library drift, fail loudly
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def check(status):
    if status != abi.RM_OK:
        L = lib()
        raise RaymarcherError(status, f"{L.rm_status_string(status).decode()}: {L.rm_last_error().decode()}")
