"""Synthetic scene tables (no file IO) for the benchmark configurations of BASELINE.json.

Each builder restates a reference scenefile as constants and builds the camera through the
library's rm_camera_build, so the bench and the GPU tests feed the kernel exactly what the loader would.
"""
import json


def _scenefile(camera_pos, lights, primitive):
    """A one-primitive scenefile in the reference's schema (src/utils/scenefilereader.cpp), as text for rm_scene_load_string."""
    groups = [{"lights": [dict(type="directional", color=list(c), direction=list(d))]} for c, d in lights]
    groups.append({"groups": [{"primitives": [primitive]}]})
    return json.dumps({"name": "root",
                       "globalData": {"ambientCoeff": 0.5, "diffuseCoeff": 0.5, "specularCoeff": 0.5, "transparentCoeff": 0.5},
                       "cameraData": {"position": list(camera_pos), "up": [0.0, 1.0, 0.0], "heightAngle": 30.0, "focus": [0.0, 0.0, 0.0]},
                       "groups": groups})


def _tables(text, W, H):
    from .render import Scene
    sc = Scene(text=text)
    try:
        return sc.tables(W, H)
    finally:
        sc.close()


def mandelbulb(W, H):
    """scenefiles/simple/unit_mandelbulb.json — the north-star workload (BASELINE.json configs[2]) — restated as constants
    and passed through the library's own loader and camera (rm_scene_load_string, rm_camera_build), so the kernel is fed
    exactly the tables the scenefile gives (tests/test_host_loader.py compares them byte for byte)."""
    return _tables(_scenefile((0, 0, 4.5),
                              [((1, 1, 1), (0.0, 0.0, 1.0)), ((1.5, 1.1, 0.7), (0, -1, 0)), ((1, 1, 1), (0.0, 0.0, -1.0))],
                              dict(type="mandelbulb", ambient=[0.3, 0.3, 0.3], diffuse=[1, 1, 1], specular=[1, 1, 1],
                                   shininess=100.0, ior=1.5)), W, H)


def mengersponge(W, H):
    """scenefiles/simple/unit_mengersponge.json (BASELINE.json configs[4]) restated the same way: camera (0,0,4) looking at
    the origin, the file's three directional lights, one Menger sponge with the file's material."""
    return _tables(_scenefile((0, 0, 4),
                              [((0.25, 0.2, 0.15), (-0.707, 0.0, 0.707)), ((1.5, 1.1, 0.7), (0, -1, 0)), ((2, 2, 2), (0.0, 0.0, -1.0))],
                              dict(type="mengersponge", ambient=[0.3, 0.3, 0.3], diffuse=[0.3, 0.3, 0.3], specular=[1, 1, 1],
                                   shininess=20.0, reflective=[0.8, 0.8, 0.8], ior=1.5)), W, H)
