"""Synthetic scene tables (no file IO) for the benchmark configurations of BASELINE.json.

Each builder restates a reference scenefile as constants and builds the camera through the
library's rm_camera_build, so the bench and the GPU tests feed the kernel exactly what the loader would.
"""
import math

import numpy as np

from . import abi
from .render import SceneTables, build_camera


def _obj(type_, model=None, scale_factor=1.0, ambient=(0, 0, 0), diffuse=(1, 1, 1), specular=(0, 0, 0),
         shininess=0.0, reflective=(0, 0, 0), transparent=(0, 0, 0), ior=0.0):
    o = abi.RmObject()
    o.type = type_
    M = np.eye(4) if model is None else np.asarray(model, dtype=np.float64)
    inv = np.linalg.inv(M).astype(np.float32).T.reshape(-1)
    for i in range(16):
        o.invModel[i] = float(inv[i])
    o.scaleFactor, o.shininess, o.ior = scale_factor, shininess, ior
    for i in range(3):
        o.cAmbient[i], o.cDiffuse[i], o.cSpecular[i] = ambient[i], diffuse[i], specular[i]
        o.cReflective[i], o.cTransparent[i] = reflective[i], transparent[i]
    o.texLoc, o.lightIdx = -1, -1
    return o


def _dir_light(color, direction):
    li = abi.RmLight()
    li.type = abi.RM_LIGHT_DIRECTIONAL
    for i in range(3):
        li.color[i], li.dir[i] = color[i], direction[i]
    li.func[0] = 1.0
    return li


def _globals(ka=0.5, kd=0.5, ks=0.5, kt=0.5, power=8.0):
    g = abi.RmGlobals(ka, kd, ks, kt, power)
    return g


def mandelbulb(W, H):
    """scenefiles/simple/unit_mandelbulb.json: the north-star workload (BASELINE.json configs[2])."""
    cam, _, _ = build_camera((0, 0, 4.5), (0, 0, -4.5), (0, 1, 0), math.radians(30.0), W, H)
    objs = (abi.RmObject * 1)(_obj(abi.RM_MANDELBULB, ambient=(.3, .3, .3), specular=(1, 1, 1), shininess=100.0, ior=1.5))
    lights = (abi.RmLight * 3)(_dir_light((1, 1, 1), (0, 0, 1)), _dir_light((1.5, 1.1, 0.7), (0, -1, 0)),
                               _dir_light((1, 1, 1), (0, 0, -1)))
    return SceneTables(cam, objs, 1, lights, 3, _globals())


def mengersponge(W, H):
    """scenefiles/simple/unit_mengersponge.json camera/lights with one reflective Menger sponge."""
    cam, _, _ = build_camera((3, 3, 3), (-3, -3, -3), (0, 1, 0), math.radians(30.0), W, H)
    objs = (abi.RmObject * 1)(_obj(abi.RM_MENGERSPONGE, ambient=(.3, .3, .3), specular=(1, 1, 1), shininess=25.0,
                                   reflective=(.3, .3, .3)))
    lights = (abi.RmLight * 3)(_dir_light((1, 1, 1), (-1, -1, -1)), _dir_light((.6, .6, .6), (1, -1, 0)),
                               _dir_light((.4, .4, .4), (0, -1, 1)))
    return SceneTables(cam, objs, 1, lights, 3, _globals())
