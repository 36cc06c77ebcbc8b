"""ctypes mirror of include/raymarcher_amd.h (the C-ABI PODs and enums).

Field order and types must match the header byte for byte; tests/test_abi.py checks sizeof() of every
struct against the library's own `rm_abi_sizeof`.
"""
import ctypes as C

RM_ABI_VERSION = 4
RM_COUNT_REFERENCE, RM_COUNT_EXECUTED = 1, 2
RM_MAX_LIGHTS = 10
RM_MAX_OBJECTS = 30

(RM_CUBE, RM_CONE, RM_CYLINDER, RM_SPHERE, RM_OCTAHEDRON, RM_TORUS, RM_CAPSULE, RM_DEATHSTAR, RM_RECTANGLE,
 RM_MANDELBROT, RM_MANDELBULB, RM_MENGERSPONGE, RM_SIERPINSKI, RM_CUSTOM) = range(14)
PRIMITIVE_NAMES = {
    "cube": RM_CUBE, "cone": RM_CONE, "cylinder": RM_CYLINDER, "sphere": RM_SPHERE, "octahedron": RM_OCTAHEDRON,
    "torus": RM_TORUS, "capsule": RM_CAPSULE, "deathstar": RM_DEATHSTAR, "rectangle": RM_RECTANGLE,
    "mandelbrot": RM_MANDELBROT, "mandelbulb": RM_MANDELBULB, "mengersponge": RM_MENGERSPONGE,
    "sierpinski": RM_SIERPINSKI, "custom": RM_CUSTOM,
}
RM_LIGHT_POINT, RM_LIGHT_DIRECTIONAL, RM_LIGHT_SPOT, RM_LIGHT_AREA = range(4)

RM_FEAT_SKY_BACKGROUND = 1 << 0
RM_FEAT_NIGHTSKY_BACKGROUND = 1 << 1
RM_FEAT_DARK_BACKGROUND = 1 << 2
RM_FEAT_WHITE_BACKGROUND = 1 << 3
RM_FEAT_CLOUD = 1 << 4
RM_FEAT_TERRAIN = 1 << 5
RM_FEAT_SEA = 1 << 6
RM_FEAT_PERLIN_BUMP = 1 << 7
RM_FEAT_BULB_POWER8_ALGEBRAIC = 1 << 8  # opt-in evaluation scheme, see include/raymarcher_amd.h
RM_FEAT_REFERENCE_DEFAULT = RM_FEAT_WHITE_BACKGROUND | RM_FEAT_PERLIN_BUMP

RM_OK, RM_ERR_INVALID_ARGUMENT, RM_ERR_CAPACITY, RM_ERR_UNSUPPORTED, RM_ERR_DEVICE, RM_ERR_IO, RM_ERR_PARSE = range(7)

(RM_FN_SIN, RM_FN_COS, RM_FN_ACOS, RM_FN_ATAN2, RM_FN_LOG2, RM_FN_EXP2, RM_FN_POW, RM_FN_SQRT, RM_FN_DIV,
 RM_FN_PNOISE3, RM_FN_ASIN, RM_FN_Q16, RM_FN_SQRT_FAST, RM_FN_DIVR, RM_FN_RCP, RM_FN_SMOOTHSTEP, RM_FN_MIN, RM_FN_MAX, RM_FN_FRACT,
 RM_FN_MEDIAN_ABS, RM_FN_COUNT) = range(21)

f32 = C.c_float
i32 = C.c_int32


class RmObject(C.Structure):
    _fields_ = [
        ("type", i32), ("invModel", f32 * 16), ("scaleFactor", f32), ("shininess", f32), ("blend", f32),
        ("ior", f32), ("cAmbient", f32 * 3), ("cDiffuse", f32 * 3), ("cSpecular", f32 * 3),
        ("cReflective", f32 * 3), ("cTransparent", f32 * 3), ("texLoc", i32), ("repeatU", f32),
        ("repeatV", f32), ("isEmissive", i32), ("color", f32 * 3), ("lightIdx", i32),
    ]


class RmLight(C.Structure):
    _fields_ = [
        ("type", i32), ("color", f32 * 3), ("dir", f32 * 3), ("pos", f32 * 3), ("func", f32 * 3),
        ("angle", f32), ("penumbra", f32), ("points", (f32 * 3) * 4), ("intensity", f32), ("twoSided", i32),
    ]


RM_MAX_TEXTURES = 10


class RmTexture(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", i32), ("height", i32)]


RM_LTC_SIZE = 64


class RmResources(C.Structure):
    _fields_ = [("textures", C.POINTER(RmTexture)), ("numTextures", i32), ("noise", RmTexture), ("skybox", RmTexture * 6),
                ("ltc1", C.c_void_p), ("ltc2", C.c_void_p)]


class RmCamera(C.Structure):
    _fields_ = [("invProjView", f32 * 16), ("initialFar", f32), ("eyePosition", f32 * 4)]


class RmGlobals(C.Structure):
    _fields_ = [("ka", f32), ("kd", f32), ("ks", f32), ("kt", f32), ("power", f32), ("juliaSeed", f32 * 2),
                ("iTime", f32), ("isTwoD", i32)]


class RmSettings(C.Structure):
    _fields_ = [("enableSoftShadow", i32), ("enableReflection", i32), ("enableRefraction", i32),
                ("enableAmbientOcclusion", i32), ("enableSkyBox", i32), ("maxSteps", i32), ("fractalIters", i32),
                ("mengerLevels", i32), ("numReflection", i32), ("features", C.c_uint32)]


class RmCounters(C.Structure):
    _fields_ = [("sceneEvals", C.c_uint64), ("bulbIters", C.c_uint64), ("hitPixels", C.c_uint64),
                ("shadedPoints", C.c_uint64), ("terrainEvals", C.c_uint64), ("cloudEvals", C.c_uint64), ("shapeEvals", C.c_uint64)]


class RmPostSettings(C.Structure):
    _fields_ = [("enableFXAA", i32), ("enableGammaCorrection", i32), ("enableHDR", i32), ("enableBloom", i32),
                ("exposure", f32)]


class RmHostSettings(C.Structure):
    _fields_ = [("screenWidth", i32), ("screenHeight", i32), ("nearPlane", f32), ("farPlane", f32),
                ("twoDSpace", i32), ("enableSoftShadow", i32), ("enableReflection", i32),
                ("enableRefraction", i32), ("enableAmbientOcculusion", i32), ("power", f32),
                ("juliaSeed", f32 * 2)]


class RmCameraData(C.Structure):
    _fields_ = [("pos", f32 * 4), ("look", f32 * 4), ("up", f32 * 4), ("heightAngle", f32)]


def default_settings(**over):
    """RmSettings with the reference's constants (frag:28,29,45,1056; frag:9,15)."""
    s = RmSettings(0, 0, 0, 0, 0, 256, 20, 4, 1, RM_FEAT_REFERENCE_DEFAULT)
    for k, v in over.items():
        setattr(s, k, v)
    return s
