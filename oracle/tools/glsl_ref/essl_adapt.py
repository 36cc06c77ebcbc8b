"""Mechanical adaptation of the reference fragment shader to GLSL ES 3.00 (SwiftShader), done IN MEMORY
at golden-generation time.  TEST INFRASTRUCTURE, container-only.

The reference's hot path is `#version 330 core` GLSL that needs a desktop-GL context, which this
container cannot create; an OpenGL ES 3.0 software context (SwiftShader) does work.  ESSL 3.00 differs
from GLSL 3.30 in ways that do not change arithmetic: no implicit int→float conversion, no uniform
initialisers, constant-index sampler arrays, precision qualifiers.  This script rewrites exactly those
spots — nothing of the shader's text is stored in this repository — and the result is compiled and run to
produce float fixtures that cross-check the CPU oracle (tests/golden/glsl/).

Edits (all mechanical):
  E1 header: version / precision lines
  E2 on every line the ESSL compiler rejects for an int/float mismatch: integer literals → float literals
     (driven by the compiler's own error list, repeated until it compiles)
  E3 uniform initialiser removed; global initialised from a uniform → macro; bare `return;` in a
     non-void function → `return <zero>`
  E4 sampler arrays indexed with a dynamic index → index 0 (only untextured scenes are rendered)
  E5 struct array sizes reduced to fit SwiftShader's 261 fragment uniform vectors
  E6 outputs routed through floatBitsToUint into RGBA32UI attachments for exact read-back
  E7 `#define` feature lines switched on/off per test case; loop-bound constants set per test case
  E8 partial evaluation against the test case's uniform values, so that SwiftShader (which inlines every call
     into one function and needs tens of minutes for the full shader) only compiles code that can execute:
     option uniforms that are false in the case (`enableReflection`, `enableRefraction`,
     `enableAmbientOcculusion`, `enableSkyBox`, `isTwoD`) become the literal `false` in `if` conditions; the
     area-light test `li.type == AREA` becomes `false` when the scene has no area light; `sdMatch` branches
     of primitive types that do not occur in the scene are removed.  Executed arithmetic is unchanged.
"""
import re

HEADER = ("#version 300 es\nprecision highp float;\nprecision highp int;\nprecision highp sampler2D;\n"
          "precision highp samplerCube;\n")
HEADER_LINES = HEADER.count("\n") - 1  # lines added in front of the original line 2

_INT = re.compile(r"(?<![\w.\[])(\d+)(?![\w.\]])")


def floatify(line):
    """Integer literals → float literals, leaving array subscripts, identifiers and int() casts alone."""
    code, sep, comment = line.partition("//")
    # protect int(...)/ivec/uint contexts and `for (int ...` headers
    if re.search(r"\bfor\s*\(\s*int\b", code) or re.search(r"^\s*(const\s+)?int\b", code) or "layout" in code:
        return line
    out = _INT.sub(lambda m: m.group(1) + ".0", code)
    return out + sep + comment


def specialise(text, false_uniforms=(), present_types=None, has_area_light=False):
    """E8 — see module docstring."""
    for u in false_uniforms:
        text = re.sub(r"\bif\s*\(\s*" + u + r"\b", "if (false", text)
    if not has_area_light:
        text = re.sub(r"li\.type\s*==\s*AREA", "false", text)
    if present_types is not None:
        # sdMatch: `if (type == NAME) {` / `} else if (type == NAME) {` + one `return …;` line each (frag:1262-1293)
        lines = text.split("\n")
        out, i = [], 0
        names = ["CUBE", "CONE", "CYLINDER", "SPHERE", "OCTAHEDRON", "TORUS", "CAPSULE", "DEATHSTAR", "RECTANGLE",
                 "MANDELBROT", "MANDELBULB", "MENGERSPONGE", "SIERPINSKI", "CUSTOM"]
        while i < len(lines):
            m = re.match(r"^(\s*)(\}\s*else\s+)?if\s*\(\s*type\s*==\s*(\w+)\s*\)\s*\{\s*$", lines[i])
            if m and m.group(3) in names and names.index(m.group(3)) not in present_types \
                    and i + 1 < len(lines) and re.match(r"^\s*return\s+sd\w+\(", lines[i + 1]):
                out.append(f"{m.group(1)}{m.group(2) or ''}if (false) {{")
                out.append(f"{m.group(1)}    return 0.0;")
                i += 2
                continue
            out.append(lines[i])
            i += 1
        text = "\n".join(out)
    return text


PROBE_MAIN = {
    # harness-only entry points appended after the reference's functions: evaluate ONE reference function at
    # points fetched from a float texture, so function-level values can be cross-checked (the reference's
    # own main() only exposes final colours).
    "sdscene": "SceneMin m = sdScene(q.xyz); fragColor = vec4(m.minD, float(m.minObjIdx), m.trap.y, m.trap.z);",
    "pnoise": "fragColor = vec4(pnoise(q.xyz), 0.0, 0.0, 0.0);",
    "normal": "fragColor = vec4(getNormal(q.xyz), 0.0);",
    "cloudsfbm": "fragColor = cloudsFbm(q.xyz);",
    "cloudsmap": "float nn = 0.0; vec4 r = cloudsMap(q.xyz, nn); fragColor = vec4(r.x, r.z, nn, 0.0);",
    "terrain": "vec2 e = sdTerrain(q.xz); fragColor = vec4(e.x, e.y, 0.0, 0.0);",
    "sea": "fragColor = vec4(seaMap(q.xyz), seaMapD(q.xyz), noiseW(q.xz), sea_octave(q.xz, 1.0));",
    "sinhash": "fragColor = vec4(sin(q.x), cos(q.x), hash(q.xy), 0.0);",
    "moon": "fragColor = vec4(getMoonColor(normalize(q.xyz)), noiseV(q.xyz));",
}


def define_ub10(text):
    """E9 (only for the `*_ub10` fixtures): cloudsMap leaves its `out float nnd` unset on the early return although
    cloudMarch reads it (frag:1961-1974, 1994-1995).  This gives it the value of the oracle's UB10 decision — `nnd = -d`
    before the return, the order of the function's origin — so that the REST of the cloud path can be compared tightly."""
    new, n = re.subn(r"if\(\s*d\s*>\s*0\.0\s*\)\s*return\s+vec4\(-d,\s*0\.0,\s*0\.0,\s*0\.0\);", "nnd = -d; if( d>0.0 ) return vec4(-d,0.0,0.0,0.0);", text)
    assert n == 1, n
    return new


def define_ub1(text):
    """E10 (only for the `*_ub1` fixtures): softshadow leaves `r.d` unset on a miss although getPhong multiplies by it when
    soft shadows are on (frag:1720-1722, 1928).  This assigns the oracle's UB1 value, `r.d = res` (the penumbra factor)."""
    new, n = re.subn(r"(?<![\w.])r\.intersectObj\s*=\s*-1\s*;", "r.intersectObj = -1; r.d = res;", text)
    assert n == 1, n
    return new


def adapt(src, defines=None, consts=None, max_objects=6, max_lights=4, probe=None, n_textures=1):
    """Return ESSL 3.00 source.  `defines`: {name: bool} for the #define block (frag:4-15);
    `consts`: {MAX_STEPS: n, MAX_STEPS_FRACTALS: n, NUM_REFLECTION: n, MENGER_LEVELS: n}."""
    defines = defines or {}
    consts = consts or {}
    lines = src.split("\n")
    assert lines[0].startswith("#version 330")
    lines[0] = HEADER.rstrip("\n")
    text = "\n".join(lines)
    # E7 feature defines: normalise every `// #define X` / `#define X` line of the known set
    for name in ("SKY_BACKGROUND", "NIGHTSKY_BACKGROUND", "DARK_BACKGROUND", "WHITE_BACKGROUND", "CLOUD", "TERRAIN",
                 "SEA", "PERLIN_BUMP"):
        if name in defines:
            text = re.sub(r"^\s*(//\s*)?#define\s+" + name + r"\s*$", ("#define " if defines[name] else "// #define ") + name,
                          text, flags=re.M)
    for cname, val in consts.items():
        if cname == "MENGER_LEVELS":
            text = re.sub(r"for\s*\(\s*int\s+m\s*=\s*0\s*;\s*m\s*<\s*4\s*;", f"for(int m=0; m<{int(val)};", text)
        else:
            text = re.sub(r"(const\s+int\s+" + cname + r"\s*=\s*)\d+", r"\g<1>" + str(int(val)), text)
    # E3
    text = re.sub(r"(uniform\s+float\s+terrainHeight)\s*=\s*[^;]+;", r"\1;", text)
    text = re.sub(r"^\s*float\s+SEA_TIME\s*=\s*([^;]+);", r"#define SEA_TIME (\1)", text, flags=re.M)
    # E4
    text = re.sub(r"texture\(customTextures\[[^\]]+\]", "texture(customTextures[0]", text)
    if n_textures <= 1:
        text = re.sub(r"objTextures\[texLoc\]", "objTextures[0]", text)
    else:  # ESSL 3.00 indexes sampler arrays with constant expressions only: a selection chain over the bound units
        def chain(m):
            uv, expr = m.group(1), f"texture(objTextures[{n_textures - 1}], {m.group(1)})"
            for k in range(n_textures - 2, -1, -1):
                expr = f"((texLoc == {k}) ? texture(objTextures[{k}], {uv}) : {expr})"
            return expr
        text = re.sub(r"texture\(objTextures\[texLoc\],\s*([^)]+)\)", chain, text)
    # E5
    text = re.sub(r"uniform\s+RayMarchObject\s+objects\[\d+\]", f"uniform RayMarchObject objects[{max_objects}]", text)
    text = re.sub(r"uniform\s+LightSource\s+lights\[\d+\]", f"uniform LightSource lights[{max_lights}]", text)
    # E6
    text = re.sub(r"layout\s*\(location\s*=\s*0\)\s*out\s+vec4\s+fragColor;", "vec4 fragColor;\nlayout(location = 0) out uvec4 fragBits;", text)
    text = re.sub(r"layout\s*\(location\s*=\s*1\)\s*out\s+vec4\s+BrightColor;", "vec4 BrightColor;\nlayout(location = 1) out uvec4 brightBits;", text)
    text = re.sub(r"void\s+main\s*\(\s*\)", "void main_ref()", text)
    if probe is None:
        body = "  main_ref();\n"
    else:
        text += "\nuniform highp sampler2D probePts;\n"
        body = "  vec4 q = texelFetch(probePts, ivec2(gl_FragCoord.xy), 0);\n  " + PROBE_MAIN[probe] + "\n"
    text += ("\nvoid main() {\n  fragColor = vec4(0.0); BrightColor = vec4(0.0, 0.0, 0.0, 1.0);\n" + body +
             "  fragBits = floatBitsToUint(fragColor);\n  brightBits = floatBitsToUint(BrightColor);\n}\n")
    return text


def fix_errors(text, log):
    """E2/E3: apply literal float-ification to every line the compiler complains about."""
    lines = text.split("\n")
    touched = set()
    int_consts = re.findall(r"^\s*const\s+int\s+(\w+)\s*=", text, flags=re.M)
    for m in re.finditer(r"ERROR: 0:(\d+): (.*)", log):
        ln = int(m.group(1)) - 1
        msg = m.group(2)
        if ln in touched or ln >= len(lines):
            continue
        touched.add(ln)
        new = floatify(lines[ln])
        if "return" in msg and re.search(r"\breturn\s*;", lines[ln]):
            new = re.sub(r"\breturn\s*;", "return ri;", lines[ln])  # seaRender's early exit (frag:2290)
        if new == lines[ln]:
            # int-typed names used in float arithmetic: named int constants and int loop/id variables
            for name in int_consts:
                new = re.sub(r"(?<![\w(])" + name + r"\b(?!\s*(=[^=]|\[))", f"float({name})", new)
            new = re.sub(r"\brd \+ idx\b", "rd + float(idx)", new)
            new = re.sub(r",\s*id\)", ", float(id))", new)
        lines[ln] = new
    return "\n".join(lines), len(touched)
