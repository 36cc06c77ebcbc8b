"""Run the reference's post-pass shaders (fullscreen.vert + blur.frag / hdr.frag / fxaa.frag) on SwiftShader the way
Realtime::applyBloom / applyLightEffects / applyFXAA drive them (src/realtimerender.cpp:92-165), with the
reference's framebuffer formats (RGBA16F ping-pong / HDR / bright targets, RGBA8 FXAA source).
TEST INFRASTRUCTURE, container-only.  ESSL adaptation (in memory): version/precision header, uniform
initialisers → constants, `1.0 / textureSize()` and `float * int` made explicit, final-stage outputs routed through
floatBitsToUint for exact read-back."""
import ctypes as C
import os
import re

import numpy as np

import gles
import run_ref

REF = "/root/reference/resources"
HDR = "#version 300 es\nprecision highp float;\nprecision highp int;\nprecision highp sampler2D;\n"


def _adapt(name, bits_out):
    src = open(os.path.join(REF, name)).read()
    src = src.replace("#version 330 core", HDR.rstrip("\n"))
    src = re.sub(r"uniform\s+float\s+weight\[5\]\s*=\s*float\[\]\s*\(", "const float weight[5] = float[5](", src)
    src = re.sub(r"uniform\s+float\s+multiplier\s*=\s*1\.0\s*;", "const float multiplier = 1.0;", src)
    src = src.replace("1.0 / textureSize(image, 0)", "1.0 / vec2(textureSize(image, 0))")
    src = re.sub(r"tex_offset\.(x|y) \* i\b", r"tex_offset.\1 * float(i)", src)
    if bits_out:
        src = src.replace("out vec4 FragColor;", "vec4 FragColor;\nlayout(location = 0) out uvec4 FragBits;")
        src = re.sub(r"void\s+main\s*\(\s*\)", "void main_ref()", src)
        src += "\nvoid main() { FragColor = vec4(0.0); main_ref(); FragBits = floatBitsToUint(FragColor); }\n"
    return src


def _program(frag_name, bits_out):
    c = run_ref.ctx()
    vs, ok, log = c.compile(gles.GL_VERTEX_SHADER, open(os.path.join(REF, "fullscreen.vert")).read().replace(
        "#version 330 core", "#version 300 es\nprecision highp float;"))
    assert ok, log
    fs, ok, log = c.compile(gles.GL_FRAGMENT_SHADER, _adapt(frag_name, bits_out))
    assert ok, log
    prog, ok, log = c.link(vs, fs)
    assert ok, log
    return prog


def _texture(data, internal, fmt, typ, linear=True, clamp=False):
    gl = run_ref.ctx().gl
    t = C.c_uint()
    gl.glGenTextures(1, C.byref(t))
    gl.glBindTexture(gles.GL_TEXTURE_2D, t)
    H, W = data.shape[:2]
    ptr = data.ctypes.data_as(C.c_void_p) if data is not None else None
    gl.glTexImage2D(gles.GL_TEXTURE_2D, 0, internal, W, H, 0, fmt, typ, ptr)
    f = 0x2601 if linear else 0x2600
    gl.glTexParameteri(gles.GL_TEXTURE_2D, gles.GL_TEXTURE_MIN_FILTER, f)
    gl.glTexParameteri(gles.GL_TEXTURE_2D, gles.GL_TEXTURE_MAG_FILTER, f)
    if clamp:
        gl.glTexParameteri(gles.GL_TEXTURE_2D, 0x2802, 0x812F)
        gl.glTexParameteri(gles.GL_TEXTURE_2D, 0x2803, 0x812F)
    return t


def _fbo_for(tex):
    gl = run_ref.ctx().gl
    fbo = C.c_uint()
    gl.glGenFramebuffers(1, C.byref(fbo))
    gl.glBindFramebuffer(gles.GL_FRAMEBUFFER, fbo)
    gl.glFramebufferTexture2D(gles.GL_FRAMEBUFFER, gles.GL_COLOR_ATTACHMENT0, gles.GL_TEXTURE_2D, tex, 0)
    bufs = (C.c_uint * 1)(gles.GL_COLOR_ATTACHMENT0)
    gl.glDrawBuffers(1, bufs)
    assert gl.glCheckFramebufferStatus(gles.GL_FRAMEBUFFER) == gles.GL_FRAMEBUFFER_COMPLETE
    return fbo


def _draw_quad(prog):
    """initFullScreenQuad (realtimerender.cpp:220-261): position + uv, uv (0,0) at the bottom-left."""
    gl = run_ref.ctx().gl
    q = np.array([-1, 1, 0, 0, 1, -1, -1, 0, 0, 0, 1, -1, 0, 1, 0, 1, 1, 0, 1, 1, -1, 1, 0, 0, 1, 1, -1, 0, 1, 0], dtype=np.float32)
    vao, vbo = C.c_uint(), C.c_uint()
    gl.glGenVertexArrays(1, C.byref(vao))
    gl.glBindVertexArray(vao)
    gl.glGenBuffers(1, C.byref(vbo))
    gl.glBindBuffer(gles.GL_ARRAY_BUFFER, vbo)
    gl.glBufferData(gles.GL_ARRAY_BUFFER, C.c_long(q.nbytes), q.ctypes.data_as(C.c_void_p), gles.GL_STATIC_DRAW)
    gl.glEnableVertexAttribArray(0)
    gl.glVertexAttribPointer(0, 3, gles.GL_FLOAT, 0, 20, None)
    gl.glEnableVertexAttribArray(1)
    gl.glVertexAttribPointer(1, 2, gles.GL_FLOAT, 0, 20, C.c_void_p(12))
    gl.glUseProgram(prog)
    gl.glDrawArrays(gles.GL_TRIANGLES, 0, 6)
    gl.glFinish()


def q16(a):
    return a.astype(np.float16).astype(np.float32)


def post_process(frag, bright, enableFXAA=0, enableGamma=0, enableHDR=0, enableBloom=0, exposure=1.0, fxaa_source="f32"):
    """Returns the (H, W, 4) float32 colour the last enabled pass outputs (before the 8-bit framebuffer)."""
    c = run_ref.ctx()
    gl = c.gl
    H, W = frag.shape[:2]
    gl.glViewport(0, 0, W, H)
    GL_RGBA16F, GL_RGBA, GL_RGBA8, GL_UBYTE = 0x881A, 0x1908, 0x8058, 0x1401
    loc = lambda p, n: gl.glGetUniformLocation(p, n.encode())
    light = enableHDR or enableGamma or enableBloom
    stage = frag[..., :3].astype(np.float32)
    if light:
        hdr_tex = _texture(np.ascontiguousarray(q16(frag)), GL_RGBA16F, GL_RGBA, gles.GL_FLOAT)
        bloom_tex = None
        if enableBloom:
            bright_tex = _texture(np.ascontiguousarray(q16(bright)), GL_RGBA16F, GL_RGBA, gles.GL_FLOAT, clamp=True)
            pp = [_texture(np.zeros((H, W, 4), np.float32), GL_RGBA16F, GL_RGBA, gles.GL_FLOAT, clamp=True) for _ in range(2)]
            fb = [_fbo_for(t) for t in pp]
            prog = _program("blur.frag", bits_out=False)
            gl.glUseProgram(prog)
            gl.glUniform1i(loc(prog, "image"), 0)
            horizontal = True
            for i in range(10):  # applyBloom, realtimerender.cpp:92-108
                gl.glBindFramebuffer(gles.GL_FRAMEBUFFER, fb[int(horizontal)])
                gl.glUniform1i(loc(prog, "horizontal"), int(horizontal))
                gl.glActiveTexture(0x84C0)
                gl.glBindTexture(gles.GL_TEXTURE_2D, bright_tex if i == 0 else pp[int(not horizontal)])
                _draw_quad(prog)
                horizontal = not horizontal
            bloom_tex = pp[int(horizontal)]  # `side` of applyLightEffects
        # hdr.frag into a float-bits target (the reference's destination is 8-bit; quantised below when FXAA follows)
        out_tex = C.c_uint()
        gl.glGenTextures(1, C.byref(out_tex))
        gl.glBindTexture(gles.GL_TEXTURE_2D, out_tex)
        gl.glTexStorage2D(gles.GL_TEXTURE_2D, 1, gles.GL_RGBA32UI, W, H)
        gl.glTexParameteri(gles.GL_TEXTURE_2D, gles.GL_TEXTURE_MIN_FILTER, gles.GL_NEAREST)
        gl.glTexParameteri(gles.GL_TEXTURE_2D, gles.GL_TEXTURE_MAG_FILTER, gles.GL_NEAREST)
        _fbo_for(out_tex)
        prog = _program("hdr.frag", bits_out=True)
        gl.glUseProgram(prog)
        gl.glUniform1i(loc(prog, "hdrBuffer"), 0)
        gl.glUniform1i(loc(prog, "bloomBlur"), 1)
        gl.glUniform1i(loc(prog, "hdr"), int(enableHDR))
        gl.glUniform1i(loc(prog, "bloom"), int(enableBloom))
        gl.glUniform1f(loc(prog, "exposure"), float(exposure))
        gl.glActiveTexture(0x84C0)
        gl.glBindTexture(gles.GL_TEXTURE_2D, hdr_tex)
        gl.glActiveTexture(0x84C1)
        gl.glBindTexture(gles.GL_TEXTURE_2D, bloom_tex if bloom_tex is not None else hdr_tex)
        gl.glActiveTexture(0x84C0)
        _draw_quad(prog)
        stage = c.read(W, H, 0)[..., :3].copy()
    if not enableFXAA:
        out = np.concatenate([stage, np.ones((H, W, 1), np.float32) if light else frag[..., 3:4]], -1)
        return out.astype(np.float32)
    src8 = (np.clip(stage, 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)  # RGBA8 m_customFBOColorTexture
    src8 = np.ascontiguousarray(np.concatenate([src8, np.full((H, W, 1), 255, np.uint8)], -1))
    if fxaa_source == "u8":  # the reference's own format; SwiftShader filters RGBA8 with 8-bit fixed-point weights
        fx_src = _texture(src8, GL_RGBA8, GL_RGBA, GL_UBYTE)  # LINEAR, wrap left at the GL default (REPEAT)
    else:  # same 8-bit values, held as floats so the bilinear filter runs in binary32 (OES_texture_float_linear)
        srcf = np.ascontiguousarray(src8.astype(np.float32) / np.float32(255.0))
        fx_src = _texture(srcf, 0x8814, GL_RGBA, gles.GL_FLOAT)
    out_tex = C.c_uint()
    gl.glGenTextures(1, C.byref(out_tex))
    gl.glBindTexture(gles.GL_TEXTURE_2D, out_tex)
    gl.glTexStorage2D(gles.GL_TEXTURE_2D, 1, gles.GL_RGBA32UI, W, H)
    gl.glTexParameteri(gles.GL_TEXTURE_2D, gles.GL_TEXTURE_MIN_FILTER, gles.GL_NEAREST)
    gl.glTexParameteri(gles.GL_TEXTURE_2D, gles.GL_TEXTURE_MAG_FILTER, gles.GL_NEAREST)
    _fbo_for(out_tex)
    prog = _program("fxaa.frag", bits_out=True)
    gl.glUseProgram(prog)
    gl.glUniform1i(loc(prog, "screenTexture"), 0)
    gl.glUniform2f(loc(prog, "inverseScreenSize"), 1.0 / W, 1.0 / H)
    gl.glActiveTexture(0x84C0)
    gl.glBindTexture(gles.GL_TEXTURE_2D, fx_src)
    _draw_quad(prog)
    assert c.error() == 0
    return c.read(W, H, 0).copy()


def fxaa_tie_mask(stage_rgb, eps=1e-6):
    """(H, W) bool: pixels where one of fxaa.frag's comparisons is decided by less than `eps` when the shader is
    evaluated in float64 on the 8-bit-quantised source.  With 8-bit inputs exact ties are common (symmetric edges give
    distance1 == distance2, equal neighbour lumas give |gradient1| == |gradient2| …) and the side a tie falls on is
    decided by the last-bit rounding of the implementation's sqrt/dot, so those pixels carry no parity information.
    Follows the control flow of resources/fxaa.frag:23-170; only the margins are kept, not the colour."""
    H, W = stage_rgb.shape[:2]
    src = ((np.clip(stage_rgb[..., :3], 0, 1) * np.float32(255) + np.float32(0.5)).astype(np.uint8)).astype(np.float64) / 255.0
    wl = np.array([0.299, 0.587, 0.114])

    def tex(u, v):
        x, y = u * W - 0.5, v * H - 0.5
        x0, y0 = int(np.floor(x)), int(np.floor(y))
        fx, fy = x - x0, y - y0
        g = lambda xx, yy: src[yy % H, xx % W]
        return (g(x0, y0) * (1 - fx) + g(x0 + 1, y0) * fx) * (1 - fy) + (g(x0, y0 + 1) * (1 - fx) + g(x0 + 1, y0 + 1) * fx) * fy

    luma = lambda c: np.sqrt(c @ wl)
    quality = [1, 1, 1, 1, 1, 1.5, 2, 2, 2, 2, 4, 8]

    def margin(px, py):
        m = [1.0]
        cmp = lambda a, b: m.append(abs(a - b))
        u, v, iu, iv = (px + 0.5) / W, (py + 0.5) / H, 1.0 / W, 1.0 / H
        L = lambda dx, dy: luma(tex(u + dx * iu, v + dy * iv))
        c, d, up, l, r = L(0, 0), L(0, -1), L(0, 1), L(-1, 0), L(1, 0)
        mx = max(c, d, up, l, r)
        rg = mx - min(c, d, up, l, r)
        thr = max(0.0312, mx * 0.125)
        cmp(rg, thr)
        if rg < thr:
            return min(m)
        dl, ur, ul, dr = L(-1, -1), L(1, 1), L(-1, 1), L(1, -1)
        eh = abs(-2 * l + dl + ul) + abs(-2 * c + d + up) * 2 + abs(-2 * r + dr + ur)
        ev = abs(-2 * up + ur + ul) + abs(-2 * c + l + r) * 2 + abs(-2 * d + dl + dr)
        cmp(eh, ev)
        hor = eh >= ev
        l1, l2 = (d, up) if hor else (l, r)
        g1, g2 = l1 - c, l2 - c
        cmp(abs(g1), abs(g2))
        s1 = abs(g1) >= abs(g2)
        gs = 0.25 * max(abs(g1), abs(g2))
        st = iv if hor else iu
        if s1:
            st = -st
        la = 0.5 * ((l1 if s1 else l2) + c)
        cu = np.array([u, v])
        cu[1 if hor else 0] += st * 0.5
        off = np.array([iu, 0.0]) if hor else np.array([0.0, iv])
        u1, u2 = cu - off, cu + off
        e1, e2 = luma(tex(*u1)) - la, luma(tex(*u2)) - la
        cmp(abs(e1), gs)
        cmp(abs(e2), gs)
        r1, r2 = abs(e1) >= gs, abs(e2) >= gs
        if not r1:
            u1 = u1 - off
        if not r2:
            u2 = u2 + off
        if not (r1 and r2):
            for i in range(2, 12):
                if not r1:
                    e1 = luma(tex(*u1)) - la
                    cmp(abs(e1), gs)
                if not r2:
                    e2 = luma(tex(*u2)) - la
                    cmp(abs(e2), gs)
                r1, r2 = abs(e1) >= gs, abs(e2) >= gs
                if not r1:
                    u1 = u1 - off * quality[i]
                if not r2:
                    u2 = u2 + off * quality[i]
                if r1 and r2:
                    break
        k = 0 if hor else 1
        d1, d2 = (u, v)[k] - u1[k], u2[k] - (u, v)[k]
        m.append(abs(d1 - d2) * max(W, H))
        cmp(c, la)
        cmp(e1 if d1 < d2 else e2, 0.0)
        return min(m)

    out = np.zeros((H, W), bool)
    for y in range(H):
        for x in range(W):
            out[y, x] = margin(x, y) < eps
    return out
