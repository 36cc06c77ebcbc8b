"""Minimal ctypes EGL + OpenGL ES 3.0 harness on the SwiftShader software rasteriser that ships inside the
`kaleido` wheel of this container.  TEST INFRASTRUCTURE, container-only (never used on the GPU box, never
by the product)."""
import ctypes as C

import numpy as np

SS = "/usr/local/lib/python3.10/dist-packages/kaleido/executable/bin/swiftshader/"

GL_FRAGMENT_SHADER, GL_VERTEX_SHADER = 0x8B30, 0x8B31
GL_COMPILE_STATUS, GL_LINK_STATUS, GL_INFO_LOG_LENGTH = 0x8B81, 0x8B82, 0x8B84
GL_ARRAY_BUFFER, GL_STATIC_DRAW, GL_FLOAT, GL_TRIANGLES = 0x8892, 0x88E4, 0x1406, 0x0004
GL_TEXTURE_2D, GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 = 0x0DE1, 0x8D40, 0x8CE0
GL_RGBA32UI, GL_RGBA_INTEGER, GL_UNSIGNED_INT = 0x8D70, 0x8D99, 0x1405
GL_TEXTURE_MIN_FILTER, GL_TEXTURE_MAG_FILTER, GL_NEAREST = 0x2801, 0x2800, 0x2600
GL_FRAMEBUFFER_COMPLETE = 0x8CD5


class Context:
    def __init__(self):
        self.gl = C.CDLL(SS + "libGLESv2.so", mode=C.RTLD_GLOBAL)
        self.egl = C.CDLL(SS + "libEGL.so", mode=C.RTLD_GLOBAL)
        egl = self.egl
        egl.eglGetDisplay.restype = C.c_void_p
        egl.eglGetDisplay.argtypes = [C.c_void_p]
        dpy = egl.eglGetDisplay(None)
        maj, mi = C.c_int(), C.c_int()
        egl.eglInitialize.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        assert egl.eglInitialize(dpy, C.byref(maj), C.byref(mi))
        egl.eglBindAPI(0x30A0)
        attrs = (C.c_int * 5)(0x3033, 0x0001, 0x3040, 0x0040, 0x3038)
        cfg, n = C.c_void_p(), C.c_int()
        egl.eglChooseConfig.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        assert egl.eglChooseConfig(dpy, attrs, C.byref(cfg), 1, C.byref(n)) and n.value == 1
        egl.eglCreatePbufferSurface.restype = C.c_void_p
        egl.eglCreatePbufferSurface.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        surf = egl.eglCreatePbufferSurface(dpy, cfg, (C.c_int * 5)(0x3057, 16, 0x3056, 16, 0x3038))
        egl.eglCreateContext.restype = C.c_void_p
        egl.eglCreateContext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        ctx = egl.eglCreateContext(dpy, cfg, None, (C.c_int * 3)(0x3098, 3, 0x3038))
        egl.eglMakeCurrent.argtypes = [C.c_void_p] * 4
        assert egl.eglMakeCurrent(dpy, surf, surf, ctx)
        self.gl.glGetUniformLocation.argtypes = [C.c_uint, C.c_char_p]
        self.gl.glUniform1f.argtypes = [C.c_int, C.c_float]
        self.gl.glUniform2f.argtypes = [C.c_int, C.c_float, C.c_float]
        self.gl.glUniform3f.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
        self.gl.glUniform4f.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]

    def compile(self, kind, src):
        gl = self.gl
        sh = gl.glCreateShader(kind)
        b = src.encode()
        p = C.c_char_p(b)
        gl.glShaderSource(sh, 1, C.byref(p), None)
        gl.glCompileShader(sh)
        ok = C.c_int()
        gl.glGetShaderiv(sh, GL_COMPILE_STATUS, C.byref(ok))
        ln = C.c_int()
        gl.glGetShaderiv(sh, GL_INFO_LOG_LENGTH, C.byref(ln))
        log = C.create_string_buffer(max(ln.value, 1))
        gl.glGetShaderInfoLog(sh, ln.value, None, log)
        return sh, bool(ok.value), log.value.decode(errors="replace")

    def link(self, vs, fs):
        gl = self.gl
        prog = gl.glCreateProgram()
        gl.glAttachShader(prog, vs)
        gl.glAttachShader(prog, fs)
        gl.glLinkProgram(prog)
        ok = C.c_int()
        gl.glGetProgramiv(prog, GL_LINK_STATUS, C.byref(ok))
        ln = C.c_int()
        gl.glGetProgramiv(prog, GL_INFO_LOG_LENGTH, C.byref(ln))
        log = C.create_string_buffer(max(ln.value, 1))
        gl.glGetProgramInfoLog(prog, ln.value, None, log)
        return prog, bool(ok.value), log.value.decode(errors="replace")

    def target(self, W, H, attachments=1):
        """FBO with `attachments` RGBA32UI colour textures (exact fp32 read-back via floatBitsToUint)."""
        gl = self.gl
        fbo = C.c_uint()
        gl.glGenFramebuffers(1, C.byref(fbo))
        gl.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
        bufs = (C.c_uint * attachments)()
        for i in range(attachments):
            tex = C.c_uint()
            gl.glGenTextures(1, C.byref(tex))
            gl.glBindTexture(GL_TEXTURE_2D, tex)
            gl.glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32UI, W, H)
            gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST)
            gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST)
            gl.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + i, GL_TEXTURE_2D, tex, 0)
            bufs[i] = GL_COLOR_ATTACHMENT0 + i
        gl.glDrawBuffers(attachments, bufs)
        assert gl.glCheckFramebufferStatus(GL_FRAMEBUFFER) == GL_FRAMEBUFFER_COMPLETE
        gl.glViewport(0, 0, W, H)
        return fbo

    def draw_fullscreen(self, prog):
        """The reference's full-screen quad: two triangles over NDC [-1,1]² (realtimerender.cpp:192-261)."""
        gl = self.gl
        quad = np.array([-1, 1, -1, -1, 1, -1, 1, 1, -1, 1, 1, -1], dtype=np.float32)
        vao, vbo = C.c_uint(), C.c_uint()
        gl.glGenVertexArrays(1, C.byref(vao))
        gl.glBindVertexArray(vao)
        gl.glGenBuffers(1, C.byref(vbo))
        gl.glBindBuffer(GL_ARRAY_BUFFER, vbo)
        gl.glBufferData(GL_ARRAY_BUFFER, C.c_long(quad.nbytes), quad.ctypes.data_as(C.c_void_p), GL_STATIC_DRAW)
        gl.glEnableVertexAttribArray(0)
        gl.glVertexAttribPointer(0, 2, GL_FLOAT, 0, 8, None)
        gl.glUseProgram(prog)
        gl.glDrawArrays(GL_TRIANGLES, 0, 6)
        gl.glFinish()

    def read(self, W, H, attachment=0):
        gl = self.gl
        gl.glReadBuffer(GL_COLOR_ATTACHMENT0 + attachment)
        out = np.zeros((H, W, 4), dtype=np.uint32)
        gl.glReadPixels(0, 0, W, H, GL_RGBA_INTEGER, GL_UNSIGNED_INT, out.ctypes.data_as(C.c_void_p))
        return out.view(np.float32)

    def error(self):
        return self.gl.glGetError()
