// headless_gl.cpp -- definitions behind the shim headers of this directory: no-op OpenGL / GLEW / GLUT, the CUDA-GL
// interop calls of main.cpp mapped to a host "pixel buffer object", cudaDeviceReset -> ptmi355_adaptor_reset.
// Link it (with adaptor/cuda_raytrace_core.cpp and libptmi355.so) in place of -lglut -lGLEW -lGL -lcudart when building
// the reference's unchanged host sources without a display; INTEGRATION.md has the command line.
//
// Test hooks (environment): PT_SHIM_NO_PBO=1   cudaGLMapBufferObject hands out NULL (a run without a display buffer)
//                           PT_SHIM_PBO_DUMP=<file>   the mapped buffer's bytes are written there at cudaDeviceReset
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "GL/glut.h"
#include "cuda_gl_interop.h"

extern "C" void ptmi355_adaptor_reset(void);

namespace {
GLuint g_next_id = 1, g_bound_unpack = 0;
std::map<GLuint, std::vector<unsigned char> > g_buffers;       // buffer objects that were given storage
GLuint g_mapped = 0;
void (*g_display)(void) = 0;
}

extern "C" {

// ---- GLEW / GLUT ---------------------------------------------------------------------------------------------
GLenum glewInit(void) { return GLEW_OK; }
void glutInit(int *, char **) {}
void glutInitDisplayMode(unsigned int) {}
void glutInitWindowSize(int, int) {}
int glutCreateWindow(const char *) { return 1; }
void glutDisplayFunc(void (*func)(void)) { g_display = func; }
void glutKeyboardFunc(void (*)(unsigned char, int, int)) {}
void glutMainLoop(void) { for (;;) { if (g_display) g_display(); else exit(0); } }
void glutSetWindowTitle(const char *) {}
void glutPostRedisplay(void) {}
void glutSwapBuffers(void) {}

// ---- OpenGL --------------------------------------------------------------------------------------------------
void glGenBuffers(GLsizei n, GLuint *buffers) { for (GLsizei i = 0; i < n; ++i) buffers[i] = g_next_id++; }
void glBindBuffer(GLenum target, GLuint buffer) { if (target == GL_PIXEL_UNPACK_BUFFER) g_bound_unpack = buffer; }
void glBufferData(GLenum target, GLsizeiptr size, const void *data, GLenum) {
    if (target != GL_PIXEL_UNPACK_BUFFER || !g_bound_unpack || size <= 0) return;   // only the PBO needs real storage
    std::vector<unsigned char> &b = g_buffers[g_bound_unpack];
    b.assign((size_t)size, 0);
    if (data) memcpy(b.data(), data, (size_t)size);
}
void glDeleteBuffers(GLsizei n, const GLuint *buffers) { for (GLsizei i = 0; i < n; ++i) g_buffers.erase(buffers[i]); }
void glGenTextures(GLsizei n, GLuint *textures) { for (GLsizei i = 0; i < n; ++i) textures[i] = g_next_id++; }
void glBindTexture(GLenum, GLuint) {}
void glTexParameteri(GLenum, GLenum, GLint) {}
void glTexImage2D(GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *) {}
void glTexSubImage2D(GLenum, GLint, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, const void *) {}
void glDeleteTextures(GLsizei, const GLuint *) {}
void glClear(GLbitfield) {}
void glDrawElements(GLenum, GLsizei, GLenum, const void *) {}
void glVertexAttribPointer(GLuint, GLint, GLenum, GLboolean, GLsizei, const void *) {}
void glEnableVertexAttribArray(GLuint) {}
void glUseProgram(GLuint) {}
void glActiveTexture(GLenum) {}
GLint glGetUniformLocation(GLuint, const GLchar *) { return -1; }
void glUniform1i(GLint, GLint) {}
GLuint glCreateShader(GLenum) { return g_next_id++; }
void glShaderSource(GLuint, GLsizei, const GLchar **, const GLint *) {}
void glCompileShader(GLuint) {}
void glGetShaderiv(GLuint, GLenum pname, GLint *params) { if (params) *params = (pname == GL_COMPILE_STATUS) ? GL_TRUE : 0; }
void glGetShaderInfoLog(GLuint, GLsizei, GLsizei *length, GLchar *infoLog) { if (length) *length = 0; if (infoLog) *infoLog = 0; }
GLuint glCreateProgram(void) { return g_next_id++; }
void glBindAttribLocation(GLuint, GLuint, const GLchar *) {}
void glAttachShader(GLuint, GLuint) {}
void glLinkProgram(GLuint) {}
void glGetProgramiv(GLuint, GLenum pname, GLint *params) { if (params) *params = (pname == GL_LINK_STATUS) ? GL_TRUE : 0; }
void glGetProgramInfoLog(GLuint, GLsizei, GLsizei *length, GLchar *infoLog) { if (length) *length = 0; if (infoLog) *infoLog = 0; }

// ---- CUDA runtime / GL interop ---------------------------------------------------------------------------------
cudaError_t cudaGLSetGLDevice(int) { return cudaSuccess; }
cudaError_t cudaGLRegisterBufferObject(unsigned int) { return cudaSuccess; }
cudaError_t cudaGLUnregisterBufferObject(unsigned int) { return cudaSuccess; }
cudaError_t cudaGLMapBufferObject(void **devPtr, unsigned int buffer) {
    const char *no = getenv("PT_SHIM_NO_PBO");
    std::map<GLuint, std::vector<unsigned char> >::iterator it = g_buffers.find(buffer);
    *devPtr = (no && atoi(no)) || it == g_buffers.end() ? 0 : (void *)it->second.data();
    g_mapped = buffer;
    return cudaSuccess;
}
cudaError_t cudaGLUnmapBufferObject(unsigned int) { return cudaSuccess; }
cudaError_t cudaThreadSynchronize(void) { return cudaSuccess; }
cudaError_t cudaGetLastError(void) { return cudaSuccess; }
const char *cudaGetErrorString(cudaError_t) { return "no error"; }
cudaError_t cudaDeviceReset(void) {
    if (const char *path = getenv("PT_SHIM_PBO_DUMP")) {
        std::map<GLuint, std::vector<unsigned char> >::iterator it = g_buffers.find(g_mapped);
        if (it != g_buffers.end())
            if (FILE *f = fopen(path, "wb")) { fwrite(it->second.data(), 1, it->second.size(), f); fclose(f); }
    }
    ptmi355_adaptor_reset();
    return cudaSuccess;
}

}  // extern "C"
