/* cutil_gl_inline.h -- headless shim (/root/reference/src/main.h:21): nothing of it is used */
#ifndef PTMI355_SHIM_CUTIL_GL_INLINE_H
#define PTMI355_SHIM_CUTIL_GL_INLINE_H
#endif
