/* cutil_inline.h -- headless shim (/root/reference/src/main.h:20): main.cpp:292 asks for the fastest device's ordinal */
#ifndef PTMI355_SHIM_CUTIL_INLINE_H
#define PTMI355_SHIM_CUTIL_INLINE_H
static inline int cutGetMaxGflopsDeviceId(void) { return 0; }       /* the adaptor picks its device from PT_DEVICE */
#endif
