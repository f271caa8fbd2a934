// pt_host.hpp -- host-side helpers shared by the HIP TU and the scene/image TU.
#pragma once

#include "../../include/ptmi355.h"
#include "pt_camrec.hpp"

namespace pth {

// printf-style; stored per thread, returned by pt_last_error()
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

// host half of raycastFromCameraKernel (/root/reference/src/raytraceKernel.cu:47-60): the
// per-frame basis E, M, H, V evaluated once in binary32 with the host libm's tanf, exactly as
// the reference's host code path would, + the build's thin-lens unit vectors.
void camera_basis(const pt_camera *cam, const pt_config *cfg, ptd::CamRec *out);

}  // namespace pth
