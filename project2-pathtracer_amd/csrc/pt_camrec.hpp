// pt_camrec.hpp -- per-frame camera constants handed to the kernels by value.
#pragma once

namespace ptd {

// = host half of raycastFromCameraKernel (/root/reference/src/raytraceKernel.cu:47-60)
struct CamRec {
    float E[3], M[3], H[3], V[3];
    float Cn[3], Ah[3], Bh[3];   // unit view / right / up' (thin lens only)
    float wm1, hm1;              // resolution.x - 1, resolution.y - 1
    float aperture, focal;
    int W, Hh;
    int camera_mode, antialias;
    int row_offset, row_stride;
    // exact division of a pixel index (< 2^28) by W and of a row number by row_stride as multiply + shift
    // (q = (n * m) >> sh, m = floor(2^sh / d) + 1, sh = 28 + ceil(log2 d)): global pixel -> index among the owned pixels
    unsigned int mW, shW, mS, shS;
};

}  // namespace ptd
