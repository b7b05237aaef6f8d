// oracle/ref_fft_wrapper.cpp -- TEST INFRASTRUCTURE.  A C entry point around the reference's OWN 3D FFT, so that the oracle's FFT
// (oracle/snb_oracle.c, orc_fft3d) and the engine's (snb_test_fft3d) can be checked against the code the reference runs.
//
// The reference's Reference-platform PME transforms its grids with pocketfft::c2c from the header it vendors,
// openmmapi/include/internal/pocketfft_hdronly.h (call sites: platforms/reference/src/ReferencePME.cpp:788-805, 848-865; the
// reference's own GPU FFT tests use the same call as their oracle, platforms/cuda/tests/TestCudaCuFFT3D.cpp:97-103).  That header is
// self-contained standard C++, so this one piece of the reference DOES build here: oracle/Makefile compiles this file against the
// header where it lies under /root/reference (nothing of it is copied into the repository) into oracle/_ref/libref_fft.so.
// Everything else on the path needs OpenMM headers and stays restated.
#include <complex>
#include <cstddef>
#include <vector>
#include "internal/pocketfft_hdronly.h"

extern "C" {
// In-place complex 3D transform of `batch` row-major [nx][ny][nz] grids of interleaved (re, im) doubles, called exactly as
// ReferencePME.cpp:788-796 calls it (shape, byte strides, axes {0,1,2}, forward flag, factor 1.0 -- unnormalised --, default threads).
int ref_c2c_3d(double* data, int batch, int nx, int ny, int nz, int forward) {
    using std::complex;
    std::vector<size_t> shape = {(size_t) nx, (size_t) ny, (size_t) nz};
    std::vector<size_t> axes = {0, 1, 2};
    std::vector<ptrdiff_t> stride = {(ptrdiff_t) (ny * nz * sizeof(complex<double>)), (ptrdiff_t) (nz * sizeof(complex<double>)), (ptrdiff_t) sizeof(complex<double>)};
    complex<double>* grid = reinterpret_cast<complex<double>*>(data);
    for (int i = 0; i < batch; i++) {
        complex<double>* g = grid + (size_t) i * nx * ny * nz;
        pocketfft::c2c(shape, stride, stride, axes, forward != 0, g, g, 1.0, 0);
    }
    return 0;
}
}
