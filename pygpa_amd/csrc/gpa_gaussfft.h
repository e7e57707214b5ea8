// f-3: scipy.ndimage.gaussian_filter along one axis as an overlap-save FFT convolution (gpa_gaussfft.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace gpa {

// log2 of the segment transform length for an axis of n samples and a kernel of radius R (0: no length fits, use the direct sum)
int gaussfft_choose_lg(int dtype, int n, int R, bool cols);
// transfer function of the 2R + 1 normalised taps w, / L, in the register engine's spectral layout (doubles)
std::vector<double> gaussfft_table(int lg, const std::vector<double>& w);
// out = [minuend -] gaussian_filter1d(in) along `axis` (0: columns, 1: rows; minuend only with axis 1); H, tw: device tables
hipError_t launch_gaussfft(int dtype, int lg, int axis, const void* in, void* out, int n0, int n1, int R, const void* H,
                           const void* tw, const void* minuend, hipStream_t s);

}  // namespace gpa
