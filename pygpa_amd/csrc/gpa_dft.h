// Plain 2-D DFTs of any image size (a9 `moisan2011.per`, call site geometric_phase_analysis.py:429; f-3 peak finding
// :427-438; f-4 gaussian_deconvolve :892-904) and the a9 kernels that run on them.  gpa_dft2.hip.
//
// One axis of length n is transformed by one of three engines, chosen when the axis is created:
//   DFT_POW2  n = 2^lg, 64 <= n, a row fits one workgroup: the register engine of gpa_fft.h at length n itself.  Rows of a REAL
//             image go two at a time through one complex transform (split by Hermitian symmetry); columns run on tiles of C
//             adjacent columns (row pieces of C complex values, two columns per thread in f32).
//   DFT_BLUE  any n with 2n - 1 <= 2^lg that fits one workgroup: Bluestein's chirp-z on the register engine (round 1).
//   DFT_BIG   everything else, up to n = 65536: chirp-z whose length-L convolution (L = 2^lg1 * 2^lg2 >= 2n - 1) is a TWO-LEVEL
//             transform through HBM -- strided sub-transforms of length 2^lg1 with the twiddle, then forward * table * inverse of
//             length 2^lg2 on contiguous pieces, then the mirror image of the first step.  No transform ever needs more than
//             512 points in one workgroup, so neither LDS nor the register file bound n.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace gpa {

enum DftKind { DFT_NONE = 0, DFT_POW2 = 1, DFT_BLUE = 2, DFT_BIG = 3 };

struct DftAxis {
  int n = 0, kind = DFT_NONE;
  int lg = 0;                     // log2 of the transform length: n (POW2) or L (BLUE, BIG)
  int lg1 = 0, lg2 = 0;           // BIG: L = 2^lg1 * 2^lg2 (strided level, contiguous level)
  void *tw = nullptr;             // POW2 / BLUE: exp(-2 pi i t / 2^lg); BIG: the same table of length L (level twiddles)
  void *chirp = nullptr;          // BLUE / BIG: c_m = exp(i pi m^2 / n), m < n
  void *bspec = nullptr;          // BLUE: FFT_L(b) / L in the register engine's spectral layout; BIG: in the two-level layout
  void *tw1 = nullptr, *tw2 = nullptr;   // BIG: twiddles of the two sub-transform lengths
};

// scratch of the BIG engine (lines x L complex), kept by the plan and grown on demand
struct DftWork {
  void* buf = nullptr;
  size_t cap = 0;
  size_t* counted = nullptr;      // where the owner accounts its bytes (gpa_plan::ws_bytes), may be null
};
void dft_work_free(DftWork* w);

// force: 0 = choose, DFT_BLUE / DFT_BIG = that engine if it can take n (tests); returns hipErrorInvalidValue if n is out of range
hipError_t dft_axis_create(int dtype, int n, hipStream_t s, DftAxis* out, size_t* bytes, int force = 0);
void dft_axis_destroy(DftAxis* a);
const char* dft_kind_name(int kind);

// forward DFT of a complex n0 x n1 array in place (natural order in and out)
hipError_t dft2_inplace(int dtype, const DftAxis& a0, const DftAxis& a1, void* Z, DftWork* w, hipStream_t s);
// forward DFT of a REAL n0 x n1 image into the complex array Z
hipError_t dft2_forward_real(int dtype, const DftAxis& a0, const DftAxis& a1, const void* image, void* Z, DftWork* w,
                             hipStream_t s);
// the same, bins [0, n1/2] of every row only, row pitch dft2_half_pitch(a1) (0: this axis' engine cannot; power-of-two rows can)
size_t dft2_half_pitch(const DftAxis& a1);
hipError_t dft2_forward_real_half(int dtype, const DftAxis& a0, const DftAxis& a1, const void* image, void* Z, DftWork* w,
                                  hipStream_t s);
// forward DFT of `rows` contiguous complex rows of length a.n, in place
hipError_t dft_rows_inplace(int dtype, const DftAxis& a, int rows, void* Z, DftWork* w, hipStream_t s);

// a9 (Moisan 2011 periodic + smooth decomposition): per_borders writes the two border-difference vectors d0 (length n1),
// d1 (length n0) as complex values; per_combine forms P^ = U^ - V^ / (2 cos(2 pi q / n0) + 2 cos(2 pi r / n1) - 4) with V^
// from D0 = DFT(d0), D1 = DFT(d1)
hipError_t per_borders(int dtype, const void* image, int n0, int n1, void* d0, void* d1, hipStream_t s);
hipError_t per_smooth_hat(int dtype, void* Uhat_inout, const void* Phat, size_t n, hipStream_t s);
hipError_t per_components(int dtype, const DftAxis& a0, const DftAxis& a1, void* Phat_destroyed, const void* image,
                          void* p_out, void* s_out, DftWork* w, hipStream_t s);
// `tab`: per_tables_bytes() of device memory filled once by per_tables_fill (the per-axis factors, in double).
// abs_shift: out = |fftshift(P^)| as reals with the DC bin exactly 0 (what f-3 smooths) instead of P^ itself
size_t per_tables_bytes(int dtype, int n0, int n1);
hipError_t per_tables_fill(int dtype, int n0, int n1, void* tab, hipStream_t s);
// half_pitch != 0 (abs_shift only): Uhat holds bins 0 ... n1/2 of every row with that row pitch (dft2_forward_real_half)
hipError_t per_combine(int dtype, const void* Uhat, const void* D0, const void* D1, int n0, int n1, const void* tab,
                       bool abs_shift, void* out, hipStream_t s, size_t half_pitch = 0);

}  // namespace gpa
