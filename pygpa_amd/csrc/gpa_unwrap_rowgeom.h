// Geometry of the packed-pair row kernels of the fused PCG iteration (gpa_unwrap_rows.hip, gpa_unwrap_pqdct.hip) and the
// register / occupancy choices their translation units share.
#pragma once
#include "gpa_unwrap_impl.h"

namespace gpa {
namespace {

#ifndef GPA_ROW_TWLDS
#define GPA_ROW_TWLDS 1   // 16-element three-pass row transforms: pass-1 base twiddles from a small LDS table (12 VGPRs less in f32)
#endif
template <class T, int LG, bool LAT = false>
struct RowGeom {
  using F = WgFFT<T, LG, unwrap_elems(LG, sizeof(T))>;
  static constexpr bool TWLDS = GPA_ROW_TWLDS && F::E == 16 && F::P == 3;
  using TW = typename std::conditional<TWLDS, typename F::TwiddlesP1Lds, typename F::Twiddles>::type;
  static constexpr int T1N = TWLDS ? F::P1_SETS * 6 : 1;
  using D = WgDCT<T, LG, unwrap_elems(LG, sizeof(T))>;
  // threads per workgroup: 256; the latency-tuned kernels of ONE image with rows up to 512 pixels take 128 (twice the
  // workgroups on a GPU that such an image leaves mostly empty: 512^2 893 -> 935 Mpix/s; stacks prefer 256)
#ifndef GPA_ROW_THREADS_LAT
#define GPA_ROW_THREADS_LAT 128
#endif
#ifndef GPA_ROW_THREADS
#define GPA_ROW_THREADS 256
#endif
  // (rows up to 256 pixels: one wavefront per workgroup, 256^2 280 -> 290 Mpix/s; at 512 that loses 8 %)
  static constexpr int WGT = (LAT && LG <= 8) ? 64 : (LAT && LG == 9) ? GPA_ROW_THREADS_LAT : GPA_ROW_THREADS;
  static constexpr int NF = F::TPF >= WGT ? 1 : WGT / F::TPF;   // row PAIRS per workgroup
  static constexpr int RS = F::LDS_ELEMS + (NF > 1 ? (F::TPF < 32 ? F::TPF : 0) : 0);
  static constexpr int THREADS = NF * F::TPF;
  static constexpr size_t LDS_BYTES = (size_t)NF * RS * sizeof(cpx<T>);
  static constexpr bool FITS = LDS_BYTES <= 160 * 1024;
};
#ifndef GPA_F64_WAVES
#define GPA_F64_WAVES 2   // f64 row kernels: 2 waves/SIMD (256 VGPRs) beat 1 wave with AGPR spill-over
#endif

}  // namespace
}  // namespace gpa
