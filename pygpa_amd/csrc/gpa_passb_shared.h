// Interface of the shared-forward pass B (gpa_passb_shared.hip): best-of-K selection with ONE forward transform per
// x-plane row, a per-candidate shifted real Gaussian, and the exact end fix-up as a small Hankel contraction on the
// matrix cores.  Replaces, for the sweep, the per-candidate forward transforms of gpa_passb.h; same selection rule
// (geometric_phase_analysis.py:679-684), same outputs.
#pragma once
#include "gpa_internal.h"

namespace gpa {

// candidates whose end fix-up is contracted in one matrix pass (16 columns = 2 ends x NC x (re, im); f64 uses half of
// the columns so that two workgroups still share a CU's LDS at 4096 points)
template <class T> struct PassBSharedNC { static constexpr int value = sizeof(T) == 4 ? 4 : 2; };

// Device tables, element types of the plan dtype (T real, cpx<T> complex):
struct PassBSharedTables {
  void* Gb;     // [B][NBL * L / 16] T  shifted Gaussian of candidate b, its live registers in the spectral layout, / L
  void* psi;    // [B][Epad] cpx (phi_b - [periodic]) exp(2 pi i wy_b (a0 + 1)): post-factor of the end fix
  void* dyc;    // [P][n1] cpx   exp(2 pi i ky y): compensation of the winner, candidate-independent
  void* gtab;   // [2 Epad + 16] T   Hankel taps: gtab[s] = g(s) for 1 <= s <= E, 0 elsewhere
  void* pre;    // [B][Epad] cpx  exp(2 pi i wy'_b j): pre-factor of the end strips (wy' = wy + band rotation)
  void* rot16;  // [P][16] cpx    exp(-2 pi i s_p r / 16): the band rotation of peak p at column r mod 16
  int* order;   // [B] the kernel visits the candidates of a peak in the order order[pt K + 0 .. K-1] (absolute indices into the
                //   staged list: x-plane, compensation phasor and the reported kidx are those of the ORIGINAL position);
                //   every other table here is laid out in visiting order
  int* desc;    // [B] per candidate: bit 0 = first of a run of candidates on one x-plane (read the row, forward
                //   transform), bit 1 = first of a chunk of <= NC candidates (matrix pass), bits 2-3 = slot in the
                //   chunk, bits 4-6 = candidates in the chunk, bit 7 = parity of the chunk
};

// can the shared pass B run this axis (3-pass transforms, taps that vanish beyond E <= L / 16 samples, LDS)?
bool passB_shared_supports(int dtype, const Axis& a1, int E);
int passB_shared_elems(int dtype, const Axis& a1);

// wys: device [B] doubles wy_b + shift(peak) / 16; kr: device [B][2] peaks of the candidates; shifts: device [P] ints, the
// band rotation of every peak in blocks of L / 16 bins; taps: device doubles g(0 .. Etab) of the length-n circular filter;
// nbl: live spectral registers (passB_shared_nbl)
hipError_t launch_shared_tables(int dtype, const Axis& a1, const double* wys, const double* kr, const int* shifts,
                                const double* taps, int Etab, int E, int Epad, int B, int K, int nbl, const PassBSharedTables& st,
                                hipStream_t s, int elems = 16);
int passB_shared_nbl(int dtype, int need);

// elems: elements per thread of the row transform (16; 8 exists for 4096-point rows), the same value in
// launch_shared_tables (the candidate tables are laid out for it).
// a1: the shared kernel's own geometry of the y axis (periodic, or zero-padded to L >= n + E); tw1: twiddles of a1.L
hipError_t launch_passB_shared(int dtype, const Axis& a1, int n0, const void* Tbuf, const void* tw1, const SweepTables& tb,
                               const PassBSharedTables& st, int E, int Epad, int P, int K, void* out, int32_t* kidx,
                               hipStream_t s, int nimg = 1, int Bx = 0, int elems = 16, int nbl = 16, bool raw = false);
hipError_t launch_passB_shared_phases(int dtype, const Axis& a1, int n0, const void* Tbuf, const void* tw1, const SweepTables& tb,
                                      const PassBSharedTables& st, int E, int Epad, int P, int K, void* out, int32_t* kidx,
                                      void* psi_out, hipStream_t s, int Bx, int elems, int nbl);
// raw: the winners are left WITHOUT the candidate-independent compensation exp(2 pi i (ky + s_p / 16) y) (no second visit
// of the rows: 0.8 GB less traffic at 4096^2 x 3); the consumer adds its phase step along y (launch_reconstruct_setup)

// (Round 3's opt-in pass A with the forward transform shared by all x-planes of a column -- measured slower, 1.0-1.2 against
//  0.815 ms: profiles/r03_passA_shared.txt -- was removed in round 5; git history has it.)

}  // namespace gpa
