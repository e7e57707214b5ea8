// Weighted least-squares phase unwrap (a7): workspace + driver interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace gpa {

struct UnwrapWorkspace {
  int dtype = 0, n0 = 0, n1 = 0;
  void* impl = nullptr;
};

hipError_t unwrap_workspace_create(int dtype, int n0, int n1, hipStream_t s, UnwrapWorkspace* ws,
                                   size_t* bytes_out);
void unwrap_workspace_destroy(UnwrapWorkspace* ws);

// prediff == false: a = dx (n0 x (n1-1)), b = dy ((n0-1) x n1)
// prediff == true : a = psi (n0 x n1), b ignored (differences taken on the device)
// All pointers are device pointers of the workspace dtype.  Synchronises the
// stream once at the end to fetch the iteration count.
hipError_t unwrap_run(UnwrapWorkspace* ws, const void* a, const void* b, const void* weight,
                      bool from_psi, int kmax, double eps, bool axes_compat, void* phi,
                      int* iters_out, hipStream_t s);

}  // namespace gpa
