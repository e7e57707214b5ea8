// Weighted least-squares phase unwrap (a7): workspace + driver interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace gpa {

struct UnwrapWorkspace {
  int dtype = 0, n0 = 0, n1 = 0;
  void* impl = nullptr;
};

// nprob > 1: a workspace for nprob independent solves on images of one shape that run as ONE set of launches
// (blockIdx.z = problem; prepared starts only).  Problem pb uses the pb-th slice of every buffer; problems 2i and
// 2i + 1 share the weight of image i (the two displacement components of a batched driver call).
hipError_t unwrap_workspace_create(int dtype, int n0, int n1, hipStream_t s, UnwrapWorkspace* ws,
                                   size_t* bytes_out, int nprob = 1);
void unwrap_workspace_destroy(UnwrapWorkspace* ws);
// whether the workspace's image shape runs the fused iteration, the only one that takes several problems per launch
bool unwrap_supports_batch(const UnwrapWorkspace* ws);

// prediff == false: a = dx (n0 x (n1-1)), b = dy ((n0-1) x n1)
// prediff == true : a = psi (n0 x n1), b ignored (differences taken on the device)
// All pointers are device pointers of the workspace dtype.  Synchronises the
// stream once at the end to fetch the iteration count.
hipError_t unwrap_run(UnwrapWorkspace* ws, const void* a, const void* b, const void* weight,
                      bool from_psi, int kmax, double eps, bool axes_compat, void* phi,
                      int* iters_out, hipStream_t s);

// the two halves of unwrap_run, for callers that overlap several unwraps on different streams:
// enqueue everything without synchronising, then fetch the iteration count (synchronises s)
hipError_t unwrap_enqueue(UnwrapWorkspace* ws, const void* a, const void* b, const void* weight, bool from_psi,
                          int kmax, double eps, bool axes_compat, void* phi, hipStream_t s);
// prepared start: the caller has written r0 = div(W^2 wrap(grad)) into unwrap_residual_buffer() and
// nparts partial sums of ||r0||^2 into unwrap_partials_buffer() (reconstruct_setup_kernel does, fused with
// the per-pixel least squares that produces the gradients); weight as in unwrap_enqueue
void* unwrap_residual_buffer(UnwrapWorkspace* ws, int problem = 0);
double* unwrap_partials_buffer(UnwrapWorkspace* ws, int problem = 0);
hipError_t unwrap_enqueue_prepared(UnwrapWorkspace* ws, const void* weight, int nparts, int kmax, double eps,
                                   bool axes_compat, void* phi, hipStream_t s);
hipError_t unwrap_finish(UnwrapWorkspace* ws, int* iters_out, hipStream_t s);
// solve the first n problems of a workspace created for more (false: n exceeds its capacity)
bool unwrap_set_active(UnwrapWorkspace* ws, int n);
// asynchronous copy of the iteration count into (pinned) host memory, no synchronisation
// (a batched workspace: nprob counts, 4 ints apart -- host_pinned[4 * pb])
hipError_t unwrap_fetch_iters(UnwrapWorkspace* ws, int* host_pinned, hipStream_t s);
// several problems per workspace: unwrap_fetch_iters() copies 4 flag words per problem, the count is word unwrap_iters_slot()
int unwrap_iters_slot(const UnwrapWorkspace* ws);

}  // namespace gpa
