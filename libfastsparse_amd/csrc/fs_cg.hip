// fs_cg.hip -- the consumers of the A_mul_B path, device resident:
// conjugate gradients on (A'A + lambda I) with one right-hand side (bsbm_cg, cg.h:25-82) and with two
// row-major right-hand sides (bsbm_cg2, cg.h:85-187).  Every vector lives in HBM for the whole solve; per
// iteration two products (fs_spmv / fs_spmm on A and A') and three fused vector kernels run.  The scalars of the
// iteration (alpha, beta, r.r; for two right-hand sides the 2x2 algebra of solve2sym, linalg.h:77-88) are computed
// ON THE DEVICE by the one-workgroup kernel that finishes each reduction, with the reference's formulas, and stay
// there: nothing in an iteration waits for the host.  The host only has to learn WHEN to stop enqueuing: the
// "done" flag of iteration i is copied to pinned memory behind it and looked at while iteration i + 1 runs; the
// vector kernels of iterations enqueued past the end see the flag and do nothing (their products are wasted
// work: at most two iterations' worth).  Round 2 fetched every reduction to the host: two stream
// synchronisations per iteration, 0.1-0.2 ms of idle GPU in a 1.5 ms iteration.
//
// Reductions are two-stage with a fixed shape (1024 workgroup partials, then one workgroup), so results are
// reproducible run to run; they are NOT the CPU's single left-to-right sums, so iterates agree with the
// reference to rounding, not bit for bit.
#include <math.h>

#include <vector>

#include "fs_common.h"

namespace fs {

constexpr int kRedBlocks = 1024;
constexpr int kRedThreads = 256;

// sum of NV values per thread over the workgroup -> part[blockIdx * NV + j]
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *__restrict__ part)
{
  __shared__ double sm[NV][kRedThreads / 64];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    double s = v[j];
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) sm[j][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
    for (int w = 0; w < kRedThreads / 64; ++w) s += sm[threadIdx.x][w];
    part[blockIdx.x * NV + threadIdx.x] = s;
  }
}

template <int NV>
__global__ __launch_bounds__(kRedThreads) void final_sum_kernel(const double *__restrict__ part, int nblocks,
                                                               double *__restrict__ out)
{
  double v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j] = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += kRedThreads) v[j] += part[b * NV + j];
  }
  block_sum<NV>(v, out);  // gridDim == 1: out[0..NV)
}

// ---- one right-hand side --------------------------------------------------------------------------------
// x = 0, r = p = b, partial b.b                                   (cg.h:46-51)
__global__ __launch_bounds__(kRedThreads) void cg_init_kernel(int n, const double *__restrict__ b, double *__restrict__ x,
                                                             double *__restrict__ r, double *__restrict__ p,
                                                             double *__restrict__ part)
{
  double v[1] = {0.0};
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    const double bi = b[i];
    x[i] = 0.0; r[i] = bi; p[i] = bi;
    v[0] += bi * bi;
  }
  block_sum<1>(v, part);
}

// q += lambda p, partial q.p                                      (cg.h:17-21, :59)
__global__ __launch_bounds__(kRedThreads) void cg_shift_dot_kernel(int n, double lambda, double *__restrict__ q,
                                                                  const double *__restrict__ p, double *__restrict__ part)
{
  double v[1] = {0.0};
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    const double pi = p[i];
    const double qi = q[i] + lambda * pi;
    q[i] = qi;
    v[0] += qi * pi;
  }
  block_sum<1>(v, part);
}

// x += alpha p, r -= alpha q, partial r.r                         (cg.h:61-67)
__global__ __launch_bounds__(kRedThreads) void cg_update_kernel(int n, double alpha, double *__restrict__ x,
                                                               double *__restrict__ r, const double *__restrict__ p,
                                                               const double *__restrict__ q, double *__restrict__ part)
{
  double v[1] = {0.0};
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * q[i];
    r[i] = ri;
    v[0] += ri * ri;
  }
  block_sum<1>(v, part);
}

// p = r + beta p                                                  (cg.h:71-75)
__global__ __launch_bounds__(kRedThreads) void cg_direction_kernel(int n, double beta, double *__restrict__ p,
                                                                  const double *__restrict__ r)
{
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) p[i] = r[i] + beta * p[i];
}

// ---- two right-hand sides, row-major -------------------------------------------------------------------
// partial {a'a, b'b, a'b} of X with Y                             (pnormsq2 / pouter2 / pdot2sym, linalg.h:24-73)
__global__ __launch_bounds__(kRedThreads) void cg2_dot_kernel(int n, const double *__restrict__ X,
                                                             const double *__restrict__ Y, double *__restrict__ part)
{
  double v[3] = {0.0, 0.0, 0.0};
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    const double xa = X[2 * i], xb = X[2 * i + 1], ya = Y[2 * i], yb = Y[2 * i + 1];
    v[0] += xa * ya; v[1] += xb * yb; v[2] += xa * yb;
  }
  block_sum<3>(v, part);
}

// X = 0, R = P = B * inorms, partial R'R                          (cg.h:113-125)
__global__ __launch_bounds__(kRedThreads) void cg2_init_kernel(int n, double in0, double in1, const double *__restrict__ B,
                                                              double *__restrict__ X, double *__restrict__ R,
                                                              double *__restrict__ P, double *__restrict__ part)
{
  double v[3] = {0.0, 0.0, 0.0};
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    const double a = B[2 * i] * in0, c = B[2 * i + 1] * in1;
    X[2 * i] = 0.0; X[2 * i + 1] = 0.0;
    R[2 * i] = a; R[2 * i + 1] = c; P[2 * i] = a; P[2 * i + 1] = c;
    v[0] += a * a; v[1] += c * c; v[2] += a * c;
  }
  block_sum<3>(v, part);
}

// Q += lambda P, partial P'Q (symmetric form)                      (cg.h:136-142)
__global__ __launch_bounds__(kRedThreads) void cg2_shift_dot_kernel(int n, double lambda, double *__restrict__ Q,
                                                                   const double *__restrict__ P, double *__restrict__ part)
{
  double v[3] = {0.0, 0.0, 0.0};
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    const double pa = P[2 * i], pb = P[2 * i + 1];
    const double qa = Q[2 * i] + lambda * pa, qb = Q[2 * i + 1] + lambda * pb;
    Q[2 * i] = qa; Q[2 * i + 1] = qb;
    v[0] += pa * qa; v[1] += pb * qb; v[2] += pa * qb;
  }
  block_sum<3>(v, part);
}

// X += Alpha' P, R -= Alpha' Q, partial R'R                        (cg.h:148-157)
__global__ __launch_bounds__(kRedThreads) void cg2_update_kernel(int n, double a0, double a1, double a2, double a3,
                                                                double *__restrict__ X, double *__restrict__ R,
                                                                const double *__restrict__ P, const double *__restrict__ Q,
                                                                double *__restrict__ part)
{
  double v[3] = {0.0, 0.0, 0.0};
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    const double pa = P[2 * i], pb = P[2 * i + 1], qa = Q[2 * i], qb = Q[2 * i + 1];
    X[2 * i] += a0 * pa + a1 * pb;
    X[2 * i + 1] += a2 * pa + a3 * pb;
    const double ra = R[2 * i] - (a0 * qa + a1 * qb), rb = R[2 * i + 1] - (a2 * qa + a3 * qb);
    R[2 * i] = ra; R[2 * i + 1] = rb;
    v[0] += ra * ra; v[1] += rb * rb; v[2] += ra * rb;
  }
  block_sum<3>(v, part);
}

// P = R + Psi' P                                                   (cg.h:165-171)
__global__ __launch_bounds__(kRedThreads) void cg2_direction_kernel(int n, double s0, double s1, double s2, double s3,
                                                                   double *__restrict__ P, const double *__restrict__ R)
{
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    const double pa = P[2 * i], pb = P[2 * i + 1];
    P[2 * i] = R[2 * i] + s0 * pa + s1 * pb;
    P[2 * i + 1] = R[2 * i + 1] + s2 * pa + s3 * pb;
  }
}

// y += a x
__global__ __launch_bounds__(kRedThreads) void axpy_kernel(int n, double a, const double *__restrict__ x, double *__restrict__ y)
{
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) y[i] += a * x[i];
}

__global__ __launch_bounds__(kRedThreads) void cg2_scale_kernel(int n, double n0, double n1, double *__restrict__ X)
{
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    X[2 * i] *= n0;
    X[2 * i + 1] *= n1;
  }
}

// ---- the same steps with their scalars in device memory (fs_cg / fs_cg2) ---------------------------------
// state of one solve, doubles.  One right-hand side: rsq_old, alpha, beta, stop; two: RtR[3], Alpha[4], Psi[4], tolsq
enum { kStDone = kCgStateDone, kStIter = kCgStateIter, kStRsq = 2, kStAlpha = 3, kStBeta = 4, kStStop = 5,
       kSt2RtR = 2, kSt2Alpha = 5, kSt2Psi = 9, kSt2Tolsq = 13, kStDoubles = kCgStateDoubles };

__device__ __forceinline__ void solve2sym_dev(double *X, const double *A, const double *RHS)  // linalg.h:77-88
{
  const double dinv = 1.0 / (A[0] * A[1] - A[2] * A[2]);
  const double i0 = dinv * A[1], i1 = dinv * A[0], i2 = -dinv * A[2];
  X[0] = i0 * RHS[0] + i2 * RHS[1];
  X[1] = i2 * RHS[0] + i1 * RHS[1];
  X[2] = i0 * RHS[2] + i2 * RHS[3];
  X[3] = i2 * RHS[2] + i1 * RHS[3];
}

// the one workgroup that finishes a reduction, then does the iteration's scalar step (cg.h:59-76, 143-172):
//   MODE 0  b.b: rsq_old, stop = tol sqrt(b.b), done = 0, iter = 0           (tol in `arg`)
//   MODE 1  p.q: alpha = rsq_old / p.q
//   MODE 2  r.r: converged -> done; else beta = rsq_new / rsq_old, rsq_old = rsq_new, ++iter
//   MODE 3  P'KP: Alpha = solve2sym(P'KP, R'R)
//   MODE 4  R'R new: both <= tol^2 -> done; else Psi = solve2sym(R'R, R'R new), R'R = R'R new, ++iter
template <int NV, int MODE>
__global__ __launch_bounds__(kRedThreads) void final_step_kernel(const double *__restrict__ part, int nblocks,
                                                                double *__restrict__ red, double *__restrict__ st, double arg)
{
  if (MODE != 0 && st[kStDone] != 0.0) return;
  double v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j] = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += kRedThreads) v[j] += part[b * NV + j];
  }
  block_sum<NV>(v, red);
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (MODE == 0) {
    st[kStRsq] = red[0]; st[kStStop] = arg * sqrt(red[0]); st[kStDone] = 0.0; st[kStIter] = 0.0;
  } else if (MODE == 1) {
    st[kStAlpha] = st[kStRsq] / red[0];
  } else if (MODE == 2) {
    const double rsq_new = red[0];
    if (sqrt(rsq_new) <= st[kStStop]) st[kStDone] = 1.0;
    else { st[kStBeta] = rsq_new / st[kStRsq]; st[kStRsq] = rsq_new; st[kStIter] += 1.0; }
  } else if (MODE == 3) {
    const double rhs[4] = {st[kSt2RtR], st[kSt2RtR + 2], st[kSt2RtR + 2], st[kSt2RtR + 1]};
    double a[4];
    solve2sym_dev(a, red, rhs);
    st[kSt2Alpha] = a[0]; st[kSt2Alpha + 1] = a[1]; st[kSt2Alpha + 2] = a[2]; st[kSt2Alpha + 3] = a[3];
  } else {
    const double n0 = red[0], n1 = red[1], n2 = red[2], tolsq = st[kSt2Tolsq];
    if (n0 <= tolsq && n1 <= tolsq) st[kStDone] = 1.0;
    else {
      const double old[3] = {st[kSt2RtR], st[kSt2RtR + 1], st[kSt2RtR + 2]};
      const double rhs[4] = {n0, n2, n2, n1};
      double ps[4];
      solve2sym_dev(ps, old, rhs);
      st[kSt2Psi] = ps[0]; st[kSt2Psi + 1] = ps[1]; st[kSt2Psi + 2] = ps[2]; st[kSt2Psi + 3] = ps[3];
      st[kSt2RtR] = n0; st[kSt2RtR + 1] = n1; st[kSt2RtR + 2] = n2;
      st[kStIter] += 1.0;
    }
  }
}

__global__ __launch_bounds__(kRedThreads) void cg_shift_dot_dev_kernel(int n, double lambda, double *__restrict__ q,
                                                                      const double *__restrict__ p, double *__restrict__ part,
                                                                      const double *__restrict__ st)
{
  if (st[kStDone] != 0.0) return;
  double v[1] = {0.0};
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    const double pi = p[i];
    const double qi = q[i] + lambda * pi;
    q[i] = qi;
    v[0] += qi * pi;
  }
  block_sum<1>(v, part);
}

__global__ __launch_bounds__(kRedThreads) void cg_update_dev_kernel(int n, double *__restrict__ x, double *__restrict__ r,
                                                                   const double *__restrict__ p, const double *__restrict__ q,
                                                                   double *__restrict__ part, const double *__restrict__ st)
{
  if (st[kStDone] != 0.0) return;
  const double alpha = st[kStAlpha];
  double v[1] = {0.0};
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * q[i];
    r[i] = ri;
    v[0] += ri * ri;
  }
  block_sum<1>(v, part);
}

__global__ __launch_bounds__(kRedThreads) void cg_direction_dev_kernel(int n, double *__restrict__ p, const double *__restrict__ r,
                                                                      const double *__restrict__ st)
{
  if (st[kStDone] != 0.0) return;
  const double beta = st[kStBeta];
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) p[i] = r[i] + beta * p[i];
}

__global__ __launch_bounds__(kRedThreads) void cg2_shift_dot_dev_kernel(int n, double lambda, double *__restrict__ Q,
                                                                       const double *__restrict__ P, double *__restrict__ part,
                                                                       const double *__restrict__ st)
{
  if (st[kStDone] != 0.0) return;
  double v[3] = {0.0, 0.0, 0.0};
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    const double pa = P[2 * i], pb = P[2 * i + 1];
    const double qa = Q[2 * i] + lambda * pa, qb = Q[2 * i + 1] + lambda * pb;
    Q[2 * i] = qa; Q[2 * i + 1] = qb;
    v[0] += pa * qa; v[1] += pb * qb; v[2] += pa * qb;
  }
  block_sum<3>(v, part);
}

__global__ __launch_bounds__(kRedThreads) void cg2_update_dev_kernel(int n, double *__restrict__ X, double *__restrict__ R,
                                                                    const double *__restrict__ P, const double *__restrict__ Q,
                                                                    double *__restrict__ part, const double *__restrict__ st)
{
  if (st[kStDone] != 0.0) return;
  const double a0 = st[kSt2Alpha], a1 = st[kSt2Alpha + 1], a2 = st[kSt2Alpha + 2], a3 = st[kSt2Alpha + 3];
  double v[3] = {0.0, 0.0, 0.0};
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    const double pa = P[2 * i], pb = P[2 * i + 1], qa = Q[2 * i], qb = Q[2 * i + 1];
    X[2 * i] += a0 * pa + a1 * pb;
    X[2 * i + 1] += a2 * pa + a3 * pb;
    const double ra = R[2 * i] - (a0 * qa + a1 * qb), rb = R[2 * i + 1] - (a2 * qa + a3 * qb);
    R[2 * i] = ra; R[2 * i + 1] = rb;
    v[0] += ra * ra; v[1] += rb * rb; v[2] += ra * rb;
  }
  block_sum<3>(v, part);
}

__global__ __launch_bounds__(kRedThreads) void cg2_direction_dev_kernel(int n, double *__restrict__ P, const double *__restrict__ R,
                                                                       const double *__restrict__ st)
{
  if (st[kStDone] != 0.0) return;
  const double s0 = st[kSt2Psi], s1 = st[kSt2Psi + 1], s2 = st[kSt2Psi + 2], s3 = st[kSt2Psi + 3];
  for (int i = blockIdx.x * kRedThreads + threadIdx.x; i < n; i += gridDim.x * kRedThreads) {
    const double pa = P[2 * i], pb = P[2 * i + 1];
    P[2 * i] = R[2 * i] + s0 * pa + s1 * pb;
    P[2 * i + 1] = R[2 * i + 1] + s2 * pa + s3 * pb;
  }
}

struct Workspace {
  std::vector<void *> bufs;
  double *get(size_t n)
  {
    void *p = nullptr;
    if (hipMalloc(&p, sizeof(double) * (n ? n : 1)) != hipSuccess) return nullptr;
    bufs.push_back(p);
    return (double *)p;
  }
  ~Workspace() { for (void *p : bufs) (void)hipFree(p); }
};

// ---- the vector steps of CG as launchers, for callers that bring their own products (fs_dist_cg: the same steps, replicated,
// on every device of a row-sharded matrix -- same kernels on the same data, so every device takes the same decisions).
// part: kCgPartDoubles doubles of scratch, red: 4 doubles, st: kCgStateDoubles doubles (st[kCgStateDone], st[kCgStateIter]).
int cg_dev_init(int n, const double *b, double *x, double *r, double *p, double *part, double *red, double *st, double tol,
                hipStream_t s)
{
  hipLaunchKernelGGL(cg_init_kernel, dim3(kRedBlocks), dim3(kRedThreads), 0, s, n, b, x, r, p, part);
  hipLaunchKernelGGL((final_step_kernel<1, 0>), dim3(1), dim3(kRedThreads), 0, s, part, kRedBlocks, red, st, tol);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// behind q = A'(A p): q += lambda p, alpha, x and r, the convergence test and beta, the new p -- all on the device
int cg_dev_steps(int n, double lambda, double *x, double *r, double *p, double *q, double *part, double *red, double *st,
                 hipStream_t s)
{
  const dim3 g(kRedBlocks), blk(kRedThreads), one(1);
  hipLaunchKernelGGL(cg_shift_dot_dev_kernel, g, blk, 0, s, n, lambda, q, p, part, st);
  hipLaunchKernelGGL((final_step_kernel<1, 1>), one, blk, 0, s, part, kRedBlocks, red, st, 0.0);   // alpha
  hipLaunchKernelGGL(cg_update_dev_kernel, g, blk, 0, s, n, x, r, p, q, part, st);
  hipLaunchKernelGGL((final_step_kernel<1, 2>), one, blk, 0, s, part, kRedBlocks, red, st, 0.0);   // converged? beta
  hipLaunchKernelGGL(cg_direction_dev_kernel, g, blk, 0, s, n, p, r, st);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int cg_dev_init_partial(int n, const double *b, double *x, double *r, double *p, double *part, double *red_out, hipStream_t s)
{
  hipLaunchKernelGGL(cg_init_kernel, dim3(kRedBlocks), dim3(kRedThreads), 0, s, n, b, x, r, p, part);
  hipLaunchKernelGGL(final_sum_kernel<1>, dim3(1), dim3(kRedThreads), 0, s, part, kRedBlocks, red_out);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int cg_dev_step_a(int n, double lambda, const double *p, double *q, double *part, double *red_out, const double *st, hipStream_t s)
{
  hipLaunchKernelGGL(cg_shift_dot_dev_kernel, dim3(kRedBlocks), dim3(kRedThreads), 0, s, n, lambda, q, p, part, st);
  hipLaunchKernelGGL(final_sum_kernel<1>, dim3(1), dim3(kRedThreads), 0, s, part, kRedBlocks, red_out);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int cg_dev_step_b(int n, double *x, double *r, const double *p, const double *q, double *part, double *red_out, const double *st,
                  hipStream_t s)
{
  hipLaunchKernelGGL(cg_update_dev_kernel, dim3(kRedBlocks), dim3(kRedThreads), 0, s, n, x, r, p, q, part, st);
  hipLaunchKernelGGL(final_sum_kernel<1>, dim3(1), dim3(kRedThreads), 0, s, part, kRedBlocks, red_out);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int cg_dev_step_c(int n, double *p, const double *r, const double *st, hipStream_t s)
{
  hipLaunchKernelGGL(cg_direction_dev_kernel, dim3(kRedBlocks), dim3(kRedThreads), 0, s, n, p, r, st);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int cg_dev_final(int mode, const double *partials, int count, double *red_out, double *st, double arg, hipStream_t s)
{
  const dim3 one(1), blk(kRedThreads);
  if (mode == 0)      hipLaunchKernelGGL((final_step_kernel<1, 0>), one, blk, 0, s, partials, count, red_out, st, arg);
  else if (mode == 1) hipLaunchKernelGGL((final_step_kernel<1, 1>), one, blk, 0, s, partials, count, red_out, st, arg);
  else                hipLaunchKernelGGL((final_step_kernel<1, 2>), one, blk, 0, s, partials, count, red_out, st, arg);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// ---- the same for two right-hand sides (fs_cg2, fs_dist_cg2).  cg2_dev_init is synchronous (two reductions go to the host: the
// norms of B's columns scale the system, cg.h:105-125); norms[2] is returned for the final cg2_dev_finish
int cg2_dev_init(int n, const double *B, double *X, double *R, double *P, double *part, double *red, double *st, double tol,
                 double *norms, hipStream_t s)
{
  const dim3 g(kRedBlocks), blk(kRedThreads);
  double h[3], RtR[3];
  hipLaunchKernelGGL(cg2_dot_kernel, g, blk, 0, s, n, B, B, part);
  hipLaunchKernelGGL(final_sum_kernel<3>, dim3(1), blk, 0, s, part, kRedBlocks, red);
  FS_HIP(hipGetLastError());
  FS_HIP(hipMemcpyAsync(h, red, sizeof(h), hipMemcpyDeviceToHost, s));
  FS_HIP(hipStreamSynchronize(s));
  norms[0] = sqrt(h[0]); norms[1] = sqrt(h[1]);
  hipLaunchKernelGGL(cg2_init_kernel, g, blk, 0, s, n, 1.0 / norms[0], 1.0 / norms[1], B, X, R, P, part);
  hipLaunchKernelGGL(final_sum_kernel<3>, dim3(1), blk, 0, s, part, kRedBlocks, red);
  FS_HIP(hipGetLastError());
  FS_HIP(hipMemcpyAsync(RtR, red, sizeof(RtR), hipMemcpyDeviceToHost, s));
  FS_HIP(hipStreamSynchronize(s));
  double st0[kStDoubles] = {0.0};
  st0[kSt2RtR] = RtR[0]; st0[kSt2RtR + 1] = RtR[1]; st0[kSt2RtR + 2] = RtR[2]; st0[kSt2Tolsq] = tol * tol;
  FS_HIP(hipMemcpyAsync(st, st0, sizeof(st0), hipMemcpyHostToDevice, s));
  FS_HIP(hipStreamSynchronize(s));            // (st0 is on this stack frame)
  return FS_OK;
}

// everything of a block-CG iteration behind Q = A'(A P): Q += lambda P, Alpha, X and R, the convergence test and Psi, the new P
int cg2_dev_steps(int n, double lambda, double *X, double *R, double *P, double *Q, double *part, double *red, double *st, hipStream_t s)
{
  const dim3 g(kRedBlocks), blk(kRedThreads), one(1);
  hipLaunchKernelGGL(cg2_shift_dot_dev_kernel, g, blk, 0, s, n, lambda, Q, P, part, st);
  hipLaunchKernelGGL((final_step_kernel<3, 3>), one, blk, 0, s, part, kRedBlocks, red, st, 0.0);   // Alpha
  hipLaunchKernelGGL(cg2_update_dev_kernel, g, blk, 0, s, n, X, R, P, Q, part, st);
  hipLaunchKernelGGL((final_step_kernel<3, 4>), one, blk, 0, s, part, kRedBlocks, red, st, 0.0);   // converged? Psi
  hipLaunchKernelGGL(cg2_direction_dev_kernel, g, blk, 0, s, n, P, R, st);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int cg2_dev_finish(int n, const double *norms, double *X, hipStream_t s)                          // X back in B's scale (cg.h:175-181)
{
  hipLaunchKernelGGL(cg2_scale_kernel, dim3(kRedBlocks), dim3(kRedThreads), 0, s, n, norms[0], norms[1], X);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int CgFlags::init()
{
  FS_HIP(hipHostMalloc((void **)&h, sizeof(double) * 4));
  h[0] = h[1] = h[2] = h[3] = 0.0;
  FS_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
  FS_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
  return FS_OK;
}

CgFlags::~CgFlags()
{
  if (h) (void)hipHostFree(h);
  for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
}

int CgFlags::after_iteration(int iter, const double *st, hipStream_t s, bool *stop)
{
  FS_HIP(hipMemcpyAsync(h + 2 * (iter & 1), st + kCgStateDone, sizeof(double) * 2, hipMemcpyDeviceToHost, s));
  FS_HIP(hipEventRecord(ev[iter & 1], s));
  *stop = false;
  if (iter >= 1) {
    FS_HIP(hipEventSynchronize(ev[(iter - 1) & 1]));
    *stop = h[2 * ((iter - 1) & 1)] != 0.0;
  }
  return FS_OK;
}

}  // namespace fs

using namespace fs;

extern "C" {

// y += a x on device vectors (the "+ lambda x" of bsbm_AtA, cg.h:17-21)
int fs_axpy(int n, double a, const double *x, double *y, fs_stream_t stream)
{
  if (n < 0 || !x || !y) { set_error("fs_axpy: bad argument"); return FS_ERR_ARG; }
  hipLaunchKernelGGL(axpy_kernel, dim3(kRedBlocks), dim3(kRedThreads), 0, (hipStream_t)stream, n, a, x, y);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// (A'A + lambda I) x = b with A given by its handle and the handle of its transpose (the reference passes both
// matrices, cg.h:26-27); x, b device vectors of F = ncol(A) doubles.  Stops like cg.h:69 (||r|| <= tol ||b||) or
// after F iterations; *out_iter as the reference reports it.
int fs_cg(fs_matrix_t A, fs_matrix_t At, double *x, const double *b, double lambda, double tol, int *out_iter,
          fs_stream_t stream)
{
  FS_RANGE("fs_cg");
  if (!A || !At || !x || !b) { set_error("fs_cg: NULL argument"); return FS_ERR_ARG; }
  const int N = A->a.nrow, F = A->a.ncol;
  if (At->a.nrow != F || At->a.ncol != N) { set_error("fs_cg: At is not the transpose shape of A"); return FS_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  FixedOrderScope fixed(options().cg_fixed_order != 0);   // the products of a solve add in a fixed order: bit-identical run to run
  Workspace ws;
  double *r = ws.get(F), *p = ws.get(F), *q = ws.get(F), *tmp = ws.get(N), *part = ws.get(kRedBlocks * 3), *red = ws.get(4);
  if (!r || !p || !q || !tmp || !part || !red) { set_error("fs_cg: out of device memory"); return FS_ERR_HIP; }
  double *st = ws.get(kStDoubles);
  CgFlags fl;
  if (!st) { set_error("fs_cg: out of device memory"); return FS_ERR_HIP; }
  if (int rc = fl.init()) return rc;
  if (int rc = cg_dev_init(F, b, x, r, p, part, red, st, tol, s)) return rc;
  for (int iter = 0; iter < F; iter++) {
    if (int rc = fs_spmv(A, tmp, p, stream)) return rc;
    if (int rc = fs_spmv(At, q, tmp, stream)) return rc;
    if (int rc = cg_dev_steps(F, lambda, x, r, p, q, part, red, st, s)) return rc;
    bool stop = false;
    if (int rc = fl.after_iteration(iter, st, s, &stop)) return rc;
    if (stop) break;
  }
  double fin[2] = {0.0, 0.0};
  FS_HIP(hipMemcpyAsync(fin, st + kStDone, sizeof(fin), hipMemcpyDeviceToHost, s));
  FS_HIP(hipStreamSynchronize(s));
  if (out_iter) *out_iter = (int)fin[1];
  return FS_OK;
}

// two right-hand sides, X and B row-major F x 2 (cg.h:85-187)
int fs_cg2(fs_matrix_t A, fs_matrix_t At, double *X, const double *B, double lambda, double tol, int *out_iter,
           fs_stream_t stream)
{
  FS_RANGE("fs_cg2");
  if (!A || !At || !X || !B) { set_error("fs_cg2: NULL argument"); return FS_ERR_ARG; }
  const int N = A->a.nrow, F = A->a.ncol;
  if (At->a.nrow != F || At->a.ncol != N) { set_error("fs_cg2: At is not the transpose shape of A"); return FS_ERR_ARG; }
  FixedOrderScope fixed(options().cg_fixed_order != 0);   // the products of a solve add in a fixed order: bit-identical run to run
  // the two-column copies of both matrices, before the first iteration (fs_spmm itself never builds)
  if (int rc = fs_matrix_prepare(A, 2, 0, stream)) return rc;
  if (int rc = fs_matrix_prepare(At, 2, 0, stream)) return rc;
  hipStream_t s = (hipStream_t)stream;
  Workspace ws;
  double *R = ws.get(2 * (size_t)F), *P = ws.get(2 * (size_t)F), *Q = ws.get(2 * (size_t)F), *tmp = ws.get(2 * (size_t)N);
  double *part = ws.get(kRedBlocks * 3), *red = ws.get(4);
  if (!R || !P || !Q || !tmp || !part || !red) { set_error("fs_cg2: out of device memory"); return FS_ERR_HIP; }
  // the scalars live on the device from the first iteration on (see the head of this file)
  double *st = ws.get(kStDoubles);
  CgFlags fl;
  if (!st) { set_error("fs_cg2: out of device memory"); return FS_ERR_HIP; }
  if (int rc = fl.init()) return rc;
  double norms[2];
  if (int rc = cg2_dev_init(F, B, X, R, P, part, red, st, tol, norms, s)) return rc;
  for (int iter = 0; iter < F; iter++) {
    if (int rc = fs_spmm(A, tmp, P, 2, stream)) return rc;
    if (int rc = fs_spmm(At, Q, tmp, 2, stream)) return rc;
    if (int rc = cg2_dev_steps(F, lambda, X, R, P, Q, part, red, st, s)) return rc;
    bool stop = false;
    if (int rc = fl.after_iteration(iter, st, s, &stop)) return rc;
    if (stop) break;
  }
  if (int rc = cg2_dev_finish(F, norms, X, s)) return rc;
  double fin[2] = {0.0, 0.0};
  FS_HIP(hipMemcpyAsync(fin, st + kStDone, sizeof(fin), hipMemcpyDeviceToHost, s));
  FS_HIP(hipStreamSynchronize(s));
  if (out_iter) *out_iter = (int)fin[1];
  return FS_OK;
}

}  // extern "C"
