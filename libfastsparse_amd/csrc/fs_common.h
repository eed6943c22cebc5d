// fs_common.h -- internal declarations shared by the HIP translation units of
// libfastsparse_hip.so.  Not installed; the public surface is include/*.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>

#include "fastsparse_hip.h"

namespace fs {

// ---- kernel geometry (see DESIGN.md "Kernels") ---------------------------------------
constexpr int kBlock = 256;            // threads per workgroup = 4 wave64
constexpr int kChunk = 2048;           // non-zeros one workgroup streams through LDS
constexpr int kPerThread = kChunk / kBlock;

// One CSR in HBM plus the chunk schedule of the streaming SpMV kernel.
struct DeviceCsr {
  int nrow = 0, ncol = 0;
  int64_t nnz = 0;
  int *row_ptr = nullptr;      // nrow + 1
  int *cols = nullptr;         // nnz
  double *vals = nullptr;      // nnz, or nullptr for a pattern-only matrix
  bool owns = true;            // false: arrays borrowed from the caller
  // schedule: chunk c streams non-zeros [c*kChunk, (c+1)*kChunk) and finishes the rows
  // whose first non-zero lies in that range: rows [first_row[c], first_row[c+1]).
  int nchunks = 0;
  int *first_row = nullptr;    // nchunks + 1
  double *head = nullptr;      // nchunks: sum of the chunk's leading non-zeros that belong to an earlier row
  double *tail = nullptr;      // nchunks: sum of the chunk's trailing non-zeros of a row that ends later
  int spanning = 0;            // number of rows that cross a chunk boundary (0 => no fix-up launch)
};

}  // namespace fs

struct fs_matrix_s {
  fs::DeviceCsr a;             // A
  fs::DeviceCsr at;            // A' (built on demand)
  bool has_t = false;
  int device = 0;
  std::mutex lock;             // serialises products that share head/tail scratch
};

struct fs_cbcsr_s {
  int nrow = 0, ncol = 0, nblocks = 0, colblocksize = 0;
  int64_t nnz = 0;
  int *row_ptr = nullptr;      // nblocks*nrow + 1, cell = block*nrow + row
  int *cols = nullptr;
  int device = 0;
};

namespace fs {

// ---- error plumbing -------------------------------------------------------------------
void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define FS_HIP(call)                                                   \
  do {                                                                 \
    hipError_t e_ = (call);                                            \
    if (e_ != hipSuccess) return ::fs::hip_fail(e_, #call, __FILE__, __LINE__); \
  } while (0)

struct Options {
  int strict_order = 0;
  int spmv_kernel = 0;
};
Options &options();

// ---- launchers implemented in fs_kernels.hip --------------------------------------------
int launch_spmv(const DeviceCsr &A, double *y, const double *x, hipStream_t s);
int launch_spmm(const DeviceCsr &A, double *Y, const double *X, int k, hipStream_t s);
int launch_cbcsr(const fs_cbcsr_s &A, double *y, const double *x, hipStream_t s);

// ---- format work implemented in fs_format.hip --------------------------------------------
int build_schedule(DeviceCsr &A, hipStream_t s);
int coo_to_csr_device(DeviceCsr &out, int nrow, int ncol, int64_t nnz, const int *rows_dev,
                      const int *cols_dev, const double *vals_dev, hipStream_t s);
int transpose_device(const DeviceCsr &A, DeviceCsr &At, hipStream_t s);
void free_csr(DeviceCsr &A);

}  // namespace fs
