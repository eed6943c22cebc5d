/*
 * fs_synth.c -- TEST INFRASTRUCTURE ONLY: CPU twin of the device-side synthetic generators in
 * libfastsparse_amd/csrc/fs_format.hip (fs_synth_uniform / _powerlaw_lengths / _fill), so that
 * bench.py's CPU baseline and the parity tests can regenerate, on the host, exactly the matrix a
 * GPU generated for itself.  Counter-based (splitmix64 of seed, row, slot): no state, any shard
 * of rows can be produced independently.  Workloads per SURVEY.md 8(d): C2/C3/C4 uniform columns,
 * fixed entries per row; C5 power-law row lengths.
 */
#include <stdint.h>

#define FSO_API __attribute__((visibility("default")))

static inline uint64_t splitmix64(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static inline void synth_entry(uint64_t seed, int64_t grow, int slot, int ncol, int *c, double *v)
{
  uint64_t h = splitmix64(seed ^ ((uint64_t)grow * 0x100000001B3ull + (uint64_t)slot));
  *c = (int)(((unsigned __int128)h * (uint64_t)ncol) >> 64);
  uint64_t h2 = splitmix64(h ^ 0xABCDEF0123456789ull);
  *v = (double)(h2 >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

FSO_API void fso_synth_uniform(int nrow, int ncol, int per_row, uint64_t seed, int64_t row_offset,
                               int *row_ptr, int *cols, double *vals)
{
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrow; r++) {
    if (row_ptr) row_ptr[r] = (int)(r * per_row);
    for (int s = 0; s < per_row; s++) {
      int c; double v;
      synth_entry(seed, row_offset + r, s, ncol, &c, &v);
      cols[r * per_row + s] = c;
      if (vals) vals[r * per_row + s] = v;
    }
  }
  if (row_ptr) row_ptr[nrow] = (int)((int64_t)nrow * per_row);
}

FSO_API void fso_synth_powerlaw_lengths(int nrow, double scale, int max_len, uint64_t seed,
                                        int64_t row_offset, int *len)
{
#pragma omp parallel for schedule(static)
  for (int r = 0; r < nrow; r++) {
    uint64_t h = splitmix64(seed ^ (0xC0FFEEull + (uint64_t)(row_offset + r) * 0x9E3779B97F4A7C15ull));
    double u = (double)((h >> 11) + 1) * (1.0 / 9007199254740992.0);
    double L = scale / u;
    if (L > (double)max_len) L = (double)max_len;
    int n = (int)L;
    len[r] = n < 1 ? 1 : n;
  }
}

FSO_API void fso_synth_fill(int nrow, int ncol, uint64_t seed, int64_t row_offset, const int *row_ptr,
                            int *cols, double *vals)
{
#pragma omp parallel for schedule(dynamic, 1024)
  for (int r = 0; r < nrow; r++) {
    for (int i = row_ptr[r]; i < row_ptr[r + 1]; i++) {
      int c; double v;
      synth_entry(seed, row_offset + r, i - row_ptr[r], ncol, &c, &v);
      cols[i] = c;
      if (vals) vals[i] = v;
    }
  }
}
