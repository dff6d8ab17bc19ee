/*
 * CPU restatement of the torch_sputnik hot path in plain C -- TEST
 * INFRASTRUCTURE ONLY (checker for tests/, smoke() and bench.py's
 * cpu_baseline leg; never linked into or called from the product).
 *
 * What is restated is the contract at the reference's call sites:
 *   spmm        src/spmm_cuda.cu:48-57        (sputnik::CudaSpmm)
 *   sddmm       src/sddmm_cuda.cu:45-54       (sputnik::CudaSddmm)
 *   softmax     src/softmax_cuda.cu:35-43     (sputnik::SparseSoftmax)
 *   transpose   src/transpose_cuda.cu:90-99   (cusparseCsr2cscEx2, ALG1)
 * The arithmetic itself lives in google-research/sputnik (un-vendored
 * submodule, pin unknown) and cuSPARSE; parity with their bit patterns is
 * unpinned -- see oracle/__init__.py.
 *
 * All "_f64acc" entry points accumulate in double and round once to float,
 * so a float32 device kernel is compared against a result that is within
 * half an ulp of the exact one.  oracle_spmm_f32 accumulates in float and is
 * what bench.py times as the sparse CPU baseline.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int t) {
#ifdef _OPENMP
  omp_set_num_threads(t);
#else
  (void)t;
#endif
}

/* C[m,n] = A_csr[m,k] * B[k,n]; row_indices is only a processing order. */
void oracle_spmm_f64acc(int m, int k, int n, const int* row_offsets,
                        const int* column_indices, const float* values,
                        const float* dense, float* out) {
  (void)k;
#pragma omp parallel
  {
    double* acc = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 8)
    for (int i = 0; i < m; ++i) {
      for (int c = 0; c < n; ++c) acc[c] = 0.0;
      for (int p = row_offsets[i]; p < row_offsets[i + 1]; ++p) {
        const double a = (double)values[p];
        const float* b = dense + (size_t)column_indices[p] * (size_t)n;
        for (int c = 0; c < n; ++c) acc[c] += a * (double)b[c];
      }
      float* o = out + (size_t)i * (size_t)n;
      for (int c = 0; c < n; ++c) o[c] = (float)acc[c];
    }
    free(acc);
  }
}

/* Same contract, float accumulation: the CPU baseline that bench.py times. */
void oracle_spmm_f32(int m, int k, int n, const int* row_offsets,
                     const int* column_indices, const float* values,
                     const float* dense, float* out) {
  (void)k;
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < m; ++i) {
    float* o = out + (size_t)i * (size_t)n;
    for (int c = 0; c < n; ++c) o[c] = 0.0f;
    for (int p = row_offsets[i]; p < row_offsets[i + 1]; ++p) {
      const float a = values[p];
      const float* b = dense + (size_t)column_indices[p] * (size_t)n;
#pragma omp simd
      for (int c = 0; c < n; ++c) o[c] += a * b[c];
    }
  }
}

/* out[p] = <lhs[i_p,:], rhs[j_p,:]>, lhs [m,k], rhs [n,k] row-major. */
void oracle_sddmm_f64acc(int m, int k, int n, const int* row_offsets,
                         const int* column_indices, const float* lhs,
                         const float* rhs, float* out) {
  (void)n;
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < m; ++i) {
    const float* l = lhs + (size_t)i * (size_t)k;
    for (int p = row_offsets[i]; p < row_offsets[i + 1]; ++p) {
      const float* r = rhs + (size_t)column_indices[p] * (size_t)k;
      double acc = 0.0;
      for (int t = 0; t < k; ++t) acc += (double)l[t] * (double)r[t];
      out[p] = (float)acc;
    }
  }
}

/* Row-wise softmax over stored entries only; empty rows write nothing. */
void oracle_softmax_f64(int m, const int* row_offsets, const float* values,
                        float* out) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int i = 0; i < m; ++i) {
    const int p0 = row_offsets[i], p1 = row_offsets[i + 1];
    if (p0 == p1) continue;
    double mx = -INFINITY;
    for (int p = p0; p < p1; ++p)
      if ((double)values[p] > mx) mx = (double)values[p];
    double s = 0.0;
    for (int p = p0; p < p1; ++p) s += exp((double)values[p] - mx);
    for (int p = p0; p < p1; ++p)
      out[p] = (float)(exp((double)values[p] - mx) / s);
  }
}

/*
 * CSR(m x n) -> CSR(n x m) by a stable counting sort on the column index:
 * within each output row the source row ids ascend (cuSPARSE CSR2CSC_ALG1
 * ordering).  `replicas` value arrays of length nnz share the permutation.
 */
void oracle_csr_transpose(int m, int n, int nnz, int replicas,
                          const float* values, const int* row_offsets,
                          const int* column_indices, float* values_t,
                          int* row_offsets_t, int* column_indices_t) {
  int* cursor = (int*)calloc((size_t)n + 1, sizeof(int));
  for (int p = 0; p < nnz; ++p) cursor[column_indices[p] + 1]++;
  for (int c = 0; c < n; ++c) cursor[c + 1] += cursor[c];
  memcpy(row_offsets_t, cursor, sizeof(int) * ((size_t)n + 1));
  for (int i = 0; i < m; ++i) {
    for (int p = row_offsets[i]; p < row_offsets[i + 1]; ++p) {
      const int q = cursor[column_indices[p]]++;
      column_indices_t[q] = i;
      for (int r = 0; r < replicas; ++r)
        values_t[(size_t)r * nnz + q] = values[(size_t)r * nnz + p];
    }
  }
  free(cursor);
}
