// gpboost_amd/csrc/laplace_kernels.h -- launch interface of laplace_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

namespace gpb {

struct LapMat {            // B = I - A of the Vecchia factor and its transposed index
  const double* A; const double* D; const int* nn; const int* t_ptr; const int* t_pos; int n, m;
};
struct LapTri {            // one level-scheduled triangular solve; rows stored in LEVEL ORDER (position q)
  const int* ptr;          // [nlev + 1] level boundaries (positions)
  const int* rows;         // [n] row index of position q
  const int* hsrc;         // [n * 32] head: source row of the entry (-1 = none)
  const int* optr;         // [n + 1] overflow CSR (entries 33.. of a row)
  const int* osrc;         // [max(novf, 1)]
  const double* hval;      // [n * 32] matrix entries in head layout (refreshed per evaluation by lap_permute_factor)
  const double* oval;      // [max(novf, 1)]
  int nlev;
};
struct LapLevels { LapTri fwd, bwd; };   // fwd: (D^-1 + W) B z = t;  bwd: B^T t = r
struct CgScalars {         // per-column CG scalars on the device
  double* a; double* a_old; double* b; double* rz_old; double* rnorm; double* Td; double* Ts;
};

hipError_t lap_newton_setup(const double* mode, const int* y, const double* D, int n, double* W, double* rhs, double* dw, hipStream_t st);
hipError_t lap_apply(const LapMat& B, const double* W, const double* h, double* v, double* tmp, int ncol, hipStream_t st);
hipError_t lap_B(const LapMat& B, const double* x, double* out, int ncol, hipStream_t st);
hipError_t lap_Bt(const LapMat& B, const double* x, double* out, int ncol, hipStream_t st);
hipError_t lap_objective(const double* x, const int* y, const double* Bx, const double* D, int n, double* out2, hipStream_t st);
hipError_t lap_vadu(const LapMat& B, const LapLevels& lv, const double* dw, const double* r, double* z, double* t, int ncol, hipStream_t st);
hipError_t lap_permute_factor(const double* A, const int* hpos, const int* opos, size_t nh, size_t novf, double* hval, double* oval, hipStream_t st);
hipError_t lap_cg_alpha(const double* r, const double* z, const double* h, const double* v, int n, int ncol, const CgScalars& sc, hipStream_t st);
hipError_t lap_cg_update(double* u, double* r, const double* h, const double* v, int n, int ncol, const CgScalars& sc, hipStream_t st);
hipError_t lap_cg_beta(const double* r, const double* z, double* h, int n, int ncol, const CgScalars& sc, int j, int p_max, hipStream_t st);
hipError_t lap_lincomb(double* out, const double* x, const double* y, double cx, double cy, int n, hipStream_t st);
hipError_t lap_scale_probes(const double* rv, const double* dw, int n, int ncol, double* out, hipStream_t st);
hipError_t lap_logsums(const double* D, const double* dw, int n, double* out2, hipStream_t st);
hipError_t lap_dot(const double* x, const double* y, int n, double* out2, hipStream_t st);

}  // namespace gpb
