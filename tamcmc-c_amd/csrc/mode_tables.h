// mode_tables.h -- host-side scalar helpers and table builders (see mode_tables.cpp).
#pragma once
#include <cstdint>

#include "../../include/tamcmc_hip.h"

namespace tamcmc {

long double Pslm(int s, int l, int m);                 // acoefs.cpp:51-110
double Qlm(int l, int m);                              // build_lorentzian.cpp:583-592
void amplitude_ratio(int l, double beta_deg, double *V);  // function_rot.cpp:15-41
double lin_interpol(const double *x, const double *y, long n, double xi);  // interpol.cpp:13-43
void linfit(const double *x, const double *y, long n, double out[2]);      // linfit.cpp:17-35
double eta0_from_dnu(double dnu);                      // models.cpp:6073-6084
double eta0_fct(const double *fl0, long n);            // models.cpp:6065-6071
int set_imin_imax(double x_first, double x_last, int64_t Nx, int l, double fc, double gamma, double f_s, double c,
                  double step, int *i0, int *i1);      // build_lorentzian.cpp:595-676

int build_mode_table(int model_id, const double *params, const int32_t *plength, const double *x, int64_t Nx,
                     tamcmc_multiplet *mults, int max_mults, int *n_mults, double *noise_abs, int *nharvey,
                     int *nnoise);
// number of multiplets a parameter vector of this layout produces (-1: unknown model)
int count_multiplets(int model_id, const int32_t *plength);

}  // namespace tamcmc
