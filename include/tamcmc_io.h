/*
 * tamcmc_io.h -- C ABI of the input front end kept from the reference (first step):
 * the ASCII `.data` reader and the `.model` reader + parameter-vector builder of the LOCAL fit (BASELINE config C1).
 * It produces what Config::read_inputs_priors_local leaves in `modeling.inputs` (tamcmc/sources/config.cpp:709-723):
 * Input_Data{inputs, relax, priors(4 x N), priors_names -> switch ids, plength[11], extra_priors}
 * (tamcmc/headers/data.h:51-62), i.e. exactly the arrays tamcmc_sampler_create / tamcmc_hip_* take.
 *
 * Replaces, for model_MS_local_basic:
 *   Config::read_data_ascii_Ncols          tamcmc/sources/config.cpp:907-1060   (.data)
 *   data range selection of Config::setup  tamcmc/sources/config.cpp:312-347
 *   read_MCMC_file_local                   tamcmc/sources/io_local.cpp:25-327   (.model)
 *   build_init_local + set_noise_params_local   io_local.cpp:329-1238, IO_models io_models.cpp:40-297
 *   Config::convert_priors_names_to_switch config.cpp:725-752 (ids of Config/default/primepriors_ctrl.list)
 * Not covered yet (TAMCMC_IO_ERR_UNSUPPORTED): model_MS_local_Hnlm, the other global variants (a1n/a1l/a2a3/ajAlm/AppWidth/
 * Classic_v2,v3), the asymptotic (RGB) and ajfit dialects, the .cfg files.
 * Parity: the reference cannot be run here and ships no expected Input_Data dump: "parity unpinned"; tests pin the
 * result against values derived by hand from the shipped file with the rules cited above.
 */
#ifndef TAMCMC_IO_H
#define TAMCMC_IO_H

#include <stdint.h>

struct tamcmc_sampler_config; /* include/tamcmc_sampler.h */

#ifdef __cplusplus
extern "C" {
#endif

#define TAMCMC_IO_OK 0
#define TAMCMC_IO_ERR_OPEN (-21)         /* file cannot be opened (the reference exits) */
#define TAMCMC_IO_ERR_SYNTAX (-22)       /* malformed file (the reference exits or reads out of bounds) */
#define TAMCMC_IO_ERR_UNSUPPORTED (-23)  /* a model / keyword of another dialect */
#define TAMCMC_IO_ERR_EMPTY_RANGE (-24)  /* no mode inside the slice's frequency range (io_local.cpp:559-564) */
#define TAMCMC_IO_ERR_ARG (-25)

const char *tamcmc_io_last_error(void);

/* `.data`: '#' header lines, optional '!' label line, optional '*' unit line, then whitespace-separated numeric columns.
 * Returns a malloc'ed row-major [nrows x ncols] table (free with tamcmc_io_free); unparsable fields become NaN. */
int tamcmc_io_read_data(const char *path, double **table, int64_t *nrows, int64_t *ncols);
void tamcmc_io_free(void *p);
/* rows [imin, imax) of column x_col inside [xmin, xmax) as Config::setup selects them (first x >= xmin, then while x < xmax) */
int tamcmc_io_select_range(const double *table, int64_t nrows, int64_t ncols, int x_col, double xmin, double xmax,
                           int64_t *imin, int64_t *imax);

typedef struct tamcmc_inputs tamcmc_inputs;

/* `.model` of a local fit, slice `slice_ind` (0-based '*' range line); resol = x[2]-x[1] of the WHOLE data file
 * (config.cpp:720), used as the lower bound of the automatic width prior. */
int tamcmc_io_load_model_local(const char *model_path, int slice_ind, double resol, tamcmc_inputs **out);
/* `.model` of a global main-sequence fit (one '*' range): model_MS_Global_aj_HarveyLike (id 23) and
 * model_MS_Global_a1etaa3_HarveyLike_Classic (id 3).  Replaces read_MCMC_file_MS_Global + build_init_MS_Global +
 * set_noise_params (tamcmc/sources/io_ms_global.cpp:27-360, :362-1445, :1447-1536, :1718-1850); prior_class 2. */
int tamcmc_io_load_model_global(const char *model_path, double resol, tamcmc_inputs **out);
/* `.model` of a red-giant fit: model_RGB_asympt_aj_AppWidth_HarveyLike_v4 (id 25) and model_RGB_asympt_aj_CteWidth_HarveyLike_v4
 * (id 27).  Same file layout as the global fits, with the nodes of the frequency-bias spline in the "hyper priors" section and the
 * mixed-mode keywords (delta01, DP1, alpha_g, q, Wfactor, Hfactor, rot_env, rot_core, ...) among the common parameters.  Replaces
 * build_init_asymptotic + settings_aj_splittings_RGB (tamcmc/sources/io_asymptotic.cpp:32-955) and set_width_App2016_params_v2
 * (io_ms_global.cpp:1625-1720); prior_class 4. */
int tamcmc_io_load_model_asymptotic(const char *model_path, double resol, tamcmc_inputs **out);
void tamcmc_inputs_free(tamcmc_inputs *in);

int tamcmc_inputs_nparams(const tamcmc_inputs *in);
/* any pointer may be NULL.  priors: 4 x Nparams row-major; extra_priors: 10 slots (unused ones 0);
 * model_id per Config/default/models_ctrl.list, prior_class per priors_ctrl.list (3 = io_local). */
int tamcmc_inputs_get(const tamcmc_inputs *in, double *params, int32_t *relax, double *priors, int32_t *priors_switch,
                      int32_t *plength /*11*/, double *extra_priors /*10*/, double *freq_range /*2*/, int32_t *model_id,
                      int32_t *prior_class, double *dnu, double *c_l);
const char *tamcmc_inputs_name(const tamcmc_inputs *in, int i);        /* Input_Data.inputs_names[i] */
const char *tamcmc_inputs_prior_name(const tamcmc_inputs *in, int i);  /* Input_Data.priors_names[i] */
const char *tamcmc_inputs_model_name(const tamcmc_inputs *in);         /* Input_Data.model_fullname */

/* ---------------- `.cfg` files ----------------
 * Config/default/config_default.cfg dialect: `!Group:` lines, `key=value; comment` entries, '#' comment lines, `/END;`
 * (Config::format_line + Config::read_cfg_file, tamcmc/sources/config.cpp:1062-1112, :1223-1732). */
typedef struct tamcmc_cfg tamcmc_cfg;
const char *tamcmc_cfg_last_error(void);
int tamcmc_cfg_open(const char *path, tamcmc_cfg **out);
void tamcmc_cfg_free(tamcmc_cfg *cfg);
int tamcmc_cfg_string(const tamcmc_cfg *cfg, const char *group, const char *key, char *buf, int n);
int tamcmc_cfg_numbers(const tamcmc_cfg *cfg, const char *group, const char *key, double *out, int max, int *n);
/* !MALA, !Modeling (likelihood, prior class) and !Outputs (Nsamples, Nbuffer) -> the scalar fields of tamcmc_sampler_config
 * (include/tamcmc_sampler.h);
 * Nt_learn / periods_learn are written to the caller's buffers (max_learn entries) and linked into the struct. */
int tamcmc_cfg_sampler(const tamcmc_cfg *cfg, struct tamcmc_sampler_config *out, int64_t *Nt_learn, int64_t *periods_learn,
                       int max_learn, int64_t *Nsamples, int64_t *Nbuffer);
/* Config/default/errors_default.cfg (Config::read_defautlerrors, config.cpp:2096-2150): initial proposal standard
 * deviations err = A*value + B per FREE parameter, matched by name, 1 without a match (MALA::init_proposal, MALA.cpp:246-262). */
int tamcmc_io_init_errors(const char *errors_path, const char *const *names, const double *values, int64_t nvars, double *errors);

#ifdef __cplusplus
}
#endif
#endif
