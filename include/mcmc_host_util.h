/*
 * mcmc_host_util.h -- host utilities the reference's drivers call around the
 * samplers: allocation (src/util/alloc_util.h), command line + result printing
 * (src/util/io.h) and CSV data in / out (src/util/data_util.h).  Same names
 * and argument lists, so src/(all)/(all)_host.c and tests/(all).c of the reference
 * link against libmcmc_amd.so without its util/ objects.  Off the hot path
 * (SURVEY.md section 8f, rows f1 and f2).
 */
#ifndef MCMC_AMD_HOST_UTIL_H
#define MCMC_AMD_HOST_UTIL_H

#include <stdbool.h>

#include "mcmc_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- src/util/alloc_util.h:8-33 ---- */
void malloc_data_vectors(data_str *data, mcmc_str mcin);
void malloc_data_vectors_sp(data_str *data, mcmc_str mcin);
void malloc_data_vectors_mp(data_str *data, mcmc_str mcin);
void malloc_sample_vectors(mcmc_v_str *mcdata, mcmc_str mcin);
void malloc_sample_vectors_sp(mcmc_v_str *mcdata, mcmc_str mcin);
void malloc_autocorrelation_vectors(sec_v_str *secv, sec_str sec);
void malloc_autocorrelation_vectors_sp(sec_v_str *secv, sec_str sec);
void malloc_mcmc_vectors(mcmc_int_v *mclocv, mcmc_str mcin);
void malloc_mcmc_vectors_sp(mcmc_int_v *mclocv, mcmc_str mcin);
void free_data_vectors(data_str data, mcmc_str mcin);
void free_data_vectors_sp(data_str data, mcmc_str mcin);
void free_sample_vectors(mcmc_v_str mcdata);
void free_sample_vectors_sp(mcmc_v_str mcdata);
void free_autocorrelation_vectors(sec_v_str secv);
void free_autocorrelation_vectors_sp(sec_v_str secv);
void free_mcmc_vectors(mcmc_int_v mclocv, mcmc_str mcin);
void free_mcmc_vectors_sp(mcmc_int_v mclocv, mcmc_str mcin);

/* ---- src/util/io.h ---- */
void read_inputs(int an, char *av[], mcmc_str *mcin, sec_str *sec);
void read_inputs_gpu(int an, char *av[], mcmc_str *mcin, sec_str *sec, gpu_v_str *gpu);
void write_bandwidth_test_out(out_str res);
void write_test_out(out_str res);
void print_normalised_sample_means(mcmc_v_str mcdata, mcmc_str mcin);
void print_normalised_sample_means_sp(mcmc_v_str mcdata, mcmc_str mcin);
void write_perf_out_csv(char *rootdir, out_str res, bool first);
void write_csv_outputs(char *rootdir, mcmc_v_str mcdata, mcmc_str mcin, sec_str sec, sec_v_str secv);
void write_csv_outputs_sp(char *rootdir, mcmc_v_str mcdata, mcmc_str mcin, sec_str sec, sec_v_str secv);

/* ---- src/util/data_util.h ---- */
void read_data(char *dir, int store, data_str data, mcmc_str mcin);
void read_data_sp(char *dir, int store, data_str data, mcmc_str mcin);
void write_data(char *filename, double *data, int sz, int dim);
void write_data_sp(char *filename, float *data, int sz, int dim);
void write_performance_data(char *filename, out_str res, bool first);
void write_autocorr(char *filename, double *autocorrelation, sec_str sec);
void write_autocorr_sp(char *filename, float *autocorrelation, sec_str sec);

/* Not in the reference: -in <csv> and -out <prefix> given on the command line
 * (NULL when absent).  The reference hard-codes data/host/synthetic.csv and
 * out/host/synthetic/gpu_ (src/mcmc_gpu/gpu_host.c:36-42). */
const char *mcmc_amd_cli_input(void);
const char *mcmc_amd_cli_output(void);

#ifdef __cplusplus
}
#endif
#endif
