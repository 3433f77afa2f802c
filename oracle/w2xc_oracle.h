/*
 * w2xc_oracle.h -- C interface of the CPU oracle (TEST INFRASTRUCTURE ONLY; see
 * w2xc_oracle.c for the parity-pinning statement).
 */
#ifndef W2XC_ORACLE_H_
#define W2XC_ORACLE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One conv layer as the reference's Model holds it (modelHandler.hpp:27-31):
 * weight[o][i][kh][kw] as float (the JSON double cast to float,
 * modelHandler.cpp:95-97), index o*nin+i; bias[o] as double (:109-112). */
typedef struct {
    int nin, nout;
    const float *weight;   /* nout*nin*9 */
    const double *bias;    /* nout */
} w2xc_oracle_layer;

/* Model::filter (modelHandler.cpp:26-72): planar in [nin][h][w] -> planar out
 * [nout][h][w], same size, replicate border.  Returns -1 on plane-count mismatch. */
int w2xc_oracle_filter(const w2xc_oracle_layer *L, int n_in_planes, const float *in, int w, int h,
                       float *out, int njob);

/* convertWithModels (convertRoutine.cpp:21-169).  Strides in floats. */
int w2xc_oracle_convert(const w2xc_oracle_layer *layers, int nlayers, const float *in,
                        size_t in_stride, int w, int h, float *out, size_t out_stride,
                        int block_splitting, int block_w, int block_h, int njob);

/* fp64-accumulate truth of the same network (error budgeting). out is w*h doubles. */
int w2xc_oracle_convert_f64(const w2xc_oracle_layer *layers, int nlayers, const float *in,
                            size_t in_stride, int w, int h, double *out, int njob);

#ifdef __cplusplus
}
#endif
#endif
