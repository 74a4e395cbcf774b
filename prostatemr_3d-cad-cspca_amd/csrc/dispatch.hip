// dispatch.hip -- public conv entry points: choose the MFMA implicit-GEMM path (conv_mfma.hip) when the
// layer shape allows it, else the generic direct path (conv_direct.hip).  Also status strings.
#include "common.h"

extern "C" {
int m1_conv3d_fwd_direct(const m1_conv_desc_t*, const float*, const float*, void*, void*);
int m1_conv3d_dgrad_direct(const m1_conv_desc_t*, const float*, const void*, void* const*, void*);
int m1_conv3d_wgrad_direct(const m1_conv_desc_t*, const void*, float*, float*, float*, void*);
int m1_convT3d_fwd_direct(const m1_conv_desc_t*, const float*, const float*, void*, void*);
int m1_convT3d_dgrad_direct(const m1_conv_desc_t*, const float*, const void*, void* const*, void*);
int m1_convT3d_wgrad_direct(const m1_conv_desc_t*, const void*, float*, float*, float*, void*);
}

static inline bool desc_ok(const m1_conv_desc_t* d) {
    return d && d->N > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->kd > 0 && d->kh > 0 &&
           d->kw > 0 && d->sd > 0 && d->sh > 0 && d->sw > 0 && d->nsrc >= 1 && d->nsrc <= M1_MAX_SRC;
}
static inline double esz(int dt) { return dt == M1_BF16 ? 2.0 : 4.0; }
static inline void conv_out_dims(const m1_conv_desc_t* d, int* od, int* oh, int* ow) {
    *od = (d->D + d->sd - 1) / d->sd; *oh = (d->H + d->sh - 1) / d->sh; *ow = (d->W + d->sw - 1) / d->sw;
}
// algorithmic work: flops = 2*MAC; bytes = read each logical input once + write output once
static inline double conv_macs(const m1_conv_desc_t* d, int transposed) {
    int od, oh, ow; conv_out_dims(d, &od, &oh, &ow);
    const double taps = (double)d->kd * d->kh * d->kw;
    const double vox = transposed ? (double)d->D * d->H * d->W : (double)od * oh * ow;   // coarse-grid voxels
    return (double)d->N * vox * taps * d->Cin * d->Cout;
}
static inline double conv_in_elems(const m1_conv_desc_t* d) { return (double)d->N * d->D * d->H * d->W * d->Cin; }
static inline double conv_out_elems(const m1_conv_desc_t* d, int transposed) {
    if (transposed) return (double)d->N * d->D * d->sd * d->H * d->sh * d->W * d->sw * d->Cout;
    int od, oh, ow; conv_out_dims(d, &od, &oh, &ow);
    return (double)d->N * od * oh * ow * d->Cout;
}

extern "C" const char* m1_status_name(int s) {
    switch (s) {
        case M1_OK: return "M1_OK";
        case M1_ERR_BAD_ARG: return "M1_ERR_BAD_ARG";
        case M1_ERR_UNSUPPORTED: return "M1_ERR_UNSUPPORTED";
        case M1_ERR_LAUNCH: return "M1_ERR_LAUNCH";
        case M1_ERR_WORKSPACE: return "M1_ERR_WORKSPACE";
        default: return "M1_ERR_UNKNOWN";
    }
}
extern "C" int m1_abi_version(void) { return 1; }

extern "C" int m1_conv3d_fwd(const m1_conv_desc_t* d, const float* w, const float* bias, void* y, void* stream) {
    if (!desc_ok(d)) return M1_ERR_BAD_ARG;
    M1ProfScope ps("conv3d_fwd", 2.0 * conv_macs(d, 0), (conv_in_elems(d) + conv_out_elems(d, 0)) * esz(d->dtype), (hipStream_t)stream);
    return m1_conv3d_fwd_direct(d, w, bias, y, stream);
}
extern "C" int m1_conv3d_dgrad(const m1_conv_desc_t* d, const float* w, const void* dy, void* const* dx, void* stream) {
    if (!desc_ok(d)) return M1_ERR_BAD_ARG;
    M1ProfScope ps("conv3d_dgrad", 2.0 * conv_macs(d, 0), (conv_in_elems(d) + conv_out_elems(d, 0)) * esz(d->dtype), (hipStream_t)stream);
    return m1_conv3d_dgrad_direct(d, w, dy, dx, stream);
}
extern "C" int m1_conv3d_wgrad(const m1_conv_desc_t* d, const void* dy, float* dw, float* db, float* ws, void* stream) {
    if (!desc_ok(d)) return M1_ERR_BAD_ARG;
    M1ProfScope ps("conv3d_wgrad", 2.0 * conv_macs(d, 0), (conv_in_elems(d) + conv_out_elems(d, 0)) * esz(d->dtype), (hipStream_t)stream);
    return m1_conv3d_wgrad_direct(d, dy, dw, db, ws, stream);
}
extern "C" int m1_convT3d_fwd(const m1_conv_desc_t* d, const float* w, const float* bias, void* y, void* stream) {
    if (!desc_ok(d)) return M1_ERR_BAD_ARG;
    M1ProfScope ps("convT3d_fwd", 2.0 * conv_macs(d, 1), (conv_in_elems(d) + conv_out_elems(d, 1)) * esz(d->dtype), (hipStream_t)stream);
    return m1_convT3d_fwd_direct(d, w, bias, y, stream);
}
extern "C" int m1_convT3d_dgrad(const m1_conv_desc_t* d, const float* w, const void* dy, void* const* dx, void* stream) {
    if (!desc_ok(d)) return M1_ERR_BAD_ARG;
    M1ProfScope ps("convT3d_dgrad", 2.0 * conv_macs(d, 1), (conv_in_elems(d) + conv_out_elems(d, 1)) * esz(d->dtype), (hipStream_t)stream);
    return m1_convT3d_dgrad_direct(d, w, dy, dx, stream);
}
extern "C" int m1_convT3d_wgrad(const m1_conv_desc_t* d, const void* dy, float* dw, float* db, float* ws, void* stream) {
    if (!desc_ok(d)) return M1_ERR_BAD_ARG;
    M1ProfScope ps("convT3d_wgrad", 2.0 * conv_macs(d, 1), (conv_in_elems(d) + conv_out_elems(d, 1)) * esz(d->dtype), (hipStream_t)stream);
    return m1_convT3d_wgrad_direct(d, dy, dw, db, ws, stream);
}
