// capi.hip -- the C ABI of include/heaac_dsp.h: device context + entry points.
//
// Thin by design: argument checks, table upload, kernel launches.  No host
// fallback exists -- if no HIP device is usable every entry point fails with
// HEAAC_ERR_NODEVICE / HEAAC_ERR_HIP.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "heaac_dsp.h"
#include "tables.h"
#include "kernels.h"

struct HeaacDevice {
    int device;
    float *d_tab;
    uint16_t *d_rev;
    void *d_work;
    size_t work_bytes;
    size_t max_frames;
};

extern "C" const char *heaac_build_info(void)
{
    return "heaac-amd gfx950 (hipcc, -ffp-contract=off) " __DATE__ " " __TIME__;
}

extern "C" const char *heaac_strerror(int err)
{
    switch (err) {
    case HEAAC_OK: return "ok";
    case HEAAC_ERR_ARG: return "bad argument or unsupported configuration";
    case HEAAC_ERR_HIP: return "HIP runtime error";
    case HEAAC_ERR_NOMEM: return "out of memory";
    case HEAAC_ERR_NODEVICE: return "no usable gfx950 device";
    }
    return "unknown error";
}

extern "C" size_t heaac_device_workspace_bytes(size_t max_frames)
{
    (void)max_frames;
    return 0;
}

extern "C" int heaac_device_create(HeaacDevice **out, size_t max_frames)
{
    if (!out) return HEAAC_ERR_ARG;
    *out = NULL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return HEAAC_ERR_NODEVICE;
    HeaacDevice *d = (HeaacDevice *)calloc(1, sizeof(*d));
    if (!d) return HEAAC_ERR_NOMEM;
    if (hipGetDevice(&d->device) != hipSuccess) { free(d); return HEAAC_ERR_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d->device) != hipSuccess ||
        strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        free(d);
        return HEAAC_ERR_NODEVICE;
    }
    HeaacHostTables *t = (HeaacHostTables *)malloc(sizeof(*t));
    if (!t) { free(d); return HEAAC_ERR_NOMEM; }
    heaac_build_tables(t);
    d->max_frames = max_frames;
    d->work_bytes = heaac_device_workspace_bytes(max_frames);
    int rc = HEAAC_OK;
    if (hipMalloc((void **)&d->d_tab, sizeof(t->f)) != hipSuccess ||
        hipMalloc((void **)&d->d_rev, sizeof(t->rev)) != hipSuccess ||
        (d->work_bytes && hipMalloc(&d->d_work, d->work_bytes) != hipSuccess))
        rc = HEAAC_ERR_NOMEM;
    if (rc == HEAAC_OK &&
        (hipMemcpy(d->d_tab, t->f, sizeof(t->f), hipMemcpyHostToDevice) != hipSuccess ||
         hipMemcpy(d->d_rev, t->rev, sizeof(t->rev), hipMemcpyHostToDevice) != hipSuccess))
        rc = HEAAC_ERR_HIP;
    free(t);
    if (rc != HEAAC_OK) {
        heaac_device_destroy(d);
        return rc;
    }
    *out = d;
    return HEAAC_OK;
}

extern "C" void heaac_device_destroy(HeaacDevice *d)
{
    if (!d) return;
    if (d->d_tab) (void)hipFree(d->d_tab);
    if (d->d_rev) (void)hipFree(d->d_rev);
    if (d->d_work) (void)hipFree(d->d_work);
    free(d);
}

extern "C" int heaac_imdct_half_batch(HeaacDevice *dev, int which, float *d_out, const float *d_in,
                                      size_t n, void *stream)
{
    if (!dev || which < 0 || which > 3)
        return HEAAC_ERR_ARG;
    if (n == 0)
        return HEAAC_OK;
    if (!d_out || !d_in || d_out == d_in)
        return HEAAC_ERR_ARG;
    return heaac_launch_imdct_half(dev->d_tab, dev->d_rev, which, d_out, d_in, n, (hipStream_t)stream);
}

extern "C" int heaac_lc_decode_batch(HeaacDevice *dev, int channels,
                                     const float *d_coeffs, const HeaacIcs *d_ics,
                                     const float *d_state_in, float *d_state_out,
                                     void *d_pcm, int pcm_format, size_t n, void *stream)
{
    if (!dev || channels < 1 || channels > 2 ||
        (pcm_format != HEAAC_PCM_F32_PLANAR && pcm_format != HEAAC_PCM_S16_INTERLEAVED))
        return HEAAC_ERR_ARG;
    if (n == 0)
        return HEAAC_OK;
    if (!d_coeffs || !d_ics || !d_state_in || !d_state_out || !d_pcm)
        return HEAAC_ERR_ARG;
    return heaac_launch_lc(dev->d_tab, dev->d_rev, channels, d_coeffs, d_ics, d_state_in, d_state_out,
                           d_pcm, pcm_format, n, (hipStream_t)stream);
}
