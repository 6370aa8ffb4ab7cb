// check.hip -- heaac_he_check_batch / heaac_validate_frame: the rules of validate.h applied to a batch
// of records on the device (one thread per frame) or to one frame on the host.
#include <hip/hip_runtime.h>
#include "heaac_dsp.h"
#include "validate.h"

struct CheckResult {
    unsigned long long first;     // lowest failing frame index (~0 = none)
    unsigned count;               // failing frames
};

__global__ void k_check(const HeaacSbrFrame *g_sbr, const HeaacSbrHeader *g_hdr, unsigned long long n_hdr,
                        const HeaacPsFrame *g_ps, int ncore, unsigned long long n, CheckResult *res)
{
    for (unsigned long long f = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; f < n;
         f += (unsigned long long)gridDim.x * blockDim.x) {
        int r = heaac_check_sbr_frame(&g_sbr[f], g_hdr, (size_t)n_hdr, ncore);
        if (!r && g_ps) r = heaac_check_ps_frame(&g_ps[f]);
        if (r) {
            atomicMin(&res->first, f);
            atomicAdd(&res->count, 1u);
        }
    }
}

static int ncore_of(int cfg)
{
    return cfg == HEAAC_CFG_HEV1 ? 2 : (cfg == HEAAC_CFG_HEV1_MONO || cfg == HEAAC_CFG_HEV2) ? 1 : 0;
}

extern "C" int heaac_validate_frame(int cfg, const HeaacSbrFrame *sbr, const HeaacSbrHeader *hdr, size_t n_hdr,
                                    const HeaacPsFrame *ps)
{
    const int ncore = ncore_of(cfg);
    if (!ncore || !sbr || !hdr || (cfg == HEAAC_CFG_HEV2 && !ps))
        return -1;
    int r = heaac_check_sbr_frame(sbr, hdr, n_hdr, ncore);
    if (!r && cfg == HEAAC_CFG_HEV2) r = heaac_check_ps_frame(ps);
    return r;
}

extern "C" int heaac_he_check_batch(HeaacDevice *dev, int cfg, const HeaacSbrFrame *d_sbr,
                                    const HeaacSbrHeader *d_hdr, size_t n_hdr, const HeaacPsFrame *d_ps,
                                    size_t n, void *stream, size_t *first_bad, int *rule)
{
    const int ncore = ncore_of(cfg);
    if (first_bad) *first_bad = (size_t)-1;
    if (rule) *rule = HEAAC_BAD_NONE;
    if (!dev || !ncore)
        return HEAAC_ERR_ARG;
    if (n == 0)
        return HEAAC_OK;
    if (!d_sbr || !d_hdr || !n_hdr || (cfg == HEAAC_CFG_HEV2 && !d_ps))
        return HEAAC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    CheckResult *d_res = NULL, h_res = { ~0ull, 0 };
    if (hipMalloc((void **)&d_res, sizeof(*d_res)) != hipSuccess)
        return HEAAC_ERR_NOMEM;
    int rc = HEAAC_OK;
    if (hipMemcpyAsync(d_res, &h_res, sizeof(h_res), hipMemcpyHostToDevice, s) != hipSuccess)
        rc = HEAAC_ERR_HIP;
    if (rc == HEAAC_OK) {
        unsigned long long g = (n + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(k_check, dim3((unsigned)g), dim3(256), 0, s, d_sbr, d_hdr, (unsigned long long)n_hdr,
                           cfg == HEAAC_CFG_HEV2 ? d_ps : (const HeaacPsFrame *)NULL, ncore, (unsigned long long)n, d_res);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(&h_res, d_res, sizeof(h_res), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            rc = HEAAC_ERR_HIP;
    }
    (void)hipFree(d_res);
    if (rc != HEAAC_OK)
        return rc;
    if (!h_res.count)
        return HEAAC_OK;
    if (first_bad) *first_bad = (size_t)h_res.first;
    if (rule) {
        // name the rule of the first failing frame (one more small copy; the failing path may be slow)
        HeaacSbrFrame fr; HeaacPsFrame ps; HeaacSbrHeader hd;
        *rule = -1;
        if (hipMemcpy(&fr, d_sbr + h_res.first, sizeof(fr), hipMemcpyDeviceToHost) == hipSuccess) {
            int r = fr.hdr >= n_hdr ? HEAAC_BAD_HDR_INDEX : 0;
            if (!r && hipMemcpy(&hd, d_hdr + fr.hdr, sizeof(hd), hipMemcpyDeviceToHost) == hipSuccess) {
                fr.hdr = 0;
                r = heaac_check_sbr_frame(&fr, &hd, 1, ncore);
            }
            if (!r && cfg == HEAAC_CFG_HEV2 &&
                hipMemcpy(&ps, d_ps + h_res.first, sizeof(ps), hipMemcpyDeviceToHost) == hipSuccess)
                r = heaac_check_ps_frame(&ps);
            *rule = r;
        }
    }
    return HEAAC_ERR_ARG;
}
