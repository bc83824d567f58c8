// conv3t_kernel instantiations and launcher (the kernel: mz_conv3t.h).
#include "mz_conv3t.h"

namespace mz {

template <class TT, int EPI, bool SILU = false> static hipError_t t_launch(const ConvArgs& a, hipStream_t s) {
    constexpr size_t lds = t3::lds_bytes(EPI == EPI_FUSEDMIX);
    static bool ready[16] = {};  // per device ordinal: the dynamic-LDS limit of this instantiation has been raised
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (!ready[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3t_kernel<TT, EPI, SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        ready[dev] = true;
    }
    hipLaunchKernelGGL((conv3t_kernel<TT, EPI, SILU>), dim3(a.persist), dim3(512), lds, s, a);
    return hipGetLastError();
}
template <class TT> static hipError_t t_epi(const ConvArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_STORE: return a.silu ? t_launch<TT, EPI_STORE, true>(a, s) : t_launch<TT, EPI_STORE, false>(a, s);
        case EPI_FUSEDMIX: return a.wmix16 ? t_launch<TT, EPI_FUSEDMIX>(a, s) : hipErrorInvalidValue;  // wmix16: PackArgs::frag16 = 4
        default: return hipErrorInvalidValue;
    }
}

// a.persist workgroups of 512 threads; a.tiles_x / tiles_y / mtiles describe 12 x 64 tiles; ONE N tile of <= 48 channels; a.wpk16 =
// weights packed with three 16-channel fragments per tap (PackArgs::nfr = 3); a.nchunks16 = 3 or >= 6 chunks of 32 channels
hipError_t launch_conv3t(int dtype, const ConvArgs& a, hipStream_t s) {
    if (a.persist <= 0 || (a.persist & 7) || a.ntiles != 1 || !(a.nchunks16 == 3 || a.nchunks16 >= 6)) return hipErrorInvalidValue;
    switch (dtype) {
        case DT_BF16: return t_epi<TBF16>(a, s);
        case DT_F16: return t_epi<TF16>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mz
