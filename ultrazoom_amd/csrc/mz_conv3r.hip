// conv3r_kernel instantiations and launcher (the kernel: mz_conv3r.h).
#include "mz_conv3r.h"

#ifndef MZ_R_NSEG
#define MZ_R_NSEG 3  // weight segments per chunk of the plain / sub-pixel variants (the fused variant always uses 3).  Three since round 4:
                     // with the halo image issued behind the weights and left in flight for a step (HALO_LATE, mz_conv3r.h) the 3- and
                     // 6-chunk layers run 3.5 - 6 % faster than with two segments, the deep layers the same (tools/ab_libs.sh base n3)
#endif

namespace mz {

template <class TT, int NSEG, int EPI, bool SILU = false, int GEO = 0, bool RAG = false> static hipError_t r_launch(const ConvArgs& a, hipStream_t s) {
    constexpr size_t lds = r3::Seg<NSEG>::lds_bytes(EPI == EPI_FUSEDMIX);
    static bool ready[16] = {};  // per device ordinal: the dynamic-LDS limit of this instantiation has been raised
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (!ready[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3r_kernel<TT, NSEG, EPI, SILU, GEO, RAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        ready[dev] = true;
    }
    hipLaunchKernelGGL((conv3r_kernel<TT, NSEG, EPI, SILU, GEO, RAG>), dim3(a.persist), dim3(512), lds, s, a);
    return hipGetLastError();
}
template <class TT> static hipError_t r_epi(const ConvArgs& a, hipStream_t s) {
    if (a.ragged_planes) {  // two chunks, the second with a.ragged_planes real planes (Cin = 48): conv1 + SiLU on 8 x 48 tiles
        if (a.nchunks16 != 2 || a.ragged_planes < 1 || a.ragged_planes > 3 || a.epi != EPI_STORE || !a.silu || a.geo != 0) return hipErrorInvalidValue;
        return r_launch<TT, 3, EPI_STORE, true, 0, true>(a, s);
    }
    if (a.nchunks16 < 3) return hipErrorInvalidValue;
    if (a.geo == 1) {  // 8 x 40 pixel tiles (five pixel fragments per wave): plain / SiLU / sub-pixel stores
        switch (a.epi) {
            case EPI_STORE: return a.silu ? r_launch<TT, MZ_R_NSEG, EPI_STORE, true, 1>(a, s) : r_launch<TT, MZ_R_NSEG, EPI_STORE, false, 1>(a, s);
            case EPI_D2S: return r_launch<TT, MZ_R_NSEG, EPI_D2S, false, 1>(a, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (a.epi) {
        case EPI_STORE: return a.silu ? r_launch<TT, MZ_R_NSEG, EPI_STORE, true>(a, s) : r_launch<TT, MZ_R_NSEG, EPI_STORE, false>(a, s);
        case EPI_D2S: return r_launch<TT, MZ_R_NSEG, EPI_D2S>(a, s);
        case EPI_FUSEDMIX: return a.wmix16 ? r_launch<TT, 3, EPI_FUSEDMIX>(a, s) : hipErrorInvalidValue;  // wmix16: PackArgs::frag16 = 2
        default: return hipErrorInvalidValue;
    }
}

// a.persist workgroups of 512 threads; a.tiles_x / tiles_y / mtiles describe 8 x 48 (a.geo = 0) or 8 x 40 (a.geo = 1) tiles; 96-channel N
// tiles; >= 3 chunks (a.ragged_planes != 0: two, see mz_conv3r.h)
hipError_t launch_conv3r(int dtype, const ConvArgs& a, hipStream_t s) {
    if (a.persist <= 0 || (a.persist & 7) || a.nchunks16 < 2 || a.geo < 0 || a.geo > 1) return hipErrorInvalidValue;
    switch (dtype) {
        case DT_BF16: return r_epi<TBF16>(a, s);
        case DT_F16: return r_epi<TF16>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mz
