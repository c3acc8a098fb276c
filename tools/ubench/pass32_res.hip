// pass32_res.hip -- resource check: tools/kres.sh tools/ubench/pass32_res.hip pass32
#include "rsx_scatter2.hpp"
#include "rsx_pass32.hpp"
namespace rsx {
template __global__ void rsx_pass32a_kernel<u32, DIG_PLAIN, true>(const u32 *, u64, u32 *, u32, u32, u32, u32, const SegCtl *, u32 *, u32 *, KdfArgs<u32>);
template __global__ void rsx_pass32a_kernel<u32, DIG_GENERIC, true>(const u32 *, u64, u32 *, u32, u32, u32, u32, const SegCtl *, u32 *, u32 *, KdfArgs<u32>);
template __global__ void rsx_pass32a_kernel<u32, DIG_GENERIC, false>(const u32 *, u64, u32 *, u32, u32, u32, u32, const SegCtl *, u32 *, u32 *, KdfArgs<u32>);
template __global__ void rsx_pass32a_kernel<u32, DIG_PLAIN, false>(const u32 *, u64, u32 *, u32, u32, u32, u32, const SegCtl *, u32 *, u32 *, KdfArgs<u32>);
}
namespace rsx {
template __global__ void rsx_pass32a_kernel<u32, DIG_PLAIN, false, Pass32aCfgT<12>>(const u32 *, u64, u32 *, u32, u32, u32, u32, const SegCtl *, u32 *, u32 *, KdfArgs<u32>);
}
namespace rsx {
template __global__ void rsx_pass32a_kernel<u64, DIG_PLAIN, false, Pass32aCfgT<14>>(const u64 *, u64, u64 *, u32, u32, u32, u32, const SegCtl *, u32 *, u32 *, KdfArgs<u64>);
template __global__ void rsx_pass32a_kernel<u64, DIG_GENERIC, false, Pass32aCfgT<14>>(const u64 *, u64, u64 *, u32, u32, u32, u32, const SegCtl *, u32 *, u32 *, KdfArgs<u64>);
template __global__ void rsx_pass32a_kernel<u64, DIG_PLAIN, true, Pass32aCfgT<14>>(const u64 *, u64, u64 *, u32, u32, u32, u32, const SegCtl *, u32 *, u32 *, KdfArgs<u64>);
}
namespace rsx {
template __global__ void rsx_pass32a_kernel<u32, DIG_PLAIN, false, Pass32aCfgT<28, false>>(const u32 *, u64, u32 *, u32, u32, u32, u32, const SegCtl *, u32 *, u32 *, KdfArgs<u32>);
}
