// leafc_res.hip -- resource check: tools/kres.sh tools/ubench/leafc_res.hip leafc
#include "rsx_scatter2.hpp"
#include "rsx_leafc.hpp"
namespace rsx {
template __global__ void rsx_leafc_kernel<u32, LeafCCfg<5>>(u32 *, u32 *, const Plan *, const LeafSeg *, const SegCtl *, KdfArgs<u32>, u32, u32, const uint16_t *, u32, const u32 *);
template __global__ void rsx_leafc_kernel<u32, LeafCCfg<2>>(u32 *, u32 *, const Plan *, const LeafSeg *, const SegCtl *, KdfArgs<u32>, u32, u32, const uint16_t *, u32, const u32 *);
}
namespace rsx {
#define L16(B, CAPV, W, NB) template __global__ void rsx_leaf16_kernel<u32, Leaf16Cfg<B, CAPV, W, NB>>(u32 *, u32 *, const Plan *, const LeafSeg *, SegCtl *, KdfArgs<u32>, u32, u32, const uint16_t *, u32, u32 *, u32)
L16(512, 10240, 8, 13);
L16(1024, 10240, 8, 13);
L16(1024, 10240, 8, 14);
L16(1024, 20480, 8, 14);
L16(1024, 40960, 4, 14);
}
