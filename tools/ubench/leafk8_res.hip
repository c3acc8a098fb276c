// leafk8_res.hip -- resource check: tools/kres.sh tools/ubench/leafk8_res.hip leafk8
#include "rsx_scatter2.hpp"
#include "rsx_leaf16.hpp"
namespace rsx {
template __global__ void rsx_leafk8_kernel<u64, u64, LeafK8Cfg>(u64 *, u64 *, const Plan *, const LeafSeg *, SegCtl *, KdfArgs<u64>, u32, u32, const u64 *, u32, u32 *, u32);
}
