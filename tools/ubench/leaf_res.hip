// leaf_res.hip -- resource check of leaf-kernel variants: tools/kres.sh tools/ubench/leaf_res.hip leaf
#include "rsx_scatter2.hpp"
#include "rsx_leaf16.hpp"
#include "rsx_leafk2.hpp"
namespace rsx {
#define INST_K(CFG) template __global__ void rsx_leafk_kernel<u64, u64, CFG, false>(u64 *, u64 *, const Plan *, const LeafSeg *, SegCtl *, KdfArgs<u64>, u32, u32, const u64 *, u32, u32 *, u32)
#define INST_K2(...) template __global__ void rsx_leafk2_kernel<u64, u64, __VA_ARGS__>(u64 *, u64 *, const Plan *, const LeafSeg *, SegCtl *, KdfArgs<u64>, u32, u32, const u64 *, u32, u32 *, u32)
typedef LeafKCfg<512, 5120, 6, 12> K8;
INST_K(K8);
INST_K2(LeafK2Cfg<512, 5120, 6, 12, false>);
INST_K2(LeafK2Cfg<512, 5120, 6, 12, true>);
INST_K2(LeafK2Cfg<512, 5120, 6, 12, false, 0, true>);
INST_K2(LeafK2Cfg<512, 5120, 8, 12, false, 0, true>);
}
