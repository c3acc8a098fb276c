// leaf_res.hip -- resource check of the leaf kernels: tools/kres.sh tools/ubench/leaf_res.hip leaf
#include "rsx_scatter2.hpp"
#include "rsx_leaf16.hpp"
#include "rsx_leafk2.hpp"
namespace rsx {
#define INST_K(CT, SL, ...) template __global__ void rsx_leafk_kernel<u64, CT, __VA_ARGS__, SL>(u64 *, u64 *, const Plan *, const LeafSeg *, SegCtl *, KdfArgs<u64>, u32, u32, const u64 *, u32, u32 *, u32)
#define INST_P(...) template __global__ void rsx_leafp_kernel<u32, u32, __VA_ARGS__>(const u32 *, const u32 *, u32, u32 *, u32 *, const Plan *, const LeafSeg *, SegCtl *, KdfArgs<u32>, u32 *, u32)
#define INST_16(...) template __global__ void rsx_leaf16_kernel<u32, __VA_ARGS__>(u32 *, u32 *, const Plan *, const LeafSeg *, SegCtl *, KdfArgs<u32>, u32, u32, const uint16_t *, u32, u32 *, u32)
#define INST_16W(...) template __global__ void rsx_leaf16w_kernel<u32, __VA_ARGS__>(u32 *, u32 *, const Plan *, const LeafSeg *, SegCtl *, KdfArgs<u32>, u32, u32, const uint16_t *, u32)
#define INST_16Q(...) template __global__ void rsx_leaf16q_kernel<u32, __VA_ARGS__>(u32 *, u32 *, const Plan *, const LeafSeg *, SegCtl *, KdfArgs<u32>, u32, u32, const uint16_t *, u32)
INST_K(u64, false, LeafKCfg<512, 5120, 8, 12>);
INST_K(u64, false, LeafKCfg<256, 2560, 8, 11>);
INST_K(u64, false, LeafKCfg<128, 1280, 6, 10>);
INST_K(u64, false, LeafKCfg<64, 512, 8, 10>);
INST_K(u64, false, LeafKCfg<64, 256, 8, 9>);
INST_K(u32, false, LeafKCfg<512, 5120, 8, 12>);
INST_K(u32, true, LeafKCfg<512, 5120, 8, 12>);
INST_K(u32, false, LeafKCfg<128, 1280, 6, 10>);
INST_P(LeafKCfg<512, 5120, 8, 12>);
INST_P(LeafKCfg<256, 2560, 8, 11>);
INST_P(LeafKCfg<128, 1280, 6, 10>);
INST_P(LeafKCfg<64, 512, 8, 10>);
INST_P(LeafKCfg<64, 256, 8, 9>);
INST_16(Leaf16Cfg<256, 5120, 8, 12>);
INST_16(Leaf16Cfg<128, 2560, 8, 11>);
INST_16W(Leaf16WCfg<1024, 10, 4>);
INST_16W(Leaf16WCfg<512, 9, 4>);
INST_16Q(Leaf16QCfg<4>);
}
namespace rsx {
INST_16W(Leaf16WCfg<2048, 10, 4>);
}
