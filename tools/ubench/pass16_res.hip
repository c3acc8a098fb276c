// pass16_res.hip -- resource check: tools/kres.sh tools/ubench/pass16_res.hip pass16
#include "rsx_scatter2.hpp"
#include "rsx_pass16.hpp"
namespace rsx {
#define INST(D, W) template __global__ void rsx_pass16_kernel<u32, D, Pass16Cfg<W>>(const u32 *, const u32 *, u32, unsigned short *, const SegTile *, const u32 *, const SegCtl *, const Plan *, u32 *, u32, u32 *, KdfArgs<u32>, u32)
INST(DIG_PLAIN, 1);
INST(DIG_PLAIN, 2);
INST(DIG_GENERIC, 1);
INST(DIG_GENERIC, 2);
}
namespace rsx {
template __global__ void rsx_pass16a_kernel<u32, DIG_PLAIN>(const u32 *, const u32 *, u32, unsigned short *, const SegTile *, const SegCtl *, const Plan *, u32 *, u32, u32 *, KdfArgs<u32>);
template __global__ void rsx_pass16a_kernel<u32, DIG_GENERIC>(const u32 *, const u32 *, u32, unsigned short *, const SegTile *, const SegCtl *, const Plan *, u32 *, u32, u32 *, KdfArgs<u32>);
}
