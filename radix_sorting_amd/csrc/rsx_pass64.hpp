// rsx_pass64.hpp -- the LEVEL-2 pass of a keys-only sort of 8-byte keys without a histogram, in whole 64-byte atoms (round 6).
//
// rsx_pass2w.hpp's pass with route 5's tables (rsx_hybrid.hpp): the tile table of rsx_seg_tiles_kernel (tiles cut
// from both ends of the level-1 slots), SegCtl::shift2 as the digit's position, 65536 slots of slack_cap values in the scratch
// array, the cursors where rsx_seg_slack_plan_kernel reads the (digit, digit) counts (front cursors, then back cursors: the
// format of rsx_pass16a_kernel), SegCtl::overflow as the verdict.  OT = u32: SegCtl::narrow -- the leaves' columns all lie in the low
// word, the slots hold the low word of the DERIVED keys and rsx_leafk_kernel's SLOT32 form reads them from both ends.
// Replaces rsx_scatter2_kernel<u64, NoVal, u32, ..., KTO = u32, SEG> (chained tiles, ragged runs: 0.81-0.85 ms for 2^28 keys, 0.48-0.50
// of the HBM peak, traffic 1.07 x; RSX_NO_PASS64A=1 brings it back).
#pragma once

#include "rsx_hybrid.hpp"
#include "rsx_leaf16.hpp"
#include "rsx_pass2w.hpp"

namespace rsx {

static_assert(Pass2wCfg<u32>::BACK == LEAF16_BACK && Pass2wCfg<u64>::BACK == LEAF16_BACK, "the leaves read a slot's back where the pass writes it");

template <typename KT> struct Pass64Policy {
	const KT *kin, *kin_hi;   // the level-1 slots lie in two arrays (SegArgs, rsx_scatter2.hpp): buckets below lo_slots in kin, the others at the same element index of kin_hi
	u32 lo_slots;
	const SegTile *tiles;
	const SegCtl *ctl;
	const Plan *plan;
	u32 *cursors;             // [65536] front cursors, [65536] back cursors (zeroed by rsx_blind_precheck_kernel with the status words)
	u32 slack_cap;
	u32 *overflow;
	u32 narrow;               // the form this launch is (SegCtl::narrow decides which one works): 0 whole keys in and out, 1 the low words of 8-byte keys out, 2 low words in (KT = u32: the level-1 pass wrote them) and out
	__device__ __forceinline__ bool go() const
	{
		return ctl->blind == BLIND_GO && plan->hyb == HYB_TWO_LEVEL && ctl->narrow == narrow;
	}
	__device__ __forceinline__ u32 ntiles() const { return ctl->ntiles; }
	__device__ __forceinline__ u32 per(u32 grid) const { return (ctl->ntiles + grid - 1u) / grid; }
	__device__ __forceinline__ Pass2wTile<KT> tile(u32 t) const
	{
		const SegTile st = tiles[t];
		return Pass2wTile<KT>{((kin_hi && st.bucket >= lo_slots) ? kin_hi : kin) + st.beg, st.cnt, st.bucket};
	}
	__device__ __forceinline__ u32 shift(u32) const { return ctl->shift2; }
	__device__ __forceinline__ u32 cap(u32) const { return slack_cap; }
	__device__ __forceinline__ u32 slot(u32 bucket, u32 d) const { return (bucket * 256u + d) * slack_cap; }
	__device__ __forceinline__ u32 *front(u32 bucket, u32 d) const { return cursors + bucket * 256u + d; }
	__device__ __forceinline__ u32 *back(u32 bucket, u32 d) const { return cursors + 65536u + bucket * 256u + d; }
	__device__ __forceinline__ void lost(u32) const { atomicOr(overflow, 1u); }
	__device__ __forceinline__ u32 dump() const { return 65536u * slack_cap; }   // (a tile of padding behind the last slot)
};

// C: the tile shape (Pass2wCfg); the form that reads four-byte values takes 24 Ki-key tiles, one workgroup per CU (Pass64aCfgLow)
typedef Pass2wCfg<u32, 24> Pass64aCfgLow;
template <typename KT, typename OT, typename C = Pass2wCfg<OT> >
__global__ __launch_bounds__(C::BLOCK, 4 * C::WGS) void rsx_pass64a_kernel(const KT *__restrict__ kin, const KT *__restrict__ kin_hi,
                                                                             u32 lo_slots, OT *__restrict__ kout,
                                                                             const SegTile *__restrict__ tiles,
                                                                             const SegCtl *__restrict__ ctl, const Plan *__restrict__ plan,
                                                                             u32 *__restrict__ cursors, u32 slack_cap,
                                                                             u32 *__restrict__ overflow, KdfArgs<KT> ka)
{
	__shared__ Pass2wSmem<OT, C> sm;
	const Pass64Policy<KT> pol{kin, kin_hi, lo_slots, tiles, ctl, plan, cursors, slack_cap, overflow, sizeof(KT) == 4 ? 2u : sizeof(OT) == 4 ? 1u : 0u};
	// (ordinary loads: with non-temporal ones this pass is 2 % shorter and the leaves behind it 4 % longer -- 2^28 keys & 0xFFFFFFFFFF:
	// pass 0.795 / leaves 0.742 ms against 0.812 / 0.712, two rounds on one box, profiles/r06/pass64a_ab.txt; the form that reads
	// four-byte values: pass 0.583 / leaves 0.759 against 0.545 / 0.734, three rounds, profiles/r06/narrow_level1_ab.txt)
	pass2w_body<KT, OT, false, Pass64Policy<KT>, C>(pol, kout, ka, sm);
}

}  // namespace rsx
